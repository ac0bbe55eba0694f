"""One large aqc_zgemm (both op(A) forms) for rocprofv3 --kernel-trace: kernel time -> fp64 MFMA TFLOP/s."""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd.engine import zgemm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(1)
a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
b = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
for conj_t in (False, True):
    c = zgemm(a, b, conj_t)
    ref = (a.conj().T if conj_t else a) @ b
    print("n", n, "conj_t", conj_t, "rel err", np.abs(c - ref).max() / np.abs(ref).max(), "flops", 8.0 * n ** 3)
