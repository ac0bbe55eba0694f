"""Timing of the device SVD (aqc_svd): persistent two-level kernel vs one launch per round."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd.mps_engine import svd

rng = np.random.default_rng(0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for shape in ([(only, only)] if only else [(128, 128), (256, 256), (512, 256), (512, 512)]):
    a = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    for mode in (("1",) if only else ("1", "0")):
        os.environ["AQC_SVD_BLOCKED"] = mode
        svd(a)
        t0 = time.perf_counter()
        for _ in range(3):
            u, s, vh, sweeps = svd(a)
        dt = (time.perf_counter() - t0) / 3
        err = np.abs((u * s) @ vh - a).max()
        print(f"{shape} blocked={mode}: {dt * 1e3:8.2f} ms  sweeps {sweeps}  err {err:.1e}", flush=True)
