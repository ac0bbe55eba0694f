#!/usr/bin/env python3
"""Wall time of the native MPS engine at config 3's largest bond (16 qubits, 40 blocks, chi = 256): V^H|phi> and the
gate-by-gate gradient, three repetitions.  Run it under rocprofv3 --kernel-trace --stats to see the kernels behind it."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.mps_engine import DeviceMPS, fast_dot_gradient_mps, v_dagger_mul_mps  # noqa: E402
from oracle import aqc_oracle as orc  # noqa: E402
from tests.helpers import canonical_mps  # noqa: E402

n = 16
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(1)
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
th = orc.rand_thetas(circ.num_thetas, rng)
raw = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
phi = canonical_mps(raw / np.linalg.norm(raw), chi)
src = DeviceMPS.from_qiskit(phi)
zero = DeviceMPS.basis_state(n, 0)
for rep in range(3):
    t0 = time.perf_counter()
    vh = v_dagger_mul_mps(circ, th, src, trunc_thr=1e-16)
    t1 = time.perf_counter()
    g = fast_dot_gradient_mps(circ, th, zero, vh, trunc_thr=1e-16)
    t2 = time.perf_counter()
    print(f"rep {rep}: V^H {1e3 * (t1 - t0):.1f} ms (bonds <= {vh.bond_dims.max()}), gradient {1e3 * (t2 - t1):.1f} ms, |g| {np.linalg.norm(g):.6f}", flush=True)
