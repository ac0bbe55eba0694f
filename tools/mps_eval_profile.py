#!/usr/bin/env python3
"""One lane of the native MPS engine at n qubits: wall time of V^H and of the gradient walk (for rocprofv3 --kernel-trace --stats)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import TrotterAnsatz                                    # noqa: E402
from aqc_research_amd.circuit_structures import make_trotter_like_circuit     # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index   # noqa: E402
from aqc_research_amd.mps_engine import DeviceMPS, fast_dot_gradient_mps, v_dagger_mul_mps, v_mul_mps   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layers, thr = 2, 1e-6
circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
neel = neel_state_index(n)
th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.6 * layers, delta=1.0)
tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=0.6 * layers, delta=1.0)
basis = DeviceMPS.basis_state(n, neel)
target = v_mul_mps(tcirc, tth, basis, trunc_thr=1e-12)
for rep in range(3):
    t0 = time.perf_counter()
    vh = v_dagger_mul_mps(circ, th0, target, trunc_thr=thr)
    t1 = time.perf_counter()
    g = fast_dot_gradient_mps(circ, th0, basis, vh, trunc_thr=thr)
    t2 = time.perf_counter()
    print(f"rep {rep}: V^H {1e3 * (t1 - t0):.1f} ms, gradient {1e3 * (t2 - t1):.1f} ms; blocks {circ.num_blocks}, T {circ.num_thetas}", flush=True)
    vh.close()
