#!/usr/bin/env python3
"""v_mul_mps / v_dagger_mul_mps / fast_dot_gradient_mps of a Trotter circuit at n qubits: the single-lane engine (one ABI call, a launch per gate) against one
lockstep lane (the gates of a circuit layer in one launch).  Usage: python tools/mps_apply_probe.py [n] [layers]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import TrotterAnsatz                                    # noqa: E402
from aqc_research_amd.circuit_structures import make_trotter_like_circuit     # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index   # noqa: E402
from aqc_research_amd.mps_engine import DeviceMPS, fast_dot_gradient_mps, v_dagger_mul_mps, v_mul_mps   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 6
circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
th = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.2 * layers, delta=1.0)
basis = DeviceMPS.basis_state(n, neel_state_index(n))
for method in ("single", "lockstep"):
    out = v_mul_mps(circ, th, basis, trunc_thr=1e-12, method=method)
    t0 = time.perf_counter()
    for _ in range(3):
        out = v_mul_mps(circ, th, basis, trunc_thr=1e-12, method=method)
    fwd = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        back = v_dagger_mul_mps(circ, th, out, trunc_thr=1e-12, method=method)
    inv = (time.perf_counter() - t0) / 3
    g = fast_dot_gradient_mps(circ, th + 0.01, basis, back, trunc_thr=1e-6, method=method)
    t0 = time.perf_counter()
    g = fast_dot_gradient_mps(circ, th + 0.01, basis, back, trunc_thr=1e-6, method=method)
    grad = time.perf_counter() - t0
    print(f"n={n} layers={layers} ({circ.num_blocks} blocks) {method:9s}: V|neel> {fwd * 1e3:7.1f} ms (max bond {out.bond_dims.max()}), "
          f"V^H of it {inv * 1e3:7.1f} ms, |<neel|V^H V|neel>| = {abs(back.dot(basis)):.12f}; fast_dot_gradient {grad * 1e3:7.1f} ms (|g| {np.linalg.norm(g):.6f})",
          flush=True)
