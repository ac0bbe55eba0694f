#!/usr/bin/env python3
"""Native MPS engine beyond dense reach: objective+gradient evaluations per second at n qubits on one lane and on several
lanes driven by host threads (every DeviceMPS owns its stream; ctypes releases the GIL during the native calls).
Usage: python tools/mps_lanes_probe.py [n] [layers] [trunc_thr]"""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import TrotterAnsatz                                    # noqa: E402
from aqc_research_amd.circuit_structures import make_trotter_like_circuit     # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index   # noqa: E402
from aqc_research_amd.mps_engine import DeviceMPS, fast_dot_gradient_mps, v_dagger_mul_mps, v_mul_mps   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
neel = neel_state_index(n)
th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.6 * layers, delta=1.0)
tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=0.6 * layers, delta=1.0)
rng = np.random.default_rng(3)


def make_lane():
    basis = DeviceMPS.basis_state(n, neel)
    target = v_mul_mps(tcirc, tth, basis, trunc_thr=1e-12)
    return basis, target


def evaluate(lane, th):
    basis, target = lane
    vh = v_dagger_mul_mps(circ, th, target, trunc_thr=thr)
    h0 = basis.dot(vh)
    g = fast_dot_gradient_mps(circ, th, basis, vh, trunc_thr=thr)
    dims = vh.bond_dims.max()
    vh.close()
    return h0, g, dims


lane0 = make_lane()
print(f"n={n} layers={layers} T={circ.num_thetas} trunc_thr={thr:g} target bonds max {lane0[1].bond_dims.max()}", flush=True)
h0, g, dims = evaluate(lane0, th0)
print(f"fidelity of the Trotter point {abs(h0) ** 2:.6f}, |g| {np.linalg.norm(g):.3e}, max bond of V^H target {dims}", flush=True)
for lanes in (1, 16):
    ls = [lane0] + [make_lane() for _ in range(lanes - 1)]
    ths = [th0 + 0.02 * rng.standard_normal(th0.size) for _ in range(lanes)]
    reps = 3
    with ThreadPoolExecutor(lanes) as ex:
        list(ex.map(evaluate, ls, ths))
        t0 = time.perf_counter()
        for _ in range(reps):
            list(ex.map(evaluate, ls, ths))
        dt = time.perf_counter() - t0
    print(f"lanes {lanes:2d}: {reps * lanes / dt:8.2f} evals/s  ({dt / reps * 1e3:7.1f} ms per round)", flush=True)
    for b, t in ls[1:]:
        b.close(); t.close()

# the same batch in lockstep (aqc_mpsb_*): one launch per step of the walk for all lanes
from aqc_research_amd.mps_engine import LockstepLanes   # noqa: E402

for lanes in (1, 16, 64, 256, 1024):
    ths = np.stack([th0 + 0.02 * rng.standard_normal(th0.size) for _ in range(lanes)])
    lk = LockstepLanes(n, lanes).set_targets(lane0[1]).set_lhs(lane0[0])
    h, g = lk.evaluate(circ, ths, trunc_thr=thr)
    if lanes == 1:
        ref = evaluate(lane0, ths[0])
        print(f"lockstep vs single lane: |dh| {abs(h[0] - ref[0]):.2e}  max|dg| {np.abs(g[0] - ref[1]).max():.2e}", flush=True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        lk.evaluate(circ, ths, trunc_thr=thr)
    dt = time.perf_counter() - t0
    print(f"lockstep lanes {lanes:3d}: {reps * lanes / dt:8.2f} evals/s  ({dt / reps * 1e3:7.1f} ms per round)", flush=True)
    lk.close()
