#!/usr/bin/env python3
"""The ASP driver (time_evol.run_simulation = time_evol_best_init.py:337-395) beyond dense reach: H horizons at n qubits, S random restarts
each, MPS objective, restarts optimised together on the lockstep lanes of the MPS engine (while the bonds of target and walk stay <= 32: at 32
qubits that is the first horizon, t = 1.2; later ones fall back to one restart after the other on the single-lane engine -- minutes).  Usage: python tools/horizon32_demo.py [n] [H] [S] [maxiter]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd.model_sp_lhs import time_evol as te   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
maxiter = int(sys.argv[4]) if len(sys.argv) > 4 else 15
opts = te.UserOptions(num_qubits=n, num_horizons=H, num_layers_inc=1, maxiter=maxiter, objective="sur_fast_mps_trotter", fidelity_thr=0.999,
                      num_seeds=S, vectorised_lbfgs=True, theta_jitter=0.02, trunc_thr_target=1e-6)   # (targets at the objective's own threshold: their
                                                                                               # bonds stay within the lockstep lanes' 32 for longer)
t0 = time.perf_counter()
res = te.run_simulation(opts)
wall = time.perf_counter() - t0
for r in res:
    f = np.array(r["fidelities"])
    print(f"horizon {r['horizon']}: t = {r['evol_time']}, {r['num_layers']} layer(s), {r['num_thetas']} parameters, {S} restarts: best fidelity {r['fidelity']:.5f} "
          f"(restart {r['best_restart']}), min {f.min():.5f}, fid(t1, gt) {r['fid_t1_vs_gt']:.5f}, {r['num_fun_ev']} evaluations, {r['time']:.2f} s"
          f"{' -- ' + r['route'] if 'route' in r else ''}", flush=True)
evals = sum(r["num_fun_ev"] for r in res)
print(f"n = {n}: {H} horizons x {S} restarts in {wall:.2f} s ({evals} objective+gradient evaluations, {evals / wall:.0f} evals/s incl. target generation)", flush=True)
