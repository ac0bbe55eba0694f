#!/usr/bin/env python3
"""The run of the reference's tutorial (docs/time_evolution.ipynb: 9 qubits, horizons of 1.2 with 2, 4, ..., 12 ansatz layers, L-BFGS, both
objectives) through time_evol.run_simulation with the reference's defaults; the notebook prints 62.6 s (MPS objective) and 7.6 s (state-vector
objective) for it on unstated hardware (docs/time_evolution.ipynb:484,820).  Usage: python tools/asp9_notebook_run.py [n]"""
import sys
import time

sys.path.insert(0, ".")
from aqc_research_amd.model_sp_lhs import time_evol as te   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for objective in ("sur_fast_mps_trotter", "sur_max"):
    opts = te.UserOptions(num_qubits=n, objective=objective)     # 6 horizons, 2 layers more per horizon, maxiter 40, fidelity_thr 0.995, trunc_thr 1e-6
    te.run_simulation(te.UserOptions(num_qubits=n, objective=objective, num_horizons=1))   # warm-up: library load, plans of the first horizon
    t0 = time.perf_counter()
    res = te.run_simulation(opts)
    wall = time.perf_counter() - t0
    per = ", ".join(f"{r['time']:.2f}" for r in res)
    fids = ", ".join(f"{r['fid_a1_vs_gt']:.4f}" for r in res)
    evals = sum(r["num_fun_ev"] for r in res)
    print(f"n = {n}, objective {objective}: {wall:.2f} s in total; per horizon {per} s; fidelity vs ground truth {fids}; {evals} objective evaluations", flush=True)
