#!/usr/bin/env python3
"""Wall time of S independent L-BFGS state-preparation runs (same Trotter ansatz, different targets / starts):
one after the other on a one-lane workspace vs. in lockstep as lanes of one batched workspace."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd.lockstep import LockstepBatch  # noqa: E402
from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax  # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz, trotter_state  # noqa: E402
from aqc_research_amd.optimizer import AqcOptimizer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 20
circ = trotter_ansatz(n, 2, True)
neel = neel_state_index(n)
rng = np.random.default_rng(3)
base = trotter_state(n, evol_time=1.2, num_steps=6, delta=1.0, second_order=True)
th_t = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2, delta=1.0)
cases = []
for j in range(S):
    pert = base + 0.05 * (rng.standard_normal(base.size) + 1j * rng.standard_normal(base.size)) / np.sqrt(base.size)
    cases.append((pert / np.linalg.norm(pert), th_t + 0.05 * rng.standard_normal(th_t.size)))


def optimise(target, th0, workspace=None):
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0, maxiter=maxiter)
    if workspace is not None:
        user["workspace"] = workspace
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=maxiter).optimize(objv, circ, th0)
    return res["fidelity"], res["num_fun_ev"]


optimise(*cases[0])  # warm-up (library load, plans)
t0 = time.perf_counter()
seq = [optimise(t, th) for t, th in cases]
t_seq = time.perf_counter() - t0
batch = LockstepBatch(circ, nlanes=S)
t0 = time.perf_counter()
out = batch.run([(lambda view, t=t, th=th: optimise(t, th, workspace=view)) for t, th in cases])
t_lock = time.perf_counter() - t0
nfev = sum(r[1] for r in seq)
print(f"n={n} T={circ.num_thetas} jobs={S} maxiter={maxiter}: sequential {t_seq:.3f} s ({nfev} evaluations, {nfev / t_seq:,.0f} evals/s) | "
      f"lockstep {t_lock:.3f} s ({batch.rounds} rounds, {batch.native_calls} batched calls, {nfev / t_lock:,.0f} evals/s) | speed-up {t_seq / t_lock:.2f}x | "
      f"max fidelity difference {max(abs(a[0] - b[0]) for a, b in zip(seq, out)):.2e}")
batch.close()
