#!/usr/bin/env python3
"""cProfile of one L-BFGS state-preparation run (host-side overhead per evaluation)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax  # noqa: E402
from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz, trotter_state  # noqa: E402
from aqc_research_amd.optimizer import AqcOptimizer  # noqa: E402

n = 12
circ = trotter_ansatz(n, 2, True)
neel = neel_state_index(n)
target = trotter_state(n, evol_time=1.2, num_steps=6, delta=1.0, second_order=True)
th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2, delta=1.0) + 0.05 * np.random.default_rng(0).standard_normal(circ.num_thetas)


def run():
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0, maxiter=60)
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    return AqcOptimizer(optimizer_name="lbfgs", maxiter=60).optimize(objv, circ, th0)


run()
pr = cProfile.Profile()
pr.enable()
res = run()
pr.disable()
print("num_fun_ev", res["num_fun_ev"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
