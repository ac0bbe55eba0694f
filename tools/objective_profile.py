#!/usr/bin/env python3
"""Host-side profile of the drop-in objective under AqcOptimizer(lbfgs): where the ~45 us per objective+gradient pair that are not the
native call go (cProfile over a 60-iteration run at the headline shape)."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import ParametricCircuit                                   # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure          # noqa: E402
from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax   # noqa: E402
from aqc_research_amd.optimizer import AqcOptimizer                              # noqa: E402

n = 16
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
rng = np.random.default_rng(5)
t = rng.random(1 << n) + 1j * rng.random(1 << n)
t /= np.linalg.norm(t)
user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: 0, enable_optim_stats=False, verbose=0, maxiter=60)
objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
objv.set_target(t)
th0 = 0.2 * np.pi * (2 * rng.random(circ.num_thetas) - 1)
AqcOptimizer(optimizer_name="lbfgs", maxiter=5).optimize(objv, circ, th0)
t0 = time.perf_counter()
res = AqcOptimizer(optimizer_name="lbfgs", maxiter=60).optimize(objv, circ, th0)
dt = time.perf_counter() - t0
print(f"{res['num_fun_ev']} pairs in {dt * 1e3:.2f} ms = {dt / res['num_fun_ev'] * 1e6:.1f} us per pair")
pr = cProfile.Profile()
pr.enable()
res = AqcOptimizer(optimizer_name="lbfgs", maxiter=60).optimize(objv, circ, th0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
