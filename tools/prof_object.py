"""cProfile of SpSurrogateObjectiveMax under AqcOptimizer(lbfgs) at the headline shape (one lane): where the host time of the
literal drop-in path goes.  python tools/prof_object.py [n] [blocks]"""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax
from aqc_research_amd.optimizer import AqcOptimizer
from oracle import aqc_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(5)
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: 0, enable_optim_stats=False, verbose=0, maxiter=40, device=0)
objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
objv.set_target(orc.rand_state(n, rng))
th0 = 0.2 * np.pi * (2 * rng.random(circ.num_thetas) - 1)
AqcOptimizer(optimizer_name="lbfgs", maxiter=3).optimize(objv, circ, th0)
for rep in range(3):
    t0 = time.perf_counter()
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=40).optimize(objv, circ, th0)
    dt = time.perf_counter() - t0
    print(f"pairs {res['num_fun_ev']}  {dt * 1e3:.3f} ms  {dt / res['num_fun_ev'] * 1e6:.1f} us per pair")
# bare evaluation pairs on the object, no optimizer
t0 = time.perf_counter()
for i in range(200):
    th = th0 + 1e-3 * i
    objv.objective(th); objv.gradient(th)
print(f"bare objective()+gradient(): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per pair")
ws = objv._ws
t0 = time.perf_counter()
for i in range(200):
    ws.eval(th0 + 1e-3 * i, vdag=True, gather=True, grad=True, x_buf=0, block_range=(0, L), front_layer=True)
print(f"bare ws.eval: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
res = AqcOptimizer(optimizer_name="lbfgs", maxiter=40).optimize(objv, circ, th0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
