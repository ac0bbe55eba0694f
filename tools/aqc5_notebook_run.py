#!/usr/bin/env python3
"""The runs of the reference's AQC tutorial (docs/aqc.ipynb: 5 qubits, cyclic_spin ansatz of 180 blocks, 735 parameters, Haar-random SU(32)
target): full AQC by L-BFGS on the matrix objective (the notebook: 224.35 s, fidelity 0.9705), sketched AQC by ADAM on 16 sketching vectors
(128.53 s, 0.9535), 1000 coordinate-descent sweeps (273.73 s, 0.9647) -- here through AqcOptimizer / SketchingObjectiveEx + FullRangeSketchingVectors and core_op_matrix.coord_descent_sweeps, one lane and
64 random restarts at once.  Usage: python tools/aqc5_notebook_run.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import ParametricCircuit                                   # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure          # noqa: E402
from aqc_research_amd.core_op_matrix import coord_descent_sweeps                 # noqa: E402
from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx, skvecs_generator   # noqa: E402
from aqc_research_amd.optimizer import AqcOptimizer                              # noqa: E402

n, L = 5, 180
d = 1 << n
rng = np.random.default_rng(2024)
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "cyclic_spin", "full", L))
q, r = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))
u = q * (np.diag(r) / np.abs(np.diag(r)))            # Haar-random unitary ...
u = u / np.linalg.det(u) ** (1.0 / d)                # ... made special (target_generator.py:269-288)
th0 = np.pi * (2.0 * rng.random(circ.num_thetas) - 1.0)


# full AQC: L-BFGS on the matrix objective (X = I: all d columns)
objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(u.copy()))
objv.objective(th0); objv.gradient(th0)              # warm-up (plans, library)
objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(u.copy()))
t0 = time.perf_counter()
res = AqcOptimizer(optimizer_name="lbfgs", maxiter=1000).optimize(objv, circ, th0.copy())
t_lbfgs = time.perf_counter() - t0
print(f"full AQC, L-BFGS: {res['num_iters']} iterations, {res['num_fun_ev']} objective+gradient evaluations in {t_lbfgs:.2f} s; cost 1 - Re<V,U>/d = "
      f"{res['cost']:.5f}, i.e. fidelity |<V,U>|^2 / d^2 >= {(1.0 - res['cost']) ** 2:.4f}", flush=True)

# sketched AQC: ADAM on 16 random sketching vectors, redrawn every evaluation (the notebook: ~690 iterations in 128.53 s, fidelity 0.9535)
np.random.seed(7)
sk = SketchingObjectiveEx(circ, skvecs_generator("rand", 16, u.copy()))
t0 = time.perf_counter()
res_sk = AqcOptimizer(optimizer_name="adam", maxiter=690, learn_rate=0.05).optimize(sk, circ, th0.copy())
t_sk = time.perf_counter() - t0
full = SketchingObjectiveEx(circ, FullRangeSketchingVectors(u.copy()))
print(f"sketched AQC, ADAM, 16 random sketching vectors: {res_sk['num_iters']} iterations in {t_sk:.2f} s; full-matrix cost at the result "
      f"{full.objective(res_sk['thetas']):.5f}", flush=True)

# coordinate descent: 1000 sweeps, one lane, then 64 restarts at once
th = th0[None, :].copy()
coord_descent_sweeps(circ, th.copy(), u, 2)          # warm-up
t0 = time.perf_counter()
f1 = coord_descent_sweeps(circ, th, u, 1000)
t_cd1 = time.perf_counter() - t0
ths = np.pi * (2.0 * rng.random((64, circ.num_thetas)) - 1.0)
t0 = time.perf_counter()
f64 = coord_descent_sweeps(circ, ths, u, 1000)
t_cd64 = time.perf_counter() - t0
print(f"coordinate descent, 1000 sweeps: one lane {t_cd1:.2f} s (fidelity {1.0 - f1[0, -1]:.4f}); 64 restarts at once {t_cd64:.2f} s "
      f"(fidelities {1.0 - f64[:, -1].max():.4f} ... {1.0 - f64[:, -1].min():.4f})", flush=True)
