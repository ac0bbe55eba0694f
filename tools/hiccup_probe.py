"""Sporadic slow rounds of single-lane evaluations seen inside bench.py: which ingredient brings them?"""
import os, sys, time
T0 = time.perf_counter()
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace
mode = sys.argv[1]
n, L = 16, 40
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
ctx = HipContext.of(circ)
rng = np.random.default_rng(0)
T = circ.num_thetas
big = None
if mode in ("big", "bigrun", "bigprof"):
    B = 64
    big = Workspace(ctx, batch=B)
    tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
    big.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True)); big.set_basis(BUF_X, 0); big.gather_setup(np.arange(n + 1))
    big.theta_bank(np.pi * (2 * rng.random((4, B, T)) - 1))
    if mode in ("bigrun", "bigprof"):
        if mode == "bigprof":
            big.profile(True)
        for i in range(300):
            big.use_theta_set(i % 4); big.apply(True, BUF_Y, BUF_Z); big.gather_launch(BUF_Z); big.grad(None, True)
        big.sync()
        if mode == "bigprof":
            big.profile(False)
ws1 = Workspace(ctx, batch=1)
y = rng.random(1 << n) + 1j * rng.random(1 << n)
ws1.upload(BUF_Y, y / np.linalg.norm(y)); ws1.set_basis(BUF_X, 0); ws1.gather_setup(np.arange(n + 1))
ths = np.pi * (2 * rng.random((60, T)) - 1)
for i in range(10):
    ws1.eval(ths[i], vdag=True, gather=True, grad=True)
if len(sys.argv) > 2:
    time.sleep(float(sys.argv[2]))
rounds = []
stamps = []
for r in range(30):
    t1 = time.perf_counter()
    stamps.append(t1 - T0)
    for i in range(10, 60):
        ws1.eval(ths[i], vdag=True, gather=True, grad=True)
    rounds.append((time.perf_counter() - t1) / 50 * 1e3)
print(mode, "min %.3f median %.3f max %.3f" % (min(rounds), sorted(rounds)[15], max(rounds)), "slow rounds:", [(i, round(x, 2), "t=%.2fs" % stamps[i]) for i, x in enumerate(rounds) if x > 1.5 * min(rounds)], "loop start t=%.2fs" % stamps[0])
