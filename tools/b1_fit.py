"""B=1 cost model of the per-group kernels: time per pass vs number of gate groups -> fixed cost per launch + cost per group."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, K_APPLY, K_SWEEP, HipContext, Workspace
n = 16
rng = np.random.default_rng(0)
for L in (0, 15, 30, 60, 120):
    blocks = create_ansatz_structure(n, "spin", "full", L) if L else np.zeros((2, 0), dtype=np.int64)
    circ = ParametricCircuit(n, "cx", blocks)
    ws = Workspace(HipContext.of(circ), batch=1)
    y = rng.random(1 << n) + 1j * rng.random(1 << n)
    ws.upload(BUF_Y, y / np.linalg.norm(y)); ws.set_basis(BUF_X, 0)
    ws.set_thetas(np.pi * (2 * rng.random(circ.num_thetas) - 1))
    for _ in range(20):
        ws.apply(True, BUF_Y, BUF_Z); ws.grad(None, True)
    ws.sync(); ws.profile(True)
    for _ in range(50):
        ws.apply(True, BUF_Y, BUF_Z); ws.grad(None, True)
    ws.sync()
    a, s = ws.profile_get(K_APPLY), ws.profile_get(K_SWEEP)
    ws.profile(False)
    print(f"L={L:3d} groups={n + L:3d}: apply {a[1] / 50 * 1e3:7.1f} us in {a[0] // 50} launches, sweep {s[1] / 50 * 1e3:7.1f} us in {s[0] // 50} launches", flush=True)
    ws.close()
