"""How long does a fresh GPU take to reach its steady clock?  Per-chunk time of the bench step from process start."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace
n, L, B = 16, 40, 64
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
ws = Workspace(HipContext.of(circ), batch=B)
rng = np.random.default_rng(0)
tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True)); ws.set_basis(BUF_X, 0); ws.gather_setup(np.arange(n + 1))
ws.theta_bank(np.pi * (2 * rng.random((4, B, circ.num_thetas)) - 1))
t_start = time.perf_counter()
for chunk in range(int(os.environ.get("CHUNKS", 40))):
    ws.sync(); t0 = time.perf_counter()
    for i in range(25):
        ws.use_theta_set(i % 4); ws.apply(True, BUF_Y, BUF_Z); ws.gather_launch(BUF_Z); ws.grad(None, True)
    ws.sync(); t1 = time.perf_counter()
    print(f"t={t1 - t_start:6.3f}s  {1e3 * (t1 - t0) / 25:.3f} ms/step", flush=True)
