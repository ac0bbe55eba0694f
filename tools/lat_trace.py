"""Batch-1 evaluations of the headline ansatz for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/lat_trace.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aqc_research_amd import ParametricCircuit
from aqc_research_amd.circuit_structures import create_ansatz_structure
from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace
from oracle import aqc_oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
rng = np.random.default_rng(0)
ws = Workspace(HipContext.of(circ), batch=1)
ws.upload(BUF_Y, orc.rand_state(n, rng)); ws.set_basis(BUF_X, 0); ws.gather_setup(orc.flip_state_indices(n, 1))
ths = np.pi * (2 * rng.random((300, circ.num_thetas)) - 1)
for i in range(20):
    ws.eval(ths[i], vdag=True, gather=True, grad=True)
time.sleep(1.0)
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for i in range(20, 300):
        ws.eval(ths[i], vdag=True, gather=True, grad=True)
    best = min(best, (time.perf_counter() - t0) / 280 * 1e3)
print(f"n={n}: {best:.4f} ms per evaluation; tiles {ws.plan_info(0)}/{ws.plan_info(1)}")
# the same through the raw ABI with prebuilt pointers: what the Python wrapper itself costs
from aqc_research_amd import _lib
L = _lib.lib()
hs = np.empty((1, ws._gather_count), dtype=np.complex128); g = np.empty((1, circ.num_thetas), dtype=np.complex128)
ptrs = [_lib.dptr(np.ascontiguousarray(ths[i])) for i in range(300)]
keep = [np.ascontiguousarray(ths[i]) for i in range(300)]
ptrs = [_lib.dptr(k) for k in keep]
phs, pg = _lib.dptr(hs), _lib.dptr(g)
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for i in range(20, 300):
        L.aqc_ws_eval(ws.handle, ptrs[i], 1, phs, BUF_X, -1, -1, 1, pg)
    best = min(best, (time.perf_counter() - t0) / 280 * 1e3)
print(f"n={n}: {best:.4f} ms per evaluation through the raw ABI")
