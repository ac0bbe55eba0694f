#!/usr/bin/env python3
"""Single evaluations (batch 1: thetas from the host, amplitudes + gradient back) of the headline shape, for a kernel trace
(rocprofv3 --kernel-trace ... -- python3 tools/lat_run.py) and tools/trace_timeline.py; prints the host-side latency."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace  # noqa: E402

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 40
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
rng = np.random.default_rng(0)
ws = Workspace(HipContext.of(circ), batch=1)
tg = rng.random(1 << n) + 1j * rng.random(1 << n)
ws.upload(BUF_Y, tg / np.linalg.norm(tg))
ws.set_basis(BUF_X, 0)
idx = np.array([0] + [1 << q for q in range(n)], dtype=np.int64)
ws.gather_setup(idx)
ths = np.pi * (2 * rng.random((300, circ.num_thetas)) - 1)
for i in range(20):
    ws.eval(ths[i], vdag=True, gather=True, grad=True)
time.sleep(0.5)
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for i in range(20, 70):
        ws.eval(ths[i], vdag=True, gather=True, grad=True)
    best = min(best, (time.perf_counter() - t0) / 50)
print(f"latency {best * 1e6:.1f} us per evaluation (host view), n = {n}", flush=True)
ws.close()
