#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: the headline shape (16 qubits, 40 blocks, bench.py's default number of lanes),
four evaluations as the bench's timed loop issues them -- thetas from a resident bank, aqc_ws_objective_launch (V^H where the
evaluation reads it, flip-state amplitudes, sparse-lhs sweep).  AQC_SPARSE_SWEEP=0 gives the dense route of the same workload."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace  # noqa: E402

n, L, B = 16, 40, int(os.environ.get("AQC_PROF_BATCH", "1024"))   # bench.py's default lanes per GPU at this size
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
ctx = HipContext.of(circ)
rng = np.random.default_rng(0)
ws = Workspace(ctx, batch=B)
tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True))
ws.set_basis(BUF_X, 0)
ws.gather_setup([0] + [1 << q for q in range(n)])
ws.theta_bank(np.pi * (2 * rng.random((4, B, circ.num_thetas)) - 1))
for i in range(4):
    ws.use_theta_set(i)
    ws.objective_launch(BUF_X)
ws.sync()
print("sparse counts (sweep items, cleared, V^H items):", ws.sparse_counts())
ws.close()
