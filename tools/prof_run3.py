#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes on one kernel family (AQC_KERNEL_FAMILY): the headline shape
(16 qubits, 40 blocks, bench.py's default number of lanes), three V^H + sweep steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace  # noqa: E402

n, L, B = 16, 40, int(os.environ.get("AQC_PROF_BATCH", "1024"))   # bench.py's default lanes per GPU at this size
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
ctx = HipContext.of(circ)
rng = np.random.default_rng(0)
ws = Workspace(ctx, batch=B)
tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True))
ws.set_basis(BUF_X, 0)
for i in range(3):
    ws.set_thetas(np.pi * (2 * rng.random((B, circ.num_thetas)) - 1))
    ws.apply(True, BUF_Y, BUF_Z)
    ws.grad()
ws.sync()
ws.close()
