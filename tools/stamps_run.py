#!/usr/bin/env python3
"""In-kernel stamps of the matrix-core sweep (tuning build only): AQC_HIP_LIB=aqc_research_amd/libaqc_hip_tuning.so
AQC_STAMPS=1 python tools/stamps_run.py [lanes].  Prints, per sweep launch, the per-phase cycle split and the workgroup
lifetimes next to the launch duration measured with HIP events, so that ticks can be turned into time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, K_APPLY, K_SWEEP, HipContext, Workspace  # noqa: E402

n, L = 16, 40
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
ctx = HipContext.of(circ)
rng = np.random.default_rng(0)
ws = Workspace(ctx, batch=B)
tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True))
ws.set_basis(BUF_X, 0)
ws.profile(True)
for i in range(3):
    ws.set_thetas(np.pi * (2 * rng.random((B, circ.num_thetas)) - 1))
    ws.apply(True, BUF_Y, BUF_Z)
    ws.grad()
    ws.sync()
    a, s = ws.profile_get(K_APPLY), ws.profile_get(K_SWEEP)
    print(f"step {i}: cumulative apply {a}, sweep {s} (launches, ms)", file=sys.stderr)
ws.close()
