#!/usr/bin/env python3
"""Attribution of the V^H stage kernel's time: the launch pair of the headline V^H (16 qubits, 40 blocks) timed with the library
named by AQC_HIP_LIB -- the shipped one, or a variant with parts of the sub-stage loop compiled out (AQC_EXP_APPLY_SKIP bits:
1 LDS writes, 2 LDS reads, 8 MFMAs, 16 fp64 adds, 32 address XORs; tools/apply_budget.sh builds the table).  Variant results are
garbage by construction; only times matter.  Usage: python tools/apply_budget.py [lanes] [label]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure  # noqa: E402
from aqc_research_amd.engine import BUF_Y, BUF_Z, K_APPLY, HipContext, Workspace  # noqa: E402

n, L = 16, 40
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
label = sys.argv[2] if len(sys.argv) > 2 else "base"
names = {"base": "everything (shipped library)", "8": "no MFMAs", "16": "no fp64 adds", "32": "no address XORs", "1": "no LDS writes",
         "2": "no LDS reads", "3": "no LDS traffic", "24": "no MFMAs, no adds", "48": "no adds, no XORs",
         "56": "no MFMAs, adds, XORs (LDS traffic + frame)", "27": "XORs + frame only", "59": "frame only (loads, stores, barriers, waits)"}
circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", L))
rng = np.random.default_rng(0)
ws = Workspace(HipContext.of(circ), batch=B)
tg = rng.random((B, 1 << n)) + 1j * rng.random((B, 1 << n))
ws.upload(BUF_Y, tg / np.linalg.norm(tg, axis=1, keepdims=True))
ws.set_thetas(np.pi * (2 * rng.random((B, circ.num_thetas)) - 1))
for _ in range(20):
    ws.apply(True, BUF_Y, BUF_Z)
ws.sync()
best = None
for rnd in range(3):
    ws.profile(True)
    for _ in range(20):
        ws.apply(True, BUF_Y, BUF_Z)
    ws.sync()
    launches, ms = ws.profile_get(K_APPLY)
    ws.profile(False)
    best = ms / 20 if best is None else min(best, ms / 20)
print(f"skip {label:>4s}  {names.get(label, ''):46s} {best * 1e3:8.1f} us per V^H ({launches // 20} launches, {B} lanes)", flush=True)
ws.close()
