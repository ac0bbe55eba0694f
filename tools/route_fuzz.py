#!/usr/bin/env python3
"""Random shapes through the projected routes of the sparse-lhs sweep (csrc/aqc_ws_project.cpp) against the full-size stages
(AQC_PROJECTED=0) of the same build: a sweep from one or two random basis states per lane (projection of the checkpoint) and a
one-call evaluation from one random basis state per lane with a gather set of in-tile indices and flips above the first stage's
bits (objective by projection).  Needs a GPU.  python tools/route_fuzz.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aqc_research_amd import ParametricCircuit, TrotterAnsatz  # noqa: E402
from aqc_research_amd.circuit_structures import create_ansatz_structure, make_trotter_like_circuit  # noqa: E402
from aqc_research_amd.engine import BUF_X, BUF_X2, BUF_Y, BUF_Z, HipContext, Workspace  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
os.environ["AQC_SPARSE_MIN_ITEMS"] = "1"
os.environ["AQC_PROJECTED_VDAG_MIN_ELEMS"] = "1"
worst, ran, kinds = 0.0, 0, {}
for case in range(cases):
    n = int(rng.integers(12, 18))
    tile = int(rng.integers(8, 13))
    if rng.random() < 0.25 and n >= 13:
        layers = int(rng.integers(1, 3))
        circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
        what = f"trotter {layers}"
    else:
        blocks = int(rng.integers(4, 64))
        ent = ["cx", "cz", "cp"][int(rng.integers(0, 3))]
        circ = ParametricCircuit(n, ent, create_ansatz_structure(n, "spin", "full", blocks))
        what = f"{ent} {blocks} blocks"
    os.environ["AQC_PROJECTED"] = "1"
    info = HipContext(circ).plan_projected(tile)
    if not info:
        continue
    B = int(rng.integers(1, 6))
    T = circ.num_thetas
    th = np.pi * (2 * rng.random((2, B, T)) - 1)
    tg = rng.standard_normal((B, 1 << n)) + 1j * rng.standard_normal((B, 1 << n))
    tg /= np.linalg.norm(tg, axis=1, keepdims=True)
    idx = rng.integers(0, 1 << n, size=(B, 2))
    idx[rng.random(B) < 0.4, 1] = -1
    coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
    basis = rng.integers(0, 1 << n, size=B)
    lo = basis[0] & ((1 << tile) - 1)
    basis = (basis & ~((1 << tile) - 1)) | lo          # one index on the first stage's bits (its tile bits are the low ones), any above
    hi = n - tile
    above = sorted({int(x) for x in rng.integers(0, 1 << hi, size=4)} | {0})
    gather = np.array([lo | (f << tile) for f in above], dtype=np.int64)
    out = {}
    for proj in ("1", "0"):
        os.environ["AQC_PROJECTED"] = proj
        ws = Workspace(HipContext(circ), batch=B, tile_bits_apply=tile, tile_bits_sweep=tile)
        if proj == "1" and not ws.projected_info():
            ws.close()
            out = None
            break
        ws.upload(BUF_Y, tg)
        ws.set_thetas(th[0])
        ws.apply(True, BUF_Y, BUF_Z)
        ws.set_combo(BUF_X2, idx, coef)
        ws.grad_from(BUF_X2)
        g1 = ws.get_grads()
        ws.set_basis(BUF_X, basis)
        ws.gather_setup(gather)
        ws.set_thetas(th[1])
        ws.objective_launch(BUF_X)
        out[proj] = (g1, ws.gather_fetch().copy(), ws.get_grads().copy())
        ws.close()
    if out is None:
        continue
    err = max(float(np.abs(a - b).max()) for a, b in zip(out["1"], out["0"]))
    worst = max(worst, err)
    ran += 1
    key = (info["shared_with_first_stage"], info["stages"])
    kinds[key] = kinds.get(key, 0) + 1
    flag = "" if err < 1e-12 else "   <-- DIFFERS"
    print(f"case {case}: n={n} tile=2^{tile} {what} lanes={B} virtual={info['virtual_qubits']} (shared {info['shared_with_first_stage']}, "
          f"{info['stages']} stage(s)): max |projected - full size| = {err:.2e}{flag}", flush=True)
print(f"seed {seed}: {ran} shapes with a projected route of {cases} drawn, worst difference {worst:.2e}; (shared qubits, virtual stages) -> cases: {dict(sorted(kinds.items()))}")
sys.exit(0 if worst < 1e-12 else 1)
