#!/usr/bin/env python3
"""Random cross-check of the lockstep lanes against the single-lane engine: random registers, entanglers, block layouts (long-range pairs
included), targets and lhs states of random bonds, thresholds and bond caps.  Usage: python tools/lockstep_fuzz.py [cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd import ParametricCircuit                      # noqa: E402
from aqc_research_amd import mps_engine as me                       # noqa: E402
from oracle import aqc_oracle as orc                                # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for case in range(cases):
    n = int(rng.integers(3, 13))
    ent = ("cx", "cz", "cp")[int(rng.integers(0, 3))]
    nb = int(rng.integers(1, 3 * n))
    ctrl = rng.integers(0, n, nb)
    targ = (ctrl + rng.integers(1, n, nb)) % n
    circ = ParametricCircuit(n, entangler=ent, blocks=np.stack([ctrl, targ]))
    lanes = int(rng.integers(1, 5))
    thr = (0.0, 1e-10, 1e-6, 1e-3)[int(rng.integers(0, 4))]
    cap = (0, 0, 3, 8, 16)[int(rng.integers(0, 5))]
    ths = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
    targets = [me.DeviceMPS.from_qiskit(orc.random_mps(n, int(rng.integers(1, 9)), rng), trunc_thr=thr) for _ in range(lanes)]
    lhs = [me.DeviceMPS.from_qiskit(orc.random_mps(n, int(rng.integers(1, 4)), rng), trunc_thr=thr) for _ in range(lanes)]
    lo = int(rng.integers(0, nb))
    br = None if rng.random() < 0.5 else (lo, int(rng.integers(lo, nb + 1)))
    front = bool(rng.random() < 0.7)
    kw = dict(trunc_thr=thr, max_bond=cap, block_range=br, front_layer=front)
    try:
        h, g = me.evaluate_lanes(circ, ths, targets, lhs, method="lockstep", **kw)
    except RuntimeError as err:
        print(f"case {case}: n={n} {ent} blocks={nb} lanes={lanes} thr={thr:g} cap={cap}: lockstep refused ({str(err)[:70]})", flush=True)
        continue
    hs, gs = me.evaluate_lanes(circ, ths, targets, lhs, method="threads", **kw)
    err = max(np.abs(h - hs).max(), np.abs(g - gs).max())
    worst = max(worst, err)
    flag = "" if err < 1e-9 else "   <-- MISMATCH"
    print(f"case {case}: n={n} {ent} blocks={nb} lanes={lanes} thr={thr:g} cap={cap} range={br} front={front}: max |diff| {err:.2e}{flag}", flush=True)
    for m in targets + lhs:
        m.close()
print(f"worst difference over {cases} cases: {worst:.2e}", flush=True)
