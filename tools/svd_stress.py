#!/usr/bin/env python3
"""Stress of the engine's Jacobi SVD on the matrices MPS gates really produce: exactly rank-deficient, graded, duplicated
columns, blocks of zeros.  Reports sweeps used / failures."""
import sys

import numpy as np

sys.path.insert(0, ".")
from aqc_research_amd.mps_engine import svd   # noqa: E402

rng = np.random.default_rng(0)
fails, worst = 0, 0
cases = []
for trial in range(400):
    m = int(rng.choice([8, 16, 26, 26, 32, 52, 64]))
    kind = trial % 6
    a = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
    if kind == 0:    # exact low rank
        r = int(rng.integers(1, m))
        a = (rng.standard_normal((m, r)) + 1j * rng.standard_normal((m, r))) @ (rng.standard_normal((r, m)) + 1j * rng.standard_normal((r, m)))
    elif kind == 1:  # graded singular values down to 1e-20
        u, _ = np.linalg.qr(a)
        v, _ = np.linalg.qr(rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
        a = (u * np.logspace(0, -20, m)) @ v
    elif kind == 2:  # duplicated and zero columns
        a[:, 1] = a[:, 0]
        a[:, 2] = 0
        a[:, 3] = a[:, 0] * 1j
    elif kind == 3:  # block structure with zero blocks (two-site tensors of product-like states)
        a[: m // 2, m // 2:] = 0
        a[m // 2:, : m // 2] = 0
        a[:, m // 2:] *= 1e-9
    elif kind == 4:  # low rank + noise at 1e-17
        r = int(rng.integers(1, m // 2 + 1))
        a = (rng.standard_normal((m, r)) + 1j * rng.standard_normal((m, r))) @ (rng.standard_normal((r, m)) + 1j * rng.standard_normal((r, m)))
        a = a + 1e-17 * (rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m)))
    else:            # unitary times sparse diagonal
        u, _ = np.linalg.qr(a)
        d = np.zeros(m); d[: max(1, m // 4)] = rng.random(max(1, m // 4))
        a = u * d
    try:
        u, s, vh, sweeps = svd(a)
        worst = max(worst, sweeps)
        err = np.abs((u * s) @ vh - a).max() / max(1.0, s[0])
        if err > 1e-11 or sweeps > 30:
            print(f"trial {trial} kind {kind} m {m}: sweeps {sweeps} err {err:.2e}", flush=True)
    except RuntimeError as exc:
        fails += 1
        print(f"trial {trial} kind {kind} m {m}: FAILED {exc}", flush=True)
print(f"failures {fails} of 400, worst sweeps {worst}")
