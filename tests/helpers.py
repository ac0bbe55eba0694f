"""Shared helpers for the test-suite (golden loaders, case builders)."""
import os

import numpy as np

from oracle import aqc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-10  # north-star tolerance (complex fp64, absolute)
# kernel families of the HIP path (AQC_KERNEL_FAMILY): every parity test runs on all of them
FAMILIES = ["per-group", "register-blocked", "mfma"]
FAMILY_ENV = {"per-group": "1", "register-blocked": "2", "mfma": "3"}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def ansatz_from(d, key):
    return orc.Ansatz(
        int(d[f"{key}/n"]),
        str(d[f"{key}/ent"]),
        d[f"{key}/blocks"].astype(np.int64),
        bool(d[f"{key}/trotter"]),
        bool(d[f"{key}/second_order"]),
    )


def mps_from(d, key, tag):
    n = int(d[f"{key}/n"])
    gam = [(d[f"{key}/{tag}_g0_{q}"], d[f"{key}/{tag}_g1_{q}"]) for q in range(n)]
    lam = [d[f"{key}/{tag}_lam_{q}"] for q in range(n - 1)]
    return gam, lam


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)))) if np.size(a) else 0.0


def free_port() -> int:
    """A TCP port that is free right now (rendezvous of the 2-rank tests; avoids collisions between parallel runs)."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def canonical_mps(vec, cap):
    """Canonical Vidal form (Gamma, lambda) of a dense state by successive SVDs -- what Aer hands the reference
    (mps_operations.py:216-243) -- with the bonds capped at `cap` and the kept Schmidt values renormalised."""
    nq = int(np.log2(vec.size))
    rest = vec.reshape([2] * nq).transpose(list(range(nq - 1, -1, -1))).reshape(1, -1)   # axes (b_0, ..., b_{n-1})
    gam, lam, prev = [], [], np.ones(1)
    for _ in range(nq - 1):
        chi_l = rest.shape[0]
        u, s, vh = np.linalg.svd(rest.reshape(chi_l * 2, -1), full_matrices=False)
        k = min(cap, int((s > 1e-14 * s[0]).sum()))
        u, s, vh = u[:, :k], s[:k] / np.linalg.norm(s[:k]), vh[:k]
        a = u.reshape(chi_l, 2, k)
        gam.append((a[:, 0, :] / prev[:, None], a[:, 1, :] / prev[:, None]))
        lam.append(s.copy())
        prev = s
        rest = s[:, None] * vh
    a = rest.reshape(rest.shape[0], 2, 1)
    gam.append((a[:, 0, :] / prev[:, None], a[:, 1, :] / prev[:, None]))
    return gam, lam
