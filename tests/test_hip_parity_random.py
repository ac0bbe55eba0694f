"""GPU parity, randomised: seeded random circuits / shapes / tilings / ranges on both kernel families against
the compiled CPU restatement (oracle/aqc_ref.c).  Vectors and rectangular matrices, all entanglers, Trotter
decorations, partial block ranges, several lanes."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from oracle import aqc_ref as cref
from tests.helpers import FAMILIES, FAMILY_ENV, TOL, maxdiff

pytestmark = pytest.mark.gpu


def _random_case(seed):
    rng = np.random.default_rng(seed)
    kind = ["generic", "generic", "spin", "trotter1", "trotter2"][seed % 5]
    ncols = [1, 1, 1, 3, 8, 5][seed % 6]
    if ncols > 1 and kind.startswith("trotter"):
        kind = "generic"                       # the matrix path has no Trotter ansatz
    n = int(rng.integers(2, 12 if ncols == 1 else 9))
    if kind == "generic":
        ent = ["cx", "cz", "cp"][int(rng.integers(3))]
        depth = int(rng.integers(1, 30))
        blocks = np.stack([rng.permutation(n)[:2] for _ in range(depth)], axis=1).astype(np.int64)
        a = orc.Ansatz(n, ent, blocks)
    elif kind == "spin":
        a = orc.Ansatz(n, ["cx", "cz", "cp"][int(rng.integers(3))], orc.spin_blocks(n, int(rng.integers(1, 40))))
    else:
        a = orc.Ansatz(n, "cx", orc.trotter_blocks(n, int(rng.integers(1, 3))), True, kind == "trotter2")
    batch = int(rng.integers(1, 4))
    nbits = n + int(np.ceil(np.log2(ncols)))
    ka = int(rng.integers(2, min(nbits, 13) + 1))
    ks = int(rng.integers(2, min(nbits, 12) + 1))
    L = a.num_blocks
    br = None
    front = True
    if ncols == 1 and L > 1 and seed % 3 == 0:
        lo = int(rng.integers(0, L - 1))
        br = (lo, int(rng.integers(lo + 1, L + 1)))
        front = bool(rng.integers(2))
    return rng, a, ncols, batch, ka, ks, br, front


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("seed", range(30))
def test_random_configuration(seed, family, monkeypatch):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])
    rng, a, ncols, batch, ka, ks, br, front = _random_case(seed)
    circ = TrotterAnsatz(a.n, a.blocks, second_order=a.second_order) if a.trotter else ParametricCircuit(a.n, a.entangler, a.blocks)
    ws = Workspace(HipContext.of(circ), batch=batch, ncols=ncols, tile_bits_apply=ka, tile_bits_sweep=ks)
    shape = (batch, a.dim) if ncols == 1 else (batch, a.dim, ncols)
    th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(batch)])
    x = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    y = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    # unit Frobenius norm per lane: every inner product is then O(1) and the absolute tolerance below IS 1e-10
    x /= np.sqrt((np.abs(x.reshape(batch, -1)) ** 2).sum(axis=1)).reshape((batch,) + (1,) * (x.ndim - 1))
    y /= np.sqrt((np.abs(y.reshape(batch, -1)) ** 2).sum(axis=1)).reshape((batch,) + (1,) * (y.ndim - 1))
    ws.set_thetas(th)
    ws.upload(BUF_X, x)
    ws.upload(BUF_Y, y)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)
    ws.grad(br, front)
    g = ws.get_grads()
    ws.apply(False, BUF_X, BUF_Y)            # V x
    vx = ws.download(BUF_Y)
    for b in range(batch):
        if ncols == 1:
            zr = cref.v_dagger_mul_vec(a, th[b], y[b])
            gr = cref.grad_of_dot_product(a, th[b], x[b], zr, br, front)
            vr = cref.v_mul_vec(a, th[b], x[b])
        else:
            zr = cref.v_dagger_mul_mat(a, th[b], y[b])
            gr = cref.grad_of_matrix_dot_product(a, th[b], x[b], zr)
            vr = cref.v_mul_mat(a, th[b], x[b])
        assert maxdiff(z[b], zr) < TOL and maxdiff(vx[b], vr) < TOL
        assert maxdiff(g[b], gr) < TOL
    ws.close()
