"""GPU parity: HIP state-vector path vs golden vectors (reference outputs) and vs the CPU oracle."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import FAMILIES, FAMILY_ENV, TOL, ansatz_from, load, maxdiff

pytestmark = pytest.mark.gpu

SV = load("state_vector.npz")


def make_circ(a: orc.Ansatz):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz

    if a.trotter:
        return TrotterAnsatz(a.n, a.blocks, second_order=a.second_order)
    return ParametricCircuit(a.n, a.entangler, a.blocks)


@pytest.mark.parametrize("key", [str(k) for k in SV["names"]])
def test_golden_function_level(key):
    import aqc_research_amd.core_operations as cop

    a = ansatz_from(SV, key)
    circ = make_circ(a)
    th, x, y = SV[f"{key}/thetas"], SV[f"{key}/x"].copy(), SV[f"{key}/y"].copy()
    dim = 1 << a.n
    ws = np.zeros((3, dim), np.complex128)
    out = np.zeros(dim, np.complex128)
    res = cop.v_mul_vec(circ, th, x, out, ws[:2])
    assert res is out and maxdiff(out, SV[f"{key}/v_x"]) < TOL
    vhy = cop.v_dagger_mul_vec(circ, th, y, np.zeros(dim, np.complex128), ws[:2])
    assert maxdiff(vhy, SV[f"{key}/vh_y"]) < TOL
    x0, v0 = x.copy(), vhy.copy()
    g = cop.grad_of_dot_product(circ, th, x, vhy, ws)
    assert g.dtype == np.complex128 and g.shape == (a.num_thetas,)
    assert maxdiff(g, SV[f"{key}/grad_full"]) < TOL
    assert np.array_equal(x, x0) and np.array_equal(vhy, v0)  # inputs intact (core_operations.py:892-893)
    br = tuple(int(v) for v in SV[f"{key}/block_range"])
    gp = cop.grad_of_dot_product(circ, th, x, vhy, ws, block_range=br, front_layer=False)
    assert maxdiff(gp, SV[f"{key}/grad_part"]) < TOL
    # out aliasing vec (test_core_operations.py:270)
    buf = y.copy()
    cop.v_dagger_mul_vec(circ, th, buf, buf, ws[:2])
    assert maxdiff(buf, SV[f"{key}/vh_y"]) < TOL


CASES = [
    # n, entangler/kind, depth, tile_bits_apply, tile_bits_sweep, batch
    (7, "cx", 12, 4, 4, 1),
    (8, "cz", 15, 5, 6, 3),
    (9, "cp", 14, 6, 5, 2),
    (10, "cx", 30, 7, 7, 2),
    (11, "cp", 20, 11, 11, 1),
    (12, "cz", 25, 8, 9, 1),
    (13, "cx", 40, 13, 12, 2),
    (14, "cx", 40, 10, 10, 1),
    (14, "cz", 33, 11, 11, 3),      # 2^11 tiles, 8 tiles per lane: single-buffered scratch + second barrier
    (15, "cx", 30, 12, 11, 2),
    (16, "cp", 24, 11, 12, 5),      # persistent sweep next to the one-tile-per-workgroup V^H
]


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("n,ent,depth,ka,ks,batch", CASES)
def test_oracle_multistage_batched(n, ent, depth, ka, ks, batch, family, monkeypatch):
    """Random generic circuits; small tiles force many stages and many tiles; lanes carry
    different thetas / targets.  Both kernel families (aqc_kernels.hip / aqc_kernels2.hip)."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])

    rng = np.random.default_rng(100 * n + depth)
    blocks = np.stack([rng.permutation(n)[:2] for _ in range(depth)], axis=1).astype(np.int64)
    a = orc.Ansatz(n, ent, blocks)
    ctx = HipContext.of(make_circ(a))
    ws = Workspace(ctx, batch=batch, tile_bits_apply=ka, tile_bits_sweep=ks)
    th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(batch)])
    x = np.stack([orc.rand_state(n, rng) for _ in range(batch)])
    y = np.stack([orc.rand_state(n, rng) for _ in range(batch)])
    ws.set_thetas(th)
    ws.upload(BUF_Y, y)
    ws.upload(BUF_X, x)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)
    ws.grad(None, True)
    g = ws.get_grads()
    hs = ws.vdot(BUF_X, BUF_Z)
    for b in range(batch):
        zr = orc.v_dagger_mul_vec(a, th[b], y[b])
        assert maxdiff(z[b], zr) < TOL
        assert maxdiff(g[b], orc.grad_of_dot_product(a, th[b], x[b], zr)) < TOL
        assert abs(hs[b] - np.vdot(x[b], zr)) < TOL
    # forward apply and V V^H = I
    ws.apply(False, BUF_Z, BUF_Y)
    assert maxdiff(ws.download(BUF_Y), y) < TOL
    ws.close()


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("n,layers,order2", [(6, 2, True), (9, 2, True), (12, 2, True), (12, 1, False)])
def test_trotter_vs_oracle(n, layers, order2, family, monkeypatch):
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])

    rng = np.random.default_rng(7 * n + layers)
    a = orc.Ansatz(n, "cx", orc.trotter_blocks(n, layers), True, order2)
    ctx = HipContext.of(make_circ(a))
    for ka, ks in ((0, 0), (6, 5)):
        ws = Workspace(ctx, batch=1, tile_bits_apply=ka, tile_bits_sweep=ks)
        th, y = orc.rand_thetas(a.num_thetas, rng), orc.rand_state(n, rng)
        ws.set_thetas(th)
        ws.upload(BUF_Y, y)
        ws.apply(True, BUF_Y, BUF_Z)
        zr = orc.v_dagger_mul_vec(a, th, y)
        zd = ws.download(BUF_Z)[0]
        assert maxdiff(zd, zr) < TOL
        idx = int(rng.integers(0, 1 << n))
        ws.set_basis(BUF_X, idx)
        x = np.zeros(1 << n, complex); x[idx] = 1
        bpl = 3 * (n - 1)
        for br, front in ((None, True), ((0, bpl), False), ((bpl, 2 * bpl) if layers > 1 else (1, bpl - 1), False)):
            ws.grad(br, front)
            assert maxdiff(ws.get_grads()[0], orc.grad_of_dot_product(a, th, x, zr, br, front)) < TOL
        got = ws.gather(BUF_Z, [0, idx, (1 << n) - 1])[0]
        assert maxdiff(got, zd[[0, idx, (1 << n) - 1]]) == 0.0
        ws.close()


@pytest.mark.parametrize("family", FAMILIES)
def test_headline_size_properties(family, monkeypatch):
    """n=16, L=40 (BASELINE configs[2] geometry): parity vs the oracle plus size-independent
    properties (unitarity, linearity of the gradient in x)."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])

    n, L = 16, 40
    rng = np.random.default_rng(16)
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, L))
    ws = Workspace(HipContext.of(make_circ(a)), batch=2)
    th = np.stack([orc.rand_thetas(a.num_thetas, rng)] * 2)
    y = orc.rand_state(n, rng)
    x1, x2 = orc.rand_state(n, rng), orc.rand_state(n, rng)
    ws.set_thetas(th)
    ws.broadcast(BUF_Y, y)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)
    zr = orc.v_dagger_mul_vec(a, th[0], y)
    assert maxdiff(z[0], zr) < TOL and np.array_equal(z[0], z[1])
    assert abs(np.linalg.norm(z[0]) - 1) < 1e-12
    ws.upload(BUF_X, np.stack([x1, x2]))
    ws.grad()
    g = ws.get_grads()
    assert maxdiff(g[0], orc.grad_of_dot_product(a, th[0], x1, zr)) < TOL
    ws.upload(BUF_X, np.stack([x1 + 2j * x2, x1]))
    ws.grad()
    g2 = ws.get_grads()
    assert maxdiff(g2[0], g[0] - 2j * g[1]) < 1e-9  # <V x|y> is anti-linear in x
    assert maxdiff(g2[1], g[0]) == 0.0  # bitwise reproducible
    ws.close()


def test_fused_eval_matches_separate_calls():
    """aqc_ws_eval (one synchronisation) == set_thetas + apply + gather + grad."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.engine import BUF_X, BUF_X2, BUF_Y, BUF_Z, HipContext, Workspace

    n = 11
    rng = np.random.default_rng(77)
    a = orc.Ansatz(n, "cz", orc.spin_blocks(n, 25))
    ws = Workspace(HipContext.of(ParametricCircuit(n, "cz", a.blocks)), batch=2)
    th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(2)])
    y = np.stack([orc.rand_state(n, rng) for _ in range(2)])
    idx = orc.flip_state_indices(n, 1)
    ws.upload(BUF_Y, y)
    ws.set_basis(BUF_X, 0)
    ws.set_basis(BUF_X2, [5, 9])
    ws.gather_setup(idx)
    hs, g = ws.eval(th, vdag=True, gather=True, grad=True, block_range=(2, 20), front_layer=False)
    _, g2 = ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2)
    for b in range(2):
        zr = orc.v_dagger_mul_vec(a, th[b], y[b])
        assert maxdiff(hs[b], zr[idx]) < TOL
        x0 = np.zeros(1 << n, complex); x0[0] = 1
        assert maxdiff(g[b], orc.grad_of_dot_product(a, th[b], x0, zr, (2, 20), False)) < TOL
        x2 = np.zeros(1 << n, complex); x2[[5, 9][b]] = 1
        assert maxdiff(g2[b], orc.grad_of_dot_product(a, th[b], x2, zr)) < TOL
    ws.close()


def test_edge_cases_minimal_and_empty():
    """n = 2 (smallest system), zero blocks (front layer only), a single block, and the validation
    the reference does with asserts (core_operations.py:626-630,869-879)."""
    import aqc_research_amd.core_operations as cop
    from aqc_research_amd import ParametricCircuit

    rng = np.random.default_rng(2)
    for n, blocks in ((2, np.zeros((2, 0), dtype=np.int64)), (2, np.array([[1], [0]])), (5, np.zeros((2, 0), dtype=np.int64))):
        circ = ParametricCircuit(n, "cp", blocks)
        a = orc.as_ansatz(circ)
        th = orc.rand_thetas(circ.num_thetas, rng)
        x, y = orc.rand_state(n, rng), orc.rand_state(n, rng)
        out = np.zeros(1 << n, np.complex128)
        assert maxdiff(cop.v_mul_vec(circ, th, x, out, None), orc.v_mul_vec(a, th, x)) < TOL
        vhy = cop.v_dagger_mul_vec(circ, th, y, np.zeros(1 << n, np.complex128), None)
        assert maxdiff(vhy, orc.v_dagger_mul_vec(a, th, y)) < TOL
        if circ.num_blocks:
            assert maxdiff(cop.grad_of_dot_product(circ, th, x, vhy, None), orc.grad_of_dot_product(a, th, x, vhy)) < TOL
    circ = ParametricCircuit(3, "cx", np.array([[0, 1], [1, 2]]))
    th = np.zeros(circ.num_thetas)
    v = np.zeros(8, np.complex128)
    with pytest.raises(ValueError):
        cop.v_mul_vec(circ, th[:-1], v, v.copy(), None)            # wrong number of thetas
    with pytest.raises(ValueError):
        cop.v_mul_vec(circ, th, v[:4], v.copy(), None)             # wrong vector size
    with pytest.raises(ValueError):
        cop.v_mul_vec(circ, th, v.astype(np.complex64), v.copy(), None)
    with pytest.raises(ValueError):
        cop.grad_of_dot_product(circ, th, v, v.copy(), None, block_range=(1, 1))
    with pytest.raises(ValueError):
        cop.grad_of_dot_product(circ, th, v, v.copy(), None, block_range=(0, 3))
    ws = np.zeros((3, 8), np.complex128)
    with pytest.raises(ValueError):
        cop.v_mul_vec(circ, th, ws[0], v.copy(), ws)               # vec overlaps the workspace


def test_20_qubit_properties():
    """BASELINE configs[3] size (2^20 amplitudes, 2nd-order Trotter ansatz): size-independent properties
    instead of a full oracle run -- unitarity, V V^H = 1, gradient == central finite differences of the
    objective computed by the same path, repeatability."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    n = 20
    rng = np.random.default_rng(20)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 1), second_order=True)
    ws = Workspace(HipContext.of(circ), batch=1)
    th = 0.4 * orc.rand_thetas(circ.num_thetas, rng)
    y = orc.rand_state(n, rng)
    ws.upload(BUF_Y, y)
    ws.set_thetas(th)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)[0]
    assert abs(np.linalg.norm(z) - 1) < 1e-12
    ws.set_basis(BUF_X, 0)
    ws.grad()
    g = ws.get_grads()[0]
    ws.grad()
    assert np.array_equal(g, ws.get_grads()[0])          # bit-reproducible
    for t in rng.choice(circ.num_thetas, 6, replace=False):  # <V e0|y> = (V^H y)[0]
        f = []
        for sgn in (+1, -1):
            e = np.zeros_like(th); e[t] = sgn * 1e-5
            ws.set_thetas(th + e)
            ws.apply(True, BUF_Y, BUF_Z)
            f.append(ws.gather(BUF_Z, [0])[0, 0])
        assert abs((f[0] - f[1]) / 2e-5 - g[t]) < 1e-8
    ws.set_thetas(th)
    ws.apply(True, BUF_Y, BUF_Z)
    ws.apply(False, BUF_Z, BUF_X)
    assert maxdiff(ws.download(BUF_X)[0], y) < 1e-12      # V V^H y = y
    ws.close()


def test_full_size_direct_parity_headline_batch():
    """BASELINE headline at its full size -- 16 qubits, 40 blocks, 64 lanes with 64 different thetas,
    exactly the bench's unit of work -- compared lane by lane with the compiled CPU restatement."""
    from oracle import aqc_ref as cref
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    n, L, B = 16, 40, 64
    rng = np.random.default_rng(1640)
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, L))
    thetas = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(B)])
    y = orc.rand_state(n, rng)
    ws = Workspace(HipContext.of(make_circ(a)), batch=B)
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, 0)
    ws.gather_setup([0])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(a, thetas, y, 0, threads=8)
    assert maxdiff(hs[:, 0], hs_ref) < TOL
    assert maxdiff(grads, g_ref) < TOL
    ws.close()


@pytest.mark.parametrize("n,B", [(13, 150), (14, 70), (12, 300), (14, 150), (15, 80)])
def test_persistent_sweep_uneven_work(n, B):
    """The 2^12 sweep runs one persistent workgroup per CU over (tile, lane) items with the next tile prefetched into
    registers: item counts that are not a multiple of the grid (300 / 280 / 300 items on 256 CUs: some workgroups take two
    items, most one; 600 / 640 items: three contiguous items per workgroup, so the segments over which a workgroup
    accumulates R start and end in the middle of lanes of 4 / 8 tiles) and a per-lane theta, every lane against the
    compiled CPU restatement."""
    from oracle import aqc_ref as cref
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    rng = np.random.default_rng(n * 1000 + B)
    a = orc.Ansatz(n, "cz", orc.spin_blocks(n, 17))
    thetas = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(B)])
    y = orc.rand_state(n, rng)
    ws = Workspace(HipContext.of(make_circ(a)), batch=B)
    assert ws.plan_info(1)[1] == 12 and ws.kernel_family(1) == 3
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, 3)
    ws.gather_setup([3])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(a, thetas, y, 3, threads=8)
    assert maxdiff(hs[:, 0], hs_ref) < TOL
    assert maxdiff(grads, g_ref) < TOL
    ws.close()


@pytest.mark.parametrize("order2", [True, False])
def test_full_size_direct_parity_20_qubits(order2):
    """BASELINE configs[3] size (2^20 amplitudes, Trotter ansatz, Neel basis state): direct comparison."""
    from oracle import aqc_ref as cref
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    n = 20
    rng = np.random.default_rng(2020)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 2), second_order=order2)
    a = orc.as_ansatz(circ)
    thetas = orc.rand_thetas(circ.num_thetas, rng)[None, :]
    y = orc.rand_state(n, rng)
    neel = int("01" * (n // 2), 2)
    ws = Workspace(HipContext.of(circ), batch=1)
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, neel)
    ws.gather_setup([neel])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(a, thetas, y, neel, threads=1)
    assert maxdiff(hs[:, 0], hs_ref) < TOL
    assert maxdiff(grads, g_ref) < TOL
    ws.close()


def test_large_state_24_qubits_properties():
    """2^24 amplitudes (256 MiB per buffer, 8 stages per pass): beyond any size the CPU restatements are practical
    for in a test, so size-independent properties only -- norm preservation, V V^H = 1 on a sample of amplitudes,
    the objective's gradient against central differences of the same path, repeatability."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    n = 24
    rng = np.random.default_rng(24)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 30))
    ws = Workspace(HipContext.of(circ), batch=1)
    th = orc.rand_thetas(circ.num_thetas, rng)
    y = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    y /= np.linalg.norm(y)
    ws.upload(BUF_Y, y)
    ws.set_basis(BUF_X, 5)
    ws.gather_setup([5])
    hs, g = ws.eval(th[None, :], gather=True)
    hs2, g2 = ws.eval(th[None, :], gather=True)
    assert np.array_equal(g, g2) and np.array_equal(hs, hs2)            # bit-reproducible
    z = ws.download(BUF_Z, lane=0)
    assert abs(np.linalg.norm(z) - 1) < 1e-12
    sample = rng.integers(0, 1 << n, 64)
    ws.apply(False, BUF_Z, BUF_X)                                      # V V^H y
    assert maxdiff(ws.gather(BUF_X, sample)[0], y[sample]) < 1e-12
    ws.set_basis(BUF_X, 5)
    for t in (0, 3 * n + 17, circ.num_thetas - 1):                     # d/dtheta <V e5|y> = d/dtheta (V^H y)[5]
        f = []
        for sgn in (+1, -1):
            e = np.zeros_like(th); e[t] = sgn * 1e-5
            f.append(ws.eval((th + e)[None, :], gather=True, grad=False)[0][0, 0])
        assert abs((f[0] - f[1]) / 2e-5 - g[0, t]) < 1e-8
    ws.close()


@pytest.mark.parametrize("kind", ["cx", "cp", "trotter2"])
def test_gradient_vs_numeric_taylor_order(kind):
    """The reference's gradient harness (utils_dot_gradient_test.py:166-305): analytic gradient of the real
    objective against central differences with steps tau = 0.25 * 2^-k -- the relative errors fall, the last ones
    are <= 1e-5, and the Taylor residual |f(t + tau d) - f(t) - tau <g, d>| shrinks with order ~2."""
    from aqc_research_amd import core_operations as cop

    n = 5
    rng = np.random.default_rng(55)
    if kind == "trotter2":
        a = orc.Ansatz(n, "cx", orc.trotter_blocks(n, 1), True, True)
    else:
        a = orc.Ansatz(n, kind, np.stack([rng.permutation(n)[:2] for _ in range(8)], axis=1).astype(np.int64))
    circ = make_circ(a)
    th = orc.rand_thetas(a.num_thetas, rng)
    x, y = orc.rand_state(n, rng), orc.rand_state(n, rng)
    ws = np.zeros((4, 1 << n), complex)

    def fobj(t):   # 1 - |<V x|y>|^2
        return 1.0 - abs(np.vdot(cop.v_mul_vec(circ, t, x, np.zeros_like(x), ws), y)) ** 2

    vhy = cop.v_dagger_mul_vec(circ, th, y, np.zeros_like(y), ws)
    h = np.vdot(x, vhy)                                       # <V x|y> = <x|V^H y>
    grad = (-2 * np.conj(h) * cop.grad_of_dot_product(circ, th, x, vhy, ws)).real   # objective_lhs_sur_max.py:160-170 form
    d = rng.standard_normal(th.size)
    d /= np.linalg.norm(d)
    f0 = fobj(th)
    taus = [0.25 * 2.0 ** -k for k in range(4, 12)]
    rel, resid = [], []
    for tau in taus:
        fd = (fobj(th + tau * d) - fobj(th - tau * d)) / (2 * tau)
        rel.append(abs(fd - grad @ d) / max(abs(grad @ d), 1e-12))
        resid.append(abs(fobj(th + tau * d) - f0 - tau * (grad @ d)))
    assert max(rel[-4:]) <= 1e-5
    orders = [np.log2(resid[i] / resid[i + 1]) for i in range(3)]   # largest steps: far from round-off
    assert all(1.8 <= o <= 2.2 for o in orders), orders
