"""Round 5 (GPU): the sparse-lhs route of the sweep, the mirrored V^H with its checkpoint, V^H where the objective reads it.

Reference behaviour pinned here: grad_of_dot_product copies x into w (core_operations.py:892-935); the objectives sweep from
one-hot states or from a combination of two (objective_base.py:42-255, objective_lhs_sur_max.py:147-191).  The sparse route
must give what the dense route gives (same kernels on fewer tiles) and what the oracle gives (1e-10).
"""
import numpy as np
import pytest

from tests.helpers import TOL, maxdiff
from oracle import aqc_oracle as orc

pytestmark = pytest.mark.gpu


def _circ(n, ent="cx", depth=24, layout="spin"):
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure

    return ParametricCircuit(n, ent, create_ansatz_structure(n, layout, "full", depth))


def _trotter(n, layers):
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit

    return TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)


def _ws(circ, batch, monkeypatch, sparse=True, lazy=True, tile=0, min_items=1):
    from aqc_research_amd.engine import HipContext, Workspace

    monkeypatch.setenv("AQC_SPARSE_SWEEP", "1" if sparse else "0")
    monkeypatch.setenv("AQC_LAZY_Z", "1" if lazy else "0")
    monkeypatch.setenv("AQC_SPARSE_MIN_ITEMS", str(min_items))
    return Workspace(HipContext(circ), batch=batch, tile_bits_apply=tile, tile_bits_sweep=tile)


def _oracle_lane(circ, th, target, idx, coef):
    vh = orc.v_dagger_mul_vec(circ, th, target)
    x = np.zeros(1 << circ.num_qubits, complex)
    for i, c in zip(idx, coef):
        if i >= 0:
            x[i] = c
    return vh, orc.grad_of_dot_product(circ, th, x, vh)


@pytest.mark.parametrize("n,tile,ent", [(13, 12, "cx"), (14, 12, "cz"), (13, 10, "cp"), (12, 9, "cx"), (11, 8, "cz"), (14, 11, "cx")])
def test_sparse_route_equals_dense_route_and_oracle(n, tile, ent, monkeypatch):
    """set_basis with a different index per lane; separate calls (apply, gather, grad): sparse route vs AQC_SPARSE_SWEEP=0."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z

    rng = np.random.default_rng(100 + n + tile)
    circ = _circ(n, ent, depth=3 * n)
    B = 5
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    idx = rng.integers(0, 1 << n, size=B)
    idx[0] = 0
    res = {}
    for sparse in (True, False):
        ws = _ws(circ, B, monkeypatch, sparse=sparse, tile=tile)
        assert ws.plan_info(1)[0] >= 2, "the case must have more than one stage"
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, idx)
        ws.set_thetas(th)
        ws.apply(True, BUF_Y, BUF_Z)
        ws.grad(None, True)
        g = ws.get_grads()
        z = ws.download(BUF_Z)
        counts = ws.sparse_counts()
        assert (counts[0] == B) if sparse else (counts[0] == -1), counts
        ws.close()
        res[sparse] = (g, z)
    assert maxdiff(res[True][1], res[False][1]) == 0.0            # the same V^H
    assert maxdiff(res[True][0], res[False][0]) < 1e-13           # z of the later stages comes from the checkpoint: last bits differ
    for b in range(B):
        vh, g_ref = _oracle_lane(circ, th[b], tg[b], [int(idx[b])], [1.0])
        assert maxdiff(res[True][1][b], vh) < TOL
        assert maxdiff(res[True][0][b], g_ref) < TOL


def test_first_stage_alone_is_bit_identical_to_the_dense_route(monkeypatch):
    """Two stages, block_range confined to the first stage's blocks... every gradient entry that belongs to a group of the first
    sweep stage must be BIT-identical between the routes (same kernel, same operands, zero tiles contribute exact zeros)."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z

    rng = np.random.default_rng(5)
    n = 14
    circ = _circ(n, "cx", depth=30)
    B = 3
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    grads = {}
    first_ops = None
    for sparse in (True, False):
        ws = _ws(circ, B, monkeypatch, sparse=sparse, tile=12)
        nsub, nops, bits = ws.plan_stage(1, 0)
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, [0, 5, (1 << n) - 1])
        ws.set_thetas(th)
        ws.apply(True, BUF_Y, BUF_Z)
        ws.grad(None, True)
        grads[sparse] = ws.get_grads()
        # gate groups of the first stage (host-only planner query: the same plan)
        import ctypes
        from aqc_research_amd import _lib
        L = _lib.lib()
        ns, nb, no = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        ops = (ctypes.c_int * 4096)()
        L.aqc_plan_query(ws.ctx.handle, 1, 1, 12, -1, 0, ctypes.byref(ns), None, ctypes.byref(nb), ops, ctypes.byref(no))
        first_ops = sorted(ops[: no.value])
        ws.close()
    theta_of = []
    for gi in first_ops:   # thetas of the groups: front groups 3 each, blocks 4 each (cx)
        theta_of += list(range(3 * gi, 3 * gi + 3)) if gi < n else list(range(3 * n + 4 * (gi - n), 3 * n + 4 * (gi - n) + 4))
    assert len(theta_of) > 20
    assert np.array_equal(grads[True][:, theta_of], grads[False][:, theta_of])


def test_combo_support_moves_between_evaluations(monkeypatch):
    """set_combo with two basis states per lane in different / equal tiles, changed from call to call: tiles of W that the new
    list drops are cleared, lanes with one tile contribute one partial."""
    from aqc_research_amd.engine import BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(11)
    n = 14
    circ = _circ(n, "cx", depth=28)
    B = 4
    ws = _ws(circ, B, monkeypatch, tile=12)
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    ws.upload(BUF_Y, tg)
    hi = 1 << 12
    plans = [
        [(0, hi), (3, -1), (hi + 7, 2 * hi + 1), (9, 10)],          # two tiles / one term / two tiles / same tile
        [(0, 2 * hi), (3, 3 * hi), (hi + 7, -1), (3 * hi, 3 * hi + 1)],
        [(5, -1), (6, -1), (7, -1), (8, -1)],
        [(hi, 3 * hi), (2 * hi, 0), (1, hi + 1), (3 * hi + 5, 2)],
    ]
    for step, plan in enumerate(plans):
        th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
        idx = np.array(plan, dtype=np.int64)
        coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
        ws.set_thetas(th)
        ws.apply(True, BUF_Y, BUF_Z)
        ws.set_combo(BUF_X2, idx, coef)
        ws.grad_from(BUF_X2)
        g = ws.get_grads()
        for b in range(B):
            _, g_ref = _oracle_lane(circ, th[b], tg[b], idx[b], coef[b])
            assert maxdiff(g[b], g_ref) < TOL, (step, b)
        want = sum(1 if (j < 0 or (i >> 12) == (j >> 12)) else 2 for i, j in plan)
        assert ws.sparse_counts()[0] == want
    ws.close()


@pytest.mark.parametrize("n,tile,layers", [(13, 9, 2), (14, 10, 3)])
def test_three_or_more_stages_second_scratch_pair(n, tile, layers, monkeypatch):
    """Plans of >= 3 stages: the later stages work on their own scratch pair, W stays clean, two evaluations in a row agree with
    the oracle (2nd-order Trotter ansatz: thetas that collect two slots)."""
    from aqc_research_amd.engine import BUF_X, BUF_Y

    rng = np.random.default_rng(21 + n)
    circ = _trotter(n, layers)
    B = 3
    ws = _ws(circ, B, monkeypatch, tile=tile)
    assert ws.plan_info(1)[0] >= 3
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    ws.upload(BUF_Y, tg)
    neel = int("01" * (n // 2) + ("0" if n % 2 else ""), 2) & ((1 << n) - 1)
    ws.set_basis(BUF_X, neel)
    ws.gather_setup([neel] + [neel ^ (1 << q) for q in range(n)])
    for _ in range(2):
        th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
        ws.set_thetas(th)
        ws.objective_launch(BUF_X)
        hs = ws.gather_fetch()
        g = ws.get_grads()
        for b in range(B):
            vh, g_ref = _oracle_lane(circ, th[b], tg[b], [neel], [1.0])
            assert abs(hs[b, 0] - vh[neel]) < TOL and abs(hs[b, n] - vh[neel ^ (1 << (n - 1))]) < TOL
            assert maxdiff(g[b], g_ref) < TOL
    ws.close()


def test_objective_launch_leaves_partial_z_that_readers_complete(monkeypatch):
    """V^H where the evaluation reads it: amplitudes and gradients as usual; download / gather / vdot / a second sweep from another
    lhs state complete Z first; once the thetas have changed a reader fails loudly, a whole-buffer writer takes Z over."""
    from aqc_research_amd.engine import BUF_X, BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(31)
    n = 14
    circ = _circ(n, "cx", depth=26)
    B = 4
    monkeypatch.setenv("AQC_PROJECTED_VDAG", "0")   # (V^H by its stages: by projection it is test_objective_by_projection_equals_the_stages_of_vdag)
    ws = _ws(circ, B, monkeypatch, tile=12)
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    ws.upload(BUF_Y, tg)
    ws.set_basis(BUF_X, 0)
    flips = [0] + [1 << q for q in range(n)]
    ws.gather_setup(flips)
    ws.set_thetas(th)
    ws.objective_launch(BUF_X)
    hs = ws.gather_fetch()
    g = ws.get_grads()
    sw, _, vd = ws.sparse_counts()
    assert sw == B and vd == B * 3            # tile 0 and the tiles of the flips on qubits 12, 13
    vhs = []
    for b in range(B):
        vh, g_ref = _oracle_lane(circ, th[b], tg[b], [0], [1.0])
        vhs.append(vh)
        assert maxdiff(hs[b], vh[flips]) < TOL and maxdiff(g[b], g_ref) < TOL
    # an index outside the computed tiles: the reader completes Z
    far = (3 << 12) + 77
    got = ws.gather(BUF_Z, [far])
    assert max(abs(got[b, 0] - vhs[b][far]) for b in range(B)) < TOL
    ws.objective_launch(BUF_X)
    assert maxdiff(ws.download(BUF_Z), np.stack(vhs)) < TOL
    ws.objective_launch(BUF_X)
    ws.set_basis(BUF_X2, far)                 # a sweep from a state the partial Z was not computed for
    ws.grad_from(BUF_X2)
    g2 = ws.get_grads()
    for b in range(B):
        _, g_ref = _oracle_lane(circ, th[b], tg[b], [far], [1.0])
        assert maxdiff(g2[b], g_ref) < TOL
    # thetas changed while Z is partial: a reader refuses, aqc_ws_apply makes Z whole again
    ws.objective_launch(BUF_X)
    ws.set_thetas(th[::-1].copy())
    with pytest.raises(RuntimeError, match="BUF_Z holds"):
        ws.download(BUF_Z)
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)
    for b in range(B):
        assert maxdiff(z[b], orc.v_dagger_mul_vec(circ, th[B - 1 - b], tg[b])) < TOL
    ws.close()


def test_one_call_evaluations_on_the_sparse_route(monkeypatch):
    """aqc_ws_eval (captured graph, replayed) and aqc_ws_surrogate_eval with the sparse route forced on a small problem: values
    equal the dense-route workspace's; the leading flip state moves between tiles from call to call."""
    from aqc_research_amd.engine import BUF_X, BUF_Y

    rng = np.random.default_rng(41)
    n = 13
    circ = _circ(n, "cx", depth=20)
    B = 6
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    flips = [0] + [1 << q for q in range(n)]
    ths = [np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)]) for _ in range(4)]
    out = {}
    for sparse in (True, False):
        ws = _ws(circ, B, monkeypatch, sparse=sparse, tile=10)
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, 0)
        ws.gather_setup(flips)
        rec = []
        for th in ths:
            hs, g = ws.eval(th, vdag=True, gather=True, grad=True)
            rec.append((hs.copy(), g.copy()))
        w = np.ones(B)
        m = np.zeros(B, dtype=np.int64)
        for th in ths:
            f, fid, hs, g = ws.surrogate_eval(th, w, m, update_state=True)
            rec.append((f.copy(), g.copy(), m.copy()))
        out[sparse] = rec
        ws.close()
    for a, b in zip(out[True], out[False]):
        for u, v in zip(a, b):
            assert maxdiff(u, v) < 1e-12
    assert len({tuple(r[2]) for r in out[True][4:]}) > 1 or max(out[True][-1][2]) > 0   # some lane is led by a flip state
    for k, th in enumerate(ths):
        for b in range(B):
            vh, g_ref = _oracle_lane(circ, th[b], tg[b], [0], [1.0])
            assert maxdiff(out[True][k][0][b], vh[flips]) < TOL and maxdiff(out[True][k][1][b], g_ref) < TOL


def test_device_lbfgs_on_the_sparse_route(monkeypatch):
    """aqc_ws_lbfgs with the sparse route forced: every lane reaches what the dense-route run reaches."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective

    rng = np.random.default_rng(51)
    n = 13
    circ = _trotter(n, 1)
    B = 4
    targets = np.stack([orc.v_mul_vec(circ, 0.3 * orc.rand_thetas(circ.num_thetas, rng), np.eye(1 << n)[0].astype(complex)) for _ in range(B)])
    x0 = 0.05 * np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    fids = {}
    for sparse in (True, False):
        monkeypatch.setenv("AQC_SPARSE_SWEEP", "1" if sparse else "0")
        monkeypatch.setenv("AQC_SPARSE_MIN_ITEMS", "1")
        monkeypatch.setenv("AQC_TILE_BITS_APPLY", "10")
        monkeypatch.setenv("AQC_TILE_BITS_SWEEP", "10")
        obj = BatchedSurrogateObjective(circ, targets)
        res = obj.minimize_on_device(x0.copy(), maxiter=25)
        fids[sparse] = (np.asarray(res["fun"]), np.asarray(res["fidelity"]), np.asarray(res["x"]))
        obj.close() if hasattr(obj, "close") else None
    # the same optimisation: the routes differ in the last bits of z, the trajectories stay together over 25 iterations
    assert np.allclose(fids[True][0], fids[False][0], rtol=1e-6, atol=1e-9)
    assert np.allclose(fids[True][1], fids[False][1], rtol=1e-6, atol=1e-9)
    assert maxdiff(fids[True][2], fids[False][2]) < 1e-6
    assert np.all(fids[True][0] < 1.0)


@pytest.mark.parametrize("n,tile,ent,trot", [(12, 12, "cx", 2), (10, 10, "cz", 0), (9, 9, "cp", 0), (8, 8, "cx", 0), (14, 12, "cx", 0), (13, 11, "cx", 1),
                                             (13, 9, "cz", 0)])
def test_zero_groups_of_w_are_skipped_without_changing_a_bit(n, tile, ent, trot, monkeypatch):
    """(Opt-in variant, AQC_SKIP_ZERO_W=1.)  Inside a stage the kernel leaves out the W / R products of 16-chunk groups (and K-steps of the W product) where w is zero
    because bits no gate has mixed yet differ from the basis index (AQC_SKIP_ZERO_W): the products would multiply exact zeros, so
    every gradient entry keeps its value bit for bit -- single-stage plans, several stages, both routes, one and two basis states
    per lane, every tile size."""
    from aqc_research_amd.engine import BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(300 + n + tile)
    circ = _trotter(n, trot) if trot else _circ(n, ent, depth=3 * n)
    B = 4
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    hi = 1 << (n - 1)
    idx = np.array([[0, -1], [5, hi + 5], [hi | 3, 3 ^ (1 << (n - 2))], [(1 << n) - 1, 1]], dtype=np.int64)
    coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
    res = {}
    monkeypatch.setenv("AQC_R_ONLY_LAST", "0")   # (the R-only last sub-stage rounds R differently; it has its own test below)
    for skip in ("1", "0"):
        for sparse in (True, False):
            monkeypatch.setenv("AQC_SKIP_ZERO_W", skip)
            ws = _ws(circ, B, monkeypatch, sparse=sparse, tile=tile)
            ws.upload(BUF_Y, tg)
            ws.set_thetas(th)
            ws.apply(True, BUF_Y, BUF_Z)
            ws.set_combo(BUF_X2, idx, coef)
            ws.grad_from(BUF_X2)
            res[(skip, sparse)] = ws.get_grads()
            ws.close()
    assert np.array_equal(res[("1", True)], res[("0", True)])
    assert np.array_equal(res[("1", False)], res[("0", False)])
    for b in range(B):
        _, g_ref = _oracle_lane(circ, th[b], tg[b], idx[b], coef[b])
        assert maxdiff(res[("1", True)][b], g_ref) < TOL


@pytest.mark.parametrize("n,tile,ent,trot", [(12, 12, "cx", 2), (12, 12, "cx", 0), (10, 10, "cz", 0), (9, 9, "cp", 0), (8, 8, "cx", 0), (14, 12, "cx", 0),
                                             (13, 11, "cx", 1), (13, 9, "cz", 0), (16, 12, "cx", 0)])
def test_r_only_last_substage_equals_full_sweep_and_oracle(n, tile, ent, trot, monkeypatch):
    """The sweep's last sub-stage forms R = U (Z W^dagger) U^dagger from its inputs instead of updating w and z (nobody reads them
    afterwards) and the gradient kernel conjugates by U: same gradient as the full form to rounding (AQC_R_ONLY_LAST=0) and as the
    oracle, on both routes, one and two basis states per lane, one and several stages; the plan says which sub-stage it is."""
    from aqc_research_amd.engine import BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(500 + n + tile)
    circ = _trotter(n, trot) if trot else _circ(n, ent, depth=3 * n)
    B = 4
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    hi = 1 << (n - 1)
    idx = np.array([[0, -1], [5, hi + 5], [hi | 3, 3 ^ (1 << (n - 2))], [(1 << n) - 1, 1]], dtype=np.int64)
    coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
    res, used = {}, {}
    for r_only in ("1", "0"):
        for sparse in (True, False):
            monkeypatch.setenv("AQC_R_ONLY_LAST", r_only)
            monkeypatch.setenv("AQC_R_ONLY_MAX_SUBS", "64")
            ws = _ws(circ, B, monkeypatch, sparse=sparse, tile=tile)
            used[(r_only, sparse)] = ws.sweep_r_only_sub()
            ws.upload(BUF_Y, tg)
            ws.set_thetas(th)
            ws.apply(True, BUF_Y, BUF_Z)
            ws.set_combo(BUF_X2, idx, coef)
            ws.grad_from(BUF_X2)
            res[(r_only, sparse)] = ws.get_grads()
            # a second evaluation with other angles through the same workspace (u planes rebuilt, R-only again)
            ws.set_thetas(th[::-1].copy())
            ws.apply(True, BUF_Y, BUF_Z)
            ws.grad_from(BUF_X2)
            res[(r_only, sparse, 2)] = ws.get_grads()
            ws.close()
    assert used[("0", True)] == -1 and used[("0", False)] == -1
    for sparse in (True, False):
        assert maxdiff(res[("1", sparse)], res[("0", sparse)]) < 1e-13
        assert maxdiff(res[("1", sparse, 2)], res[("0", sparse, 2)]) < 1e-13
    for b in range(B):
        _, g_ref = _oracle_lane(circ, th[b], tg[b], idx[b], coef[b])
        assert maxdiff(res[("1", True)][b], g_ref) < TOL
        assert maxdiff(res[("1", False)][b], g_ref) < TOL
        _, g_ref2 = _oracle_lane(circ, th[B - 1 - b], tg[b], idx[b], coef[b])
        assert maxdiff(res[("1", True, 2)][b], g_ref2) < TOL


@pytest.mark.parametrize("n,tile,kind,arg", [(16, 12, "spin", 40), (16, 12, "spin", 20), (14, 12, "spin", 40), (14, 12, "trotter", 2), (16, 12, "trotter", 2),
                                             (18, 12, "spin", 40), (15, 10, "spin", 30), (14, 9, "spin", 24), (13, 8, "spin", 18), (16, 11, "spin", 36),
                                             (15, 8, "spin", 30), (16, 8, "spin", 24)])   # (the last two: two virtual stages)
def test_projected_route_equals_full_size_sweep_and_oracle(n, tile, kind, arg, monkeypatch):
    """The sweep's stages after the first on the virtual register (AQC_PROJECTED, csrc/aqc_ws_project.cpp): the same gradient as the
    full-size stages (AQC_PROJECTED=0) to rounding and as the oracle -- one and two basis states per lane (same tile, different tiles),
    one and several virtual stages, registers padded to 8 virtual qubits, two evaluations per workspace."""
    from aqc_research_amd.engine import BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(700 + n + tile)
    circ = _trotter(n, arg) if kind == "trotter" else _circ(n, "cx", depth=arg)
    B = 5
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    hi = 1 << (n - 1)
    idx = np.array([[0, -1], [5, hi + 5], [hi | 3, 3 ^ (1 << (n - 2))], [(1 << n) - 1, 1], [(1 << (n - 2)) | 7, -1]], dtype=np.int64)
    coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
    res, info = {}, {}
    for proj in ("1", "0"):
        monkeypatch.setenv("AQC_PROJECTED", proj)
        ws = _ws(circ, B, monkeypatch, sparse=True, tile=tile)
        info[proj] = ws.projected_info()
        ws.upload(BUF_Y, tg)
        for rep in range(2):
            ws.set_thetas(th if rep == 0 else th[::-1].copy())
            ws.apply(True, BUF_Y, BUF_Z)
            ws.set_combo(BUF_X2, idx, coef)
            ws.grad_from(BUF_X2)
            res[(proj, rep)] = ws.get_grads()
        ws.close()
    assert info["0"] == {}
    if not info["1"]:   # (a shape whose later stages touch too much of the register: both runs took the full-size stages)
        assert all(np.array_equal(res[("1", rep)], res[("0", rep)]) for rep in range(2))
        return
    assert info["1"]["virtual_qubits"] + 2 <= n
    for rep in range(2):
        assert maxdiff(res[("1", rep)], res[("0", rep)]) < 1e-13
    for b in range(B):
        _, g_ref = _oracle_lane(circ, th[b], tg[b], idx[b], coef[b])
        assert maxdiff(res[("1", 0)][b], g_ref) < TOL
        _, g_ref2 = _oracle_lane(circ, th[B - 1 - b], tg[b], idx[b], coef[b])
        assert maxdiff(res[("1", 1)][b], g_ref2) < TOL


def test_projected_route_through_the_one_call_evaluations(monkeypatch):
    """aqc_ws_eval (captured graph, replayed) and the surrogate objective's one-call evaluation take the projected route: same
    amplitudes and gradients as with AQC_PROJECTED=0, three calls each with new thetas."""
    from aqc_research_amd.engine import BUF_X, BUF_Y

    n = 16
    rng = np.random.default_rng(811)
    circ = _circ(n, "cx", depth=40)
    B = 8
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    ths = [np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)]) for _ in range(3)]
    flips = np.array([0] + [1 << q for q in range(n)], dtype=np.int64)
    out = {}
    for proj in ("1", "0"):
        monkeypatch.setenv("AQC_PROJECTED", proj)
        ws = _ws(circ, B, monkeypatch, sparse=True, tile=12)
        assert bool(ws.projected_info()) == (proj == "1")
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, 0)
        ws.gather_setup(flips)
        got = []
        for th in ths:
            hs, g = ws.eval(th, vdag=True, gather=True, grad=True, x_buf=BUF_X, block_range=(0, circ.num_blocks), front_layer=True)
            got.append((hs.copy(), g.copy()))
        w = np.full(B, 0.3)
        mx = np.array([0, 1, 13, 16, 0, 5, 14, 2], dtype=np.int64)
        for th in ths:
            f, fid, hs, gc = ws.surrogate_eval(th, w.copy(), mx.copy(), 2, (0, circ.num_blocks), True)
            got.append((np.asarray(hs).copy(), np.asarray(gc).copy(), np.asarray(f).copy()))
        out[proj] = got
        ws.close()
    for a, b in zip(out["1"], out["0"]):
        for x, y in zip(a, b):
            assert maxdiff(x, y) < 1e-13
    for i, th in enumerate(ths):
        for b in range(0, B, 3):
            vh = orc.v_dagger_mul_vec(circ, th[b], tg[b])
            x = np.zeros(1 << n, complex); x[0] = 1.0
            assert maxdiff(out["1"][i][0][b], vh[flips]) < TOL
            assert maxdiff(out["1"][i][1][b], orc.grad_of_dot_product(circ, th[b], x, vh)) < TOL


@pytest.mark.parametrize("n,depth,case,tile", [(16, 40, "zero", 12), (16, 40, "shifted", 12), (14, 40, "zero", 12), (18, 40, "shifted", 12),
                                               (14, 24, "zero", 8), (15, 30, "shifted", 8)])   # (the last two: two virtual stages)
def test_objective_by_projection_equals_the_stages_of_vdag(n, depth, case, tile, monkeypatch):
    """One-call evaluations from ONE basis state per lane (set_basis) whose gather set stays in the lane's first-stage tile or flips
    bits outside it: V^H's stages are replaced by two passes over the target (AQC_PROJECTED_VDAG, csrc/aqc_ws_project.cpp).  Same
    amplitudes and gradient as with the stages (AQC_PROJECTED_VDAG=0) and as the oracle, through aqc_ws_objective_launch and
    aqc_ws_eval (graph replays); Z, partial afterwards, is completed from Y when somebody reads it."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z

    rng = np.random.default_rng(900 + n)
    circ = _circ(n, "cx", depth=depth)
    B = 40
    T = circ.num_thetas
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    ths = [np.stack([orc.rand_thetas(T, rng) for _ in range(B)]) for _ in range(2)]
    hi = n - tile   # (qubits above the first stage's tile when it takes the low ones)
    if case == "zero":
        basis = np.zeros(B, dtype=np.int64)
        gather = np.array([0] + [1 << q for q in range(n)], dtype=np.int64)
    else:   # the same index on the first stage's bits, different ones above; the gather set moves above only
        basis = np.array([5 | ((b * 7) % (1 << hi)) << tile for b in range(B)], dtype=np.int64)
        gather = np.array([5 | (f << tile) for f in sorted({0, 1, 2, (1 << hi) - 1, 1 << (hi - 1)})], dtype=np.int64)
    from aqc_research_amd._lib import K_APPLY_VIRTUAL

    out, passes = {}, {}
    monkeypatch.setenv("AQC_PROJECTED_VDAG_MIN_ELEMS", "1")
    for mode in ("1", "0"):
        monkeypatch.setenv("AQC_PROJECTED_VDAG", mode)
        ws = _ws(circ, B, monkeypatch, sparse=True, tile=tile)
        if not ws.projected_info():
            pytest.skip("the plan of this shape has no projected route")
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, basis)
        ws.gather_setup(gather)
        got = []
        for th in ths:   # aqc_ws_eval: thetas in, amplitudes and gradient out
            hs, g = ws.eval(th, vdag=True, gather=True, grad=True, x_buf=BUF_X, block_range=(0, circ.num_blocks), front_layer=True)
            got.append((hs.copy(), g.copy()))
        for th in ths:   # the driver's form: thetas resident, launches only
            ws.set_thetas(th)
            ws.objective_launch(BUF_X)
            got.append((ws.gather_fetch().copy(), ws.get_grads().copy()))
        z_after = ws.download(BUF_Z)   # (a reader of Z: completed first)
        ws.profile(True)               # which route ran: by projection the virtual plan is also applied forwards and backwards
        ws.set_thetas(ths[0])
        ws.objective_launch(BUF_X)
        ws.sync()
        passes[mode] = ws.profile_get(K_APPLY_VIRTUAL)[0]
        ws.profile(False)
        out[mode] = (got, z_after)
        ws.close()
    assert passes["1"] >= 2 and passes["0"] == 0
    for a, b in zip(out["1"][0], out["0"][0]):
        assert maxdiff(a[0], b[0]) < 1e-13 and maxdiff(a[1], b[1]) < 1e-13
    for i, th in enumerate(ths + ths):
        for b in range(0, B, 9):
            vh = orc.v_dagger_mul_vec(circ, th[b], tg[b])
            x = np.zeros(1 << n, complex); x[basis[b]] = 1.0
            assert maxdiff(out["1"][0][i][0][b], vh[gather]) < TOL
            assert maxdiff(out["1"][0][i][1][b], orc.grad_of_dot_product(circ, th[b], x, vh)) < TOL
    for b in (0, B - 1):
        assert maxdiff(out["1"][1][b], orc.v_dagger_mul_vec(circ, ths[-1][b], tg[b])) < TOL


@pytest.mark.parametrize("n,blocks", [(16, 40), (18, 30), (20, 40)])
def test_fused_pass_equals_the_two_launches(n, blocks, monkeypatch):
    """Objective by projection: both products from one fetch of the target (project_fused_kernel; at 20 qubits the summed index has 512
    values: two workgroups per item and a fixed-order sum of their partial projections) against the two launches
    (AQC_PROJECTED_FUSED=0) and against V^H by its stages (AQC_PROJECTED_VDAG=0): amplitudes and gradients."""
    from aqc_research_amd.engine import BUF_X, BUF_Y

    rng = np.random.default_rng(77 + n)
    circ = _circ(n, "cx", depth=blocks)
    B = 3
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    tg = rng.standard_normal((B, 1 << n)) + 1j * rng.standard_normal((B, 1 << n))
    tg /= np.linalg.norm(tg, axis=1, keepdims=True)
    flips = np.array([0] + [1 << q for q in range(n)], dtype=np.int64)
    monkeypatch.setenv("AQC_PROJECTED_VDAG_MIN_ELEMS", "1")
    res = {}
    for name, env in (("fused", {}), ("whole", {"AQC_PROJECTED_FUSED_MAX_SHARES": "1"}), ("two", {"AQC_PROJECTED_FUSED": "0"}),
                      ("stages", {"AQC_PROJECTED_VDAG": "0"})):   # fused: the walk split over the touched bits (few lanes); whole: one workgroup per item
        for k in ("AQC_PROJECTED_FUSED", "AQC_PROJECTED_VDAG", "AQC_PROJECTED_FUSED_MAX_SHARES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ws = _ws(circ, B, monkeypatch, sparse=True, tile=12)
        info = ws.projected_info()
        assert info and info["shared_with_first_stage"] <= 4 and info["summed_bits"] <= 10
        ws.upload(BUF_Y, tg)
        ws.set_basis(BUF_X, 0)
        ws.gather_setup(flips)
        ws.set_thetas(th)
        ws.objective_launch(BUF_X)
        res[name] = (ws.gather_fetch().copy(), ws.get_grads().copy(), info["summed_bits"])
        ws.close()
    for name in ("whole", "two", "stages"):
        assert maxdiff(res["fused"][0], res[name][0]) < 1e-13 and maxdiff(res["fused"][1], res[name][1]) < 1e-13
    if n == 20:
        assert res["fused"][2] == 9   # (the case the test is for: 512 summed values)
    if n == 16:
        vh = orc.v_dagger_mul_vec(circ, th[1], tg[1])
        x = np.zeros(1 << n, complex); x[0] = 1.0
        assert maxdiff(res["fused"][0][1], vh[flips]) < TOL and maxdiff(res["fused"][1][1], orc.grad_of_dot_product(circ, th[1], x, vh)) < TOL


# one shape per kind of virtual register the planner produces on random spin ansaetze: (qubits shared with the first stage, virtual
# stages, register padded to 8 qubits) -- found with HipContext.plan_projected on the host
_ROUTE_SHAPES = [(16, 7, 8), (14, 7, 10), (17, 5, 8), (16, 10, 9), (14, 6, 11), (17, 15, 8), (16, 22, 10), (12, 19, 10), (15, 21, 8),
                 (12, 25, 10), (13, 23, 12), (15, 29, 8), (15, 36, 10), (13, 37, 12), (16, 48, 12)]


@pytest.mark.parametrize("n,blocks,tile", _ROUTE_SHAPES)
def test_projected_routes_on_every_kind_of_virtual_register(n, blocks, tile, monkeypatch):
    """Both projected routes against the full-size stages (AQC_PROJECTED=0) on shapes that cover what proj_plan can produce: 0 to 5
    qubits shared with the first stage (5: two column blocks, the objective's two products as two launches), one and two virtual
    stages, registers padded to 8 qubits.  Sweep from two basis states per lane (projection of the checkpoint), then a one-call
    evaluation from one basis state with a flip-state gather set (objective by projection)."""
    from aqc_research_amd.engine import BUF_X, BUF_X2, BUF_Y, BUF_Z

    rng = np.random.default_rng(1000 * n + 10 * blocks + tile)
    circ = _circ(n, "cx", depth=blocks)
    B = 3
    T = circ.num_thetas
    th = np.stack([orc.rand_thetas(T, rng) for _ in range(B)])
    tg = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    top = 1 << (n - 1)
    idx = np.array([[0, -1], [3, top | 3], [top | 5, (1 << (n - 2)) | 6]], dtype=np.int64)
    coef = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
    flips = np.array([0] + [1 << q for q in range(n)], dtype=np.int64)
    monkeypatch.setenv("AQC_PROJECTED_VDAG_MIN_ELEMS", "1")
    res = {}
    for proj in ("1", "0"):
        monkeypatch.setenv("AQC_PROJECTED", proj)
        ws = _ws(circ, B, monkeypatch, sparse=True, tile=tile)
        info = ws.projected_info()
        assert bool(info) == (proj == "1"), "the shape was chosen for its route"
        ws.upload(BUF_Y, tg)
        ws.set_thetas(th)
        ws.apply(True, BUF_Y, BUF_Z)
        ws.set_combo(BUF_X2, idx, coef)
        ws.grad_from(BUF_X2)
        g_combo = ws.get_grads()
        ws.set_basis(BUF_X, 0)
        ws.gather_setup(flips)
        ws.set_thetas(th[::-1].copy())
        ws.objective_launch(BUF_X)
        res[proj] = (g_combo, ws.gather_fetch().copy(), ws.get_grads().copy())
        ws.close()
    for a, b in zip(res["1"], res["0"]):
        assert maxdiff(a, b) < 1e-13
    if n <= 14:   # (the full-size route is checked against the oracle everywhere else; here once more on the small shapes)
        _, g_ref = _oracle_lane(circ, th[1], tg[1], idx[1], coef[1])
        assert maxdiff(res["1"][0][1], g_ref) < TOL
        vh = orc.v_dagger_mul_vec(circ, th[B - 1], tg[0])
        assert maxdiff(res["1"][1][0], vh[flips]) < TOL


class _Op:
    def __init__(self, name, params=()):
        self.name, self.params = name, list(params)


class _Ins:
    def __init__(self, name, qubits, params=()):
        self.operation, self.qubits = _Op(name, params), list(qubits)


class _DuckCircuit:
    """What Qiskit's QuantumCircuit exposes to a gate walk: num_qubits, data[i].operation.name / .params, data[i].qubits."""

    def __init__(self, n):
        self.num_qubits, self.data, self.global_phase = n, [], 0.3

    def add(self, name, qubits, params=()):
        self.data.append(_Ins(name, qubits, params))
        return self


def _dense_gate(n, g, qubits):
    """Reference simulation of one gate on the full 2^n space (bit q of the index = qubit q)."""
    dim = 1 << n
    out = np.zeros((dim, dim), complex)
    for col in range(dim):
        if len(qubits) == 1:
            q = qubits[0]
            b = (col >> q) & 1
            for nb in range(2):
                out[col ^ ((b ^ nb) << q), col] += g[nb, b]
        else:
            q0, q1 = qubits
            b = 2 * ((col >> q0) & 1) + ((col >> q1) & 1)
            for nb in range(4):
                row = (col & ~((1 << q0) | (1 << q1))) | ((nb >> 1) << q0) | ((nb & 1) << q1)
                out[row, col] += g[nb, b]
    return out


def test_generic_state_handler_general_preparation_circuits():
    """state_prep_func returning a general circuit (objective_base.py:258-342, GenericStateHandler): the states S|0>, S X_i|0> are
    built gate by gate on the device; SpSurrogateObjectiveMax on them = the surrogate formula on the oracle's V^H and sweeps."""
    from aqc_research_amd.model_sp_lhs.objective_base import GenericStateHandler, _circuit_gate_matrix
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n = 6
    rng = np.random.default_rng(77)
    qc = _DuckCircuit(n)
    qc.add("h", [0]).add("ry", [1], [0.7]).add("cx", [0, 2]).add("rz", [2], [-1.1]).add("barrier", [0, 1]).add("cp", [3, 1], [0.4])
    qc.add("sx", [4]).add("swap", [4, 5]).add("u", [3], [0.3, 1.2, -0.5]).add("cz", [5, 0]).add("t", [2]).add("cy", [1, 4]).add("x", [5])
    handler = GenericStateHandler(n, 1, lambda _n: qc)
    full = np.eye(1 << n, dtype=complex)
    for ins in qc.data:
        if ins.operation.name == "barrier":
            continue
        full = _dense_gate(n, _circuit_gate_matrix(ins.operation.name, ins.operation.params), ins.qubits) @ full
    full *= np.exp(1j * qc.global_phase)
    want = np.stack([full[:, 0]] + [full[:, 1 << q] for q in range(n)])
    assert handler.num_states == n + 1 and maxdiff(np.stack([handler.init_state(i) for i in range(n + 1)]), want) < 1e-13
    # the objective on those states
    circ = _circ(n, "cx", depth=12)
    target = orc.rand_state(n, rng)
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: qc, enable_optim_stats=False, verbose=0)
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    weight, max_no = 1.0, 0
    for _ in range(3):
        th = orc.rand_thetas(circ.num_thetas, rng)
        f = objv.objective(th)
        g = objv.gradient(th)
        vh = orc.v_dagger_mul_vec(circ, th, target)
        hs = want.conj() @ vh
        hs2 = np.abs(hs) ** 2
        best = hs2[max_no]
        for i in range(n + 1):
            if 1.1 * best < hs2[i]:
                best, max_no = hs2[i], i
        f_ref = 1.0 - (1.0 - weight) * hs2[0] - weight * hs2[max_no]
        g0 = orc.grad_of_dot_product(circ, th, want[0], vh)
        if max_no == 0:
            g_ref = (g0 * (-2 * np.conj(hs[0]))).real
        else:
            gm = orc.grad_of_dot_product(circ, th, want[max_no], vh)
            g_ref = (g0 * (-2 * (1 - weight) * np.conj(hs[0]))).real + (gm * (-2 * weight * np.conj(hs[max_no]))).real
        weight += 0.1 * (np.sqrt(abs(f_ref)) - weight)
        assert abs(f - f_ref) < TOL and maxdiff(g, g_ref) < TOL
    with pytest.raises(NotImplementedError):
        GenericStateHandler(n, 1, lambda _n: _DuckCircuit(n).add("ccx", [0, 1, 2]))
