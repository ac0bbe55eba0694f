"""Pins the CPU oracle to golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import TOL, ansatz_from, load, maxdiff, mps_from

SV = load("state_vector.npz")
MAT = load("matrix.npz")
OBJ = load("objectives.npz")
MPS = load("mps.npz")
G22 = load("gate2x2.npz")


@pytest.mark.parametrize("key", [str(k) for k in SV["names"]])
def test_state_vector(key):
    a = ansatz_from(SV, key)
    th, x, y = SV[f"{key}/thetas"], SV[f"{key}/x"], SV[f"{key}/y"]
    assert maxdiff(orc.v_mul_vec(a, th, x), SV[f"{key}/v_x"]) < TOL
    vhy = orc.v_dagger_mul_vec(a, th, y)
    assert maxdiff(vhy, SV[f"{key}/vh_y"]) < TOL
    assert maxdiff(orc.grad_of_dot_product(a, th, x, vhy), SV[f"{key}/grad_full"]) < TOL
    br = tuple(int(v) for v in SV[f"{key}/block_range"])
    g = orc.grad_of_dot_product(a, th, x, vhy, block_range=br, front_layer=False)
    assert maxdiff(g, SV[f"{key}/grad_part"]) < TOL
    # V V^H = identity (reference test_core_operations.py:252-281)
    assert maxdiff(orc.v_mul_vec(a, th, vhy), y) < TOL


def test_gate2x2_degenerate():
    n, vec = int(G22["n"]), G22["vec"]
    for i, g in enumerate(G22["gates"]):
        for q in range(n):
            out = vec.copy()
            orc.gate2x2(out, 1 << q, g)
            assert maxdiff(out, G22["outs"][i, q]) < TOL


@pytest.mark.parametrize("key", [str(k) for k in MAT["names"]])
def test_matrix(key):
    a = ansatz_from(MAT, key)
    th = MAT[f"{key}/thetas"]
    if key.endswith("_cd"):
        t1, f1 = orc.coord_descent_single_sweep(a, th, MAT[f"{key}/target"])
        assert maxdiff(t1, MAT[f"{key}/thetas_1"]) < 1e-9 and abs(f1 - float(MAT[f"{key}/fobj_1"])) < 1e-9
        t2, f2 = orc.coord_descent_single_sweep(a, t1, MAT[f"{key}/target"])
        assert maxdiff(t2, MAT[f"{key}/thetas_2"]) < 1e-8 and abs(f2 - float(MAT[f"{key}/fobj_2"])) < 1e-8
        return
    x, y = MAT[f"{key}/x"], MAT[f"{key}/y"]
    assert maxdiff(orc.v_mul_mat(a, th, x), MAT[f"{key}/v_x"]) < TOL
    vhy = orc.v_dagger_mul_mat(a, th, y)
    assert maxdiff(vhy, MAT[f"{key}/vh_y"]) < TOL
    assert maxdiff(orc.grad_of_matrix_dot_product(a, th, x, vhy), MAT[f"{key}/grad"]) < TOL


@pytest.mark.parametrize("key", [str(k) for k in OBJ["names"] if str(k).startswith("surmax_")])
def test_sur_max_sequence(key):
    a = ansatz_from(OBJ, key)
    th, dth = OBJ[f"{key}/thetas"], OBJ[f"{key}/dthetas"]
    o = orc.SurMaxOracle(a, OBJ[f"{key}/target"], max_flips=1, front_layer=False)
    seq = []
    f0 = o.objective(th); seq.append((f0, o.max_no, o.weight))
    g0 = o.gradient(th); seq.append((o.fobj, o.max_no, o.weight))
    hs0 = o.hs.copy()
    f1 = o.objective(th + dth); seq.append((f1, o.max_no, o.weight))
    g1 = o.gradient(th + dth); seq.append((o.fobj, o.max_no, o.weight))
    g2 = o.gradient(th); seq.append((o.fobj, o.max_no, o.weight))
    assert abs(f0 - float(OBJ[f"{key}/f0"])) < TOL and abs(f1 - float(OBJ[f"{key}/f1"])) < TOL
    assert maxdiff(hs0, OBJ[f"{key}/hs0"]) < TOL and maxdiff(o.hs, OBJ[f"{key}/hs_last"]) < TOL
    for g, name in ((g0, "g0"), (g1, "g1"), (g2, "g2")):
        assert g.dtype == np.float64 and maxdiff(g, OBJ[f"{key}/{name}"]) < TOL
    assert maxdiff(np.asarray(seq, float), OBJ[f"{key}/seq"]) < TOL


@pytest.mark.parametrize("key", [str(k) for k in OBJ["names"] if str(k).startswith("sketch_")])
def test_sketching_objective(key):
    a = ansatz_from(OBJ, key)
    u = OBJ[f"{key}/target"]
    f, g = orc.sketching_objective_and_gradient(a, OBJ[f"{key}/thetas"], np.eye(u.shape[0], dtype=complex), u)
    assert abs(f - float(OBJ[f"{key}/fobj"])) < TOL and maxdiff(g, OBJ[f"{key}/grad"]) < TOL


@pytest.mark.parametrize("key", [str(k) for k in MPS["names"]])
def test_mps(key):
    a, b = mps_from(MPS, key, "a"), mps_from(MPS, key, "b")
    assert maxdiff(orc.mps_to_vector(a), MPS[f"{key}/vec_a"]) < TOL
    assert maxdiff(orc.mps_to_vector(b), MPS[f"{key}/vec_b"]) < TOL
    assert abs(orc.mps_dot(a, b) - complex(MPS[f"{key}/dot_ab"])) < TOL
    assert abs(orc.mps_dot(a, a) - complex(MPS[f"{key}/dot_aa"])) < TOL
    # reference test_mps.py:83: mps_dot == vdot of the dense states
    assert abs(orc.mps_dot(a, b) - np.vdot(MPS[f"{key}/vec_a"], MPS[f"{key}/vec_b"])) < TOL


def test_layout_generators_and_properties():
    rng = np.random.default_rng(5)
    # gradient == central finite differences of <V x|y>
    a = orc.Ansatz(4, "cp", orc.spin_blocks(4, 5))
    th = orc.rand_thetas(a.num_thetas, rng)
    x, y = orc.rand_state(4, rng), orc.rand_state(4, rng)
    g = orc.grad_of_dot_product(a, th, x, orc.v_dagger_mul_vec(a, th, y))
    for t in range(a.num_thetas):
        e = np.zeros_like(th); e[t] = 1e-6
        fd = (np.vdot(orc.v_mul_vec(a, th + e, x), y) - np.vdot(orc.v_mul_vec(a, th - e, x), y)) / 2e-6
        assert abs(fd - g[t]) < 1e-8
    tb = orc.trotter_blocks(5, 2)
    assert tb.shape == (2, 24) and np.all(tb[0, 0::3] == tb[1, 0::3] + 1)
    assert np.array_equal(orc.cyclic_spin_blocks(4, 4), np.array([[0, 2, 1, 3], [1, 3, 2, 0]]))


PRIM = load("primitives.npz")


def test_primitives_golden():
    """The oracle's elementary gates and inner products against the reference's own single-gate functions
    (tests/golden/make_golden.py::gen_primitives).  State-vector positions are big-endian: qubit = n-1-pos."""
    n = int(PRIM["n"])
    vec, w, z, angles, pairs = PRIM["vec"], PRIM["w"], PRIM["z"], PRIM["angles"], PRIM["pairs"]
    for name, fn in (("rx", orc.rx), ("ry", orc.ry), ("rz", orc.rz)):
        for i, a in enumerate(angles):
            for pos in range(n):
                v = vec.copy()
                fn(v, 1 << (n - 1 - pos), float(a))
                assert maxdiff(v, PRIM[f"sv/{name}"][i, pos]) < TOL
    for name, fn in (("dot_x", orc.dot_x), ("dot_y", orc.dot_y), ("dot_z", orc.dot_z)):
        for pos in range(n):
            assert abs(fn(w, z, 1 << (n - 1 - pos)) - PRIM[f"sv/{name}"][pos]) < TOL
    for name in ("cx", "cz", "cp"):
        for i, (c, t) in enumerate(pairs):
            v = vec.copy()
            orc._entangle(v, 1 << (n - 1 - c), 1 << (n - 1 - t), name, 0.83)
            assert maxdiff(v, PRIM[f"sv/{name}"][i]) < TOL
    for k in (3, 1 << n):
        m, wm, zm = PRIM[f"mat{k}/m"], PRIM[f"mat{k}/w"], PRIM[f"mat{k}/z"]
        for name, fn in (("rx", orc.rx), ("ry", orc.ry), ("rz", orc.rz)):
            for q in range(n):
                v = m.copy().ravel()
                fn(v, k << q, 0.37)
                assert maxdiff(v.reshape(m.shape), PRIM[f"mat{k}/{name}"][q]) < TOL
        for name, fn in (("x_dot", orc.dot_x), ("y_dot", orc.dot_y), ("z_dot", orc.dot_z)):
            for q in range(n):
                assert abs(fn(wm.ravel(), zm.ravel(), k << q) - PRIM[f"mat{k}/{name}"][q]) < 10 * TOL
        for i, (c, t) in enumerate(pairs):
            assert abs(orc.dot_cp11(wm.ravel(), zm.ravel(), k << c, k << t) - PRIM[f"mat{k}/derv_cphase"][i]) < 10 * TOL
            for name in ("cx", "cz", "cp"):
                v = m.copy().ravel()
                orc._entangle(v, k << c, k << t, name, 0.83)
                assert maxdiff(v.reshape(m.shape), PRIM[f"mat{k}/{name}"][i]) < TOL
