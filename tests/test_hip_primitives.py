"""GPU: the gate-level functions of core_operations / core_op_matrix against the
reference's own outputs (tests/golden/primitives.npz, gate2x2.npz)."""
import numpy as np
import pytest

from tests.helpers import TOL, load, maxdiff

pytestmark = pytest.mark.gpu

PRIM = load("primitives.npz")
G22 = load("gate2x2.npz")


def test_state_vector_primitives():
    from aqc_research_amd import core_operations as cop

    n = int(PRIM["n"])
    vec, w, z, angles, pairs = PRIM["vec"], PRIM["w"], PRIM["z"], PRIM["angles"], PRIM["pairs"]
    tmp = np.zeros_like(vec)
    for name, fn in (("rx", cop.rx_mul_vec), ("ry", cop.ry_mul_vec), ("rz", cop.rz_mul_vec)):
        for i, a in enumerate(angles):
            for pos in range(n):
                v = vec.copy()
                assert fn(n, pos, float(a), v, tmp) is v        # in place, returns its argument
                assert maxdiff(v, PRIM[f"sv/{name}"][i, pos]) < TOL
    w0, z0 = w.copy(), z.copy()
    for name, fn in (("dot_x", cop.dot_x), ("dot_y", cop.dot_y), ("dot_z", cop.dot_z)):
        for pos in range(n):
            assert abs(fn(n, pos, w, z, tmp) - PRIM[f"sv/{name}"][pos]) < TOL
    assert np.array_equal(w, w0) and np.array_equal(z, z0)
    for name, fn in (("cx", cop.cx_mul_vec), ("cz", cop.cz_mul_vec), ("cp", cop.cp_mul_vec)):
        for i, (c, t) in enumerate(pairs):
            v = vec.copy()
            assert fn(n, int(c), int(t), 0.83, v, tmp) is v
            assert maxdiff(v, PRIM[f"sv/{name}"][i]) < TOL
    for i, (c, t) in enumerate(pairs):
        out = np.full_like(vec, 7.0)
        v = vec.copy()
        assert cop.derv_cphase_mul_vec(n, int(c), int(t), 0.83, v, out) is out
        assert maxdiff(out, PRIM["sv/derv_cp"][i]) < TOL and np.array_equal(v, vec)
        mats = PRIM["block_mats"]
        for dagger, key in ((False, "sv/block"), (True, "sv/block_dagger")):
            v = vec.copy()
            cop.block_mul_vec(n, int(c), int(t), mats[0], mats[1], mats[2], v, np.zeros((2, v.size), complex), dagger)
            assert maxdiff(v, PRIM[key][i]) < 10 * TOL
    for pos in range(n):
        assert maxdiff(cop.proj00_mul_vec(n, pos, vec.copy()), PRIM["sv/proj00"][pos]) == 0.0
        assert maxdiff(cop.proj11_mul_vec(n, pos, vec.copy()), PRIM["sv/proj11"][pos]) == 0.0
    with pytest.raises(ValueError):
        cop.rx_mul_vec(n, n, 0.1, vec.copy(), tmp)
    with pytest.raises(ValueError):
        cop.cx_mul_vec(n, 1, 1, 0.0, vec.copy(), tmp)


def test_gate2x2_degenerate_cases():
    """The 8 degenerate 2x2 gates of test_core_operations.py:162-178 (+ a random one), both result placements."""
    from aqc_research_amd import core_operations as cop

    n, vec = int(G22["n"]), G22["vec"]
    for i, g in enumerate(G22["gates"]):
        for q in range(n):
            v, out = vec.copy(), np.zeros_like(vec)
            assert cop.gate2x2_mul_vec(n, cop.bit2bit_transform(n, q), g, v, out, False) is out
            assert maxdiff(out, G22["outs"][i, q]) < TOL and np.array_equal(v, vec)
            assert cop.gate2x2_mul_vec(n, cop.bit2bit_transform(n, q), g, v, out, True) is v
            assert maxdiff(v, G22["outs"][i, q]) < TOL


@pytest.mark.parametrize("k", [3, 32])
def test_matrix_primitives(k):
    from aqc_research_amd import core_op_matrix as com

    n, pairs = int(PRIM["n"]), PRIM["pairs"]
    m, wm, zm = PRIM[f"mat{k}/m"], PRIM[f"mat{k}/w"], PRIM[f"mat{k}/z"]
    ws = np.zeros_like(m)
    for name, fn in (("rx", com.rx_mul_mat), ("ry", com.ry_mul_mat), ("rz", com.rz_mul_mat)):
        for q in range(n):
            a = m.copy()
            assert fn(0.37, q, a, ws) is a
            assert maxdiff(a, PRIM[f"mat{k}/{name}"][q]) < TOL
    for q in range(n):
        assert maxdiff(com.gate2x2_mul_mat(q, PRIM["block_mats"][0], m.copy(), ws), PRIM[f"mat{k}/gate2x2"][q]) < 10 * TOL
    for name, fn in (("cx", com.cx_mul_mat), ("cz", com.cz_mul_mat), ("cp", com.cp_mul_mat)):
        for i, (c, t) in enumerate(pairs):
            assert maxdiff(fn(int(c), int(t), 0.83, m.copy(), ws), PRIM[f"mat{k}/{name}"][i]) < TOL
    for name, fn in (("x_dot", com.x_dot_mat), ("y_dot", com.y_dot_mat), ("z_dot", com.z_dot_mat)):
        for q in range(n):
            assert abs(fn(q, wm, zm, ws) - PRIM[f"mat{k}/{name}"][q]) < 10 * TOL
    for i, (c, t) in enumerate(pairs):
        assert abs(com.derv_cphase(int(c), int(t), wm, zm, ws) - PRIM[f"mat{k}/derv_cphase"][i]) < 10 * TOL
