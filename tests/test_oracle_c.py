"""Pins the C restatement (oracle/aqc_ref.c) to the reference's golden vectors and to the NumPy
restatement.  CPU only."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from oracle import aqc_ref as ref
from tests.helpers import TOL, ansatz_from, load, maxdiff

SV = load("state_vector.npz")
MAT = load("matrix.npz")


@pytest.mark.parametrize("key", [str(k) for k in SV["names"]])
def test_state_vector_golden(key):
    a = ansatz_from(SV, key)
    th, x, y = SV[f"{key}/thetas"], SV[f"{key}/x"], SV[f"{key}/y"]
    assert maxdiff(ref.v_mul_vec(a, th, x), SV[f"{key}/v_x"]) < TOL
    vhy = ref.v_dagger_mul_vec(a, th, y)
    assert maxdiff(vhy, SV[f"{key}/vh_y"]) < TOL
    assert maxdiff(ref.grad_of_dot_product(a, th, x, vhy), SV[f"{key}/grad_full"]) < TOL
    br = tuple(int(v) for v in SV[f"{key}/block_range"])
    assert maxdiff(ref.grad_of_dot_product(a, th, x, vhy, br, False), SV[f"{key}/grad_part"]) < TOL


@pytest.mark.parametrize("key", [str(k) for k in MAT["names"] if not str(k).endswith("_cd")])
def test_matrix_golden(key):
    a = ansatz_from(MAT, key)
    th, x, y = MAT[f"{key}/thetas"], MAT[f"{key}/x"], MAT[f"{key}/y"]
    assert maxdiff(ref.v_mul_mat(a, th, x), MAT[f"{key}/v_x"]) < TOL
    vhy = ref.v_dagger_mul_mat(a, th, y)
    assert maxdiff(vhy, MAT[f"{key}/vh_y"]) < TOL
    assert maxdiff(ref.grad_of_matrix_dot_product(a, th, x, vhy), MAT[f"{key}/grad"]) < TOL


@pytest.mark.parametrize("ent", ["cx", "cz", "cp"])
def test_against_numpy_restatement(ent):
    rng = np.random.default_rng(11)
    n = 9
    a = orc.Ansatz(n, ent, orc.spin_blocks(n, 17))
    th = orc.rand_thetas(a.num_thetas, rng)
    x, y = orc.rand_state(n, rng), orc.rand_state(n, rng)
    assert maxdiff(ref.v_mul_vec(a, th, x), orc.v_mul_vec(a, th, x)) < 1e-13
    vhy = orc.v_dagger_mul_vec(a, th, y)
    assert maxdiff(ref.v_dagger_mul_vec(a, th, y), vhy) < 1e-13
    assert maxdiff(ref.grad_of_dot_product(a, th, x, vhy, (3, 11), False), orc.grad_of_dot_product(a, th, x, vhy, (3, 11), False)) < 1e-13


def test_eval_batch_threads_and_trotter():
    rng = np.random.default_rng(12)
    n = 8
    a = orc.Ansatz(n, "cx", orc.trotter_blocks(n, 2), True, True)
    thetas = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(5)])
    y = orc.rand_state(n, rng)
    idx = 0b01010101
    hs1, g1 = ref.eval_batch(a, thetas, y, idx, threads=1)
    hs4, g4 = ref.eval_batch(a, thetas, y, idx, threads=4)
    assert np.array_equal(hs1, hs4) and np.array_equal(g1, g4)  # lanes are independent: bitwise equal
    x = np.zeros(1 << n, complex)
    x[idx] = 1
    for b in range(5):
        vhy = orc.v_dagger_mul_vec(a, thetas[b], y)
        assert abs(hs1[b] - vhy[idx]) < 1e-13
        assert maxdiff(g1[b], orc.grad_of_dot_product(a, thetas[b], x, vhy)) < 1e-13


def test_rejects_bad_arguments():
    a = orc.Ansatz(3, "cx", np.array([[0, 1], [1, 2]]))
    with pytest.raises(ValueError):
        ref.v_mul_vec(a, np.zeros(3), np.zeros(8))
    with pytest.raises(ValueError):
        ref.v_mul_vec(a, np.zeros(a.num_thetas), np.zeros(4))


def test_c_coordinate_descent_matches_the_golden_sweeps_and_the_numpy_restatement():
    """aqc_ref_cd_sweeps (the CPU baseline of bench.py --workload cd5_cyc180) against the NumPy restatement, which the golden
    fixture of the reference's two consecutive sweeps pins (tests/test_oracle_golden.py)."""
    import numpy as np

    from oracle import aqc_oracle as orc
    from oracle import aqc_ref as cref

    rng = np.random.default_rng(5)
    for ent, n, depth in (("cx", 4, 9), ("cz", 3, 7)):
        a = orc.Ansatz(n, ent, orc.spin_blocks(n, depth), False, False)
        d = 1 << n
        th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(3)])
        us = np.stack([np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0] for _ in range(3)])
        got, f = cref.coord_descent_sweeps(a, th, us, 2, threads=2)
        for b in range(3):
            t1, f1 = orc.coord_descent_single_sweep(a, th[b], us[b])
            t2, f2 = orc.coord_descent_single_sweep(a, t1, us[b])
            assert abs(f[b, 0] - f1) < 1e-9 and abs(f[b, 1] - f2) < 1e-8 and np.abs(got[b] - t2).max() < 1e-8
