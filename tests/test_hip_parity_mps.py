"""GPU parity: MPS helpers and the MPS-dot gradient vs golden vectors and the oracle."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import TOL, load, maxdiff, mps_from

pytestmark = pytest.mark.gpu

MPS = load("mps.npz")


@pytest.mark.parametrize("key", [str(k) for k in MPS["names"]])
def test_golden_mps(key):
    import aqc_research_amd.mps_operations as mpsop

    a, b = mps_from(MPS, key, "a"), mps_from(MPS, key, "b")
    assert mpsop.check_mps(a) and mpsop.check_mps(b)
    assert maxdiff(mpsop.mps_to_vector(a), MPS[f"{key}/vec_a"]) < TOL
    assert maxdiff(mpsop.mps_to_vector(b), MPS[f"{key}/vec_b"]) < TOL
    assert abs(mpsop.mps_dot(a, b) - complex(MPS[f"{key}/dot_ab"])) < TOL
    assert abs(mpsop.mps_dot(a, a) - complex(MPS[f"{key}/dot_aa"])) < TOL


@pytest.mark.parametrize("n,chi", [(8, 16), (12, 64), (16, 16), (16, 256)])
def test_mps_large_vs_oracle(n, chi):
    """test_mps.py:60,83: mps_dot == vdot of the dense states, tolerance scaled with n."""
    import aqc_research_amd.mps_operations as mpsop

    rng = np.random.default_rng(n * 1000 + chi)
    a, b = orc.random_mps(n, chi, rng), orc.random_mps(n, chi, rng)
    va, vb = mpsop.mps_to_vector(a), mpsop.mps_to_vector(b)
    tol = TOL * 2 ** max(n - 10, 0)
    assert maxdiff(va, orc.mps_to_vector(a)) < TOL
    assert abs(np.linalg.norm(va) - 1) < 1e-9
    d = mpsop.mps_dot(a, b)
    assert abs(d - np.vdot(va, vb)) < tol and abs(d - orc.mps_dot(a, b)) < tol
    # exact re-encoding: dense -> MPS -> dense is the identity, bit for bit
    m = mpsop.vector_to_exact_mps(va)
    assert mpsop.check_mps(m) and np.array_equal(mpsop.mps_to_vector(m), va)


@pytest.mark.parametrize("n,ent,kind", [(5, "cp", "generic"), (6, "cx", "trotter2"), (7, "cz", "generic"), (10, "cx", "trotter1")])
def test_fast_dot_gradient(n, ent, kind):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz
    import aqc_research_amd.mps_operations as mpsop
    from aqc_research_amd.mps_dot_objective import fast_dot_gradient

    rng = np.random.default_rng(n)
    if kind == "generic":
        blocks = np.stack([rng.permutation(n)[:2] for _ in range(2 * n)], axis=1).astype(np.int64)
        circ = ParametricCircuit(n, ent, blocks)
    else:
        circ = TrotterAnsatz(n, orc.trotter_blocks(n, 2), second_order=(kind == "trotter2"))
    th = orc.rand_thetas(circ.num_thetas, rng)
    lvec = orc.random_mps(n, 1, rng)          # low-entangled lhs (product state)
    phi = orc.random_mps(n, 4, rng)
    vh_phi = mpsop.v_dagger_mul_mps(circ, th, phi)
    assert mpsop.check_mps(vh_phi)
    ref_vh = orc.v_dagger_mul_vec(circ, th, orc.mps_to_vector(phi))
    assert maxdiff(mpsop.mps_to_vector(vh_phi), ref_vh) < TOL
    # V V^H = identity on MPS (test_mps.py V V^H check)
    back = mpsop.v_mul_mps(circ, th, vh_phi)
    assert maxdiff(mpsop.mps_to_vector(back), orc.mps_to_vector(phi)) < TOL
    for br, front in ((None, True), ((1, circ.num_blocks - 1), False)):
        g = fast_dot_gradient(circ, th, lvec, vh_phi, trunc_thr=1e-16, block_range=br, front_layer=front)
        ref = orc.grad_of_dot_product(circ, th, orc.mps_to_vector(lvec), ref_vh, br, front)
        assert maxdiff(g, ref) < TOL


def test_mps_objective_matches_sv_oracle():
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_fast_mps_trotter import SpSurrogateObjectiveFastMpsTrotter

    n = 9
    rng = np.random.default_rng(99)
    circ = TrotterAnsatz(n, orc.trotter_blocks(n, 2), second_order=True)
    target = orc.random_mps(n, 8, rng)
    user = dict(num_qubits=n, max_flips=1, trunc_thr=1e-16, enable_optim_stats=False)
    obj = SpSurrogateObjectiveFastMpsTrotter(user_parameters=user, circ=circ, layer_range=(0, 1))
    obj.set_target(target)
    o = orc.SurMaxOracle(circ, orc.mps_to_vector(target), 1, (0, circ.bpl), True)
    th = 0.3 * orc.rand_thetas(circ.num_thetas, rng)
    for _ in range(3):
        assert abs(obj.objective(th) - o.objective(th)) < TOL
        assert maxdiff(obj.gradient(th), o.gradient(th)) < TOL
        th = th + 0.1 * rng.standard_normal(th.size)


def test_single_gates_on_mps_dense_semantics():
    """mps_dot_objective.py:245-516 (P2): gate on MPS == gate on the dense state (what test_mps.py:57-199 pins);
    1-qubit gates keep the bond dimensions."""
    from aqc_research_amd import mps_dot_objective as mdo
    from aqc_research_amd.mps_operations import mps_to_vector

    n, chi = 6, 5
    rng = np.random.default_rng(66)
    a, b = orc.random_mps(n, chi, rng), orc.random_mps(n, chi, rng)
    va, vb = orc.mps_to_vector(a), orc.mps_to_vector(b)
    for q in range(n):
        for name, rot in (("rx", orc.rx), ("ry", orc.ry), ("rz", orc.rz)):
            out = getattr(mdo, f"{name}_mul_mps")(0.71, q, a)
            ref = va.copy()
            rot(ref, 1 << q, 0.71)
            assert maxdiff(mps_to_vector(out), ref) < TOL
            assert [g[0].shape for g in out[0]] == [g[0].shape for g in a[0]]
        for name, dot in (("dot_x", orc.dot_x), ("dot_y", orc.dot_y), ("dot_z", orc.dot_z)):
            assert abs(getattr(mdo, name)(q, a, b) - dot(va, vb, 1 << q)) < TOL
    for c, t in ((0, 1), (4, 2), (5, 0)):
        for name, ent in (("cx", "cx"), ("cz", "cz"), ("cp", "cp")):
            out = getattr(mdo, f"{name}_mul_mps")(0.4, c, t, a)
            ref = va.copy()
            orc._entangle(ref, 1 << c, 1 << t, ent, 0.4)
            assert maxdiff(mps_to_vector(out), ref) < TOL
