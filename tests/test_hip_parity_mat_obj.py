"""GPU parity: matrix path and the objective objects vs golden vectors and the oracle."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import FAMILIES, FAMILY_ENV, TOL, ansatz_from, load, maxdiff

pytestmark = pytest.mark.gpu

MAT = load("matrix.npz")
OBJ = load("objectives.npz")


def make_circ(a):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz

    return TrotterAnsatz(a.n, a.blocks, second_order=a.second_order) if a.trotter else ParametricCircuit(a.n, a.entangler, a.blocks)


@pytest.mark.parametrize("key", [str(k) for k in MAT["names"] if not str(k).endswith("_cd")])
def test_golden_matrix(key):
    import aqc_research_amd.core_op_matrix as com

    a = ansatz_from(MAT, key)
    circ = make_circ(a)
    th = MAT[f"{key}/thetas"]
    x, y = MAT[f"{key}/x"].copy(), MAT[f"{key}/y"].copy()
    work = np.zeros_like(x)
    m = x.copy()
    assert com.v_mul_mat(circ, th, m, work) is m and maxdiff(m, MAT[f"{key}/v_x"]) < TOL
    vhy = com.v_dagger_mul_mat(circ, th, y.copy(), work)
    assert maxdiff(vhy, MAT[f"{key}/vh_y"]) < TOL
    g = com.grad_of_matrix_dot_product(circ, th, x, vhy, work)
    assert maxdiff(g, MAT[f"{key}/grad"]) < TOL


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("n,ent,depth,k", [(6, "cx", 15, 64), (7, "cp", 12, 5), (8, "cz", 20, 32), (9, "cx", 24, 512)])
def test_matrix_vs_oracle(n, ent, depth, k, family, monkeypatch):
    import aqc_research_amd.core_op_matrix as com
    from aqc_research_amd.engine import HipContext

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])
    HipContext._cache.clear()  # function-level entry points cache their workspace per structure

    rng = np.random.default_rng(n * 31 + k)
    blocks = np.stack([rng.permutation(n)[:2] for _ in range(depth)], axis=1).astype(np.int64)
    a = orc.Ansatz(n, ent, blocks)
    circ = make_circ(a)
    th = orc.rand_thetas(a.num_thetas, rng)
    d = 1 << n
    x = rng.standard_normal((d, k)) + 1j * rng.standard_normal((d, k))
    y = rng.standard_normal((d, k)) + 1j * rng.standard_normal((d, k))
    x /= np.linalg.norm(x)   # unit Frobenius norm: the inner products are O(1), so 1e-10 absolute is the north-star bar
    y /= np.linalg.norm(y)
    vhy = com.v_dagger_mul_mat(circ, th, y.copy(), None)
    ref = orc.v_dagger_mul_mat(a, th, y)
    assert maxdiff(vhy, ref) < TOL
    g = com.grad_of_matrix_dot_product(circ, th, x, vhy, None)
    assert maxdiff(g, orc.grad_of_matrix_dot_product(a, th, x, ref)) < TOL
    back = com.v_mul_mat(circ, th, vhy.copy(), None)
    assert maxdiff(back, y) < TOL


@pytest.mark.parametrize("key", [str(k) for k in OBJ["names"] if str(k).startswith("surmax_")])
def test_sur_max_golden_sequence(key):
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    a = ansatz_from(OBJ, key)
    circ = make_circ(a)
    user = dict(num_qubits=a.n, max_flips=1, enable_optim_stats=True, verbose=0, maxiter=10, num_simulations=2)
    obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=False)
    obj.set_target(OBJ[f"{key}/target"].copy())
    th, dth = OBJ[f"{key}/thetas"], OBJ[f"{key}/dthetas"]
    seq = []
    f0 = obj.objective(th); seq.append((f0, obj._max_no, obj._weight))
    g0 = obj.gradient(th); seq.append((obj._fobj, obj._max_no, obj._weight))
    hs0 = obj._hs.copy()
    f1 = obj.objective(th + dth); seq.append((f1, obj._max_no, obj._weight))
    g1 = obj.gradient(th + dth); seq.append((obj._fobj, obj._max_no, obj._weight))
    g2 = obj.gradient(th); seq.append((obj._fobj, obj._max_no, obj._weight))
    assert isinstance(f0, float) and abs(f0 - float(OBJ[f"{key}/f0"])) < TOL and abs(f1 - float(OBJ[f"{key}/f1"])) < TOL
    assert maxdiff(hs0, OBJ[f"{key}/hs0"]) < TOL and maxdiff(obj._hs, OBJ[f"{key}/hs_last"]) < TOL
    for g, name in ((g0, "g0"), (g1, "g1"), (g2, "g2")):
        assert g.dtype == np.float64 and g.flags.c_contiguous and maxdiff(g, OBJ[f"{key}/{name}"]) < TOL
    assert maxdiff(np.asarray(seq, float), OBJ[f"{key}/seq"]) < TOL
    assert maxdiff(obj.statistics["fobj"], OBJ[f"{key}/stats_fobj"]) < 1e-6
    assert obj.num_states == a.n + 1 and obj.num_thetas == a.num_thetas and 0 <= obj.fidelity <= 1


@pytest.mark.parametrize("key", [str(k) for k in OBJ["names"] if str(k).startswith("sketch_")])
def test_sketching_golden(key):
    from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx

    a = ansatz_from(OBJ, key)
    objv = SketchingObjectiveEx(make_circ(a), FullRangeSketchingVectors(OBJ[f"{key}/target"].copy()))
    th = OBJ[f"{key}/thetas"]
    f = objv.objective(th)
    g = objv.gradient(th)
    assert abs(f - float(OBJ[f"{key}/fobj"])) < TOL and maxdiff(g, OBJ[f"{key}/grad"]) < TOL
    assert objv.num_iterations == 1  # gradient served from the cache (sk_core.py:258-263)
    g2 = objv.gradient(th + 1e-3)
    assert objv.num_iterations == 2 and maxdiff(g2, g) > 0
    assert objv.optim_results["num_iters"] == 2


def test_sur_max_neel_and_stoppers():
    """Neel-state preparation as a bit mask (trotter.py:389-398) and stopper propagation."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n = 8
    rng = np.random.default_rng(3)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 2), second_order=True)
    neel = sum(1 << q for q in range(0, n, 2))
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda nq: neel, enable_optim_stats=False)
    obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    target = orc.rand_state(n, rng)
    obj.set_target(target)
    o = orc.SurMaxOracle(circ, target, 1, None, True, base_index=neel)
    th = 0.2 * orc.rand_thetas(circ.num_thetas, rng)
    for step in range(3):
        assert abs(obj.objective(th) - o.objective(th)) < TOL
        assert maxdiff(obj.gradient(th), o.gradient(th)) < TOL
        th = th + 0.05 * rng.standard_normal(th.size)

    class Stop:
        def check(self, **kw):
            raise StopIteration("stop now")

    obj.set_status_trackers(timeout=None, stopper=Stop())
    with pytest.raises(StopIteration):
        obj.gradient(th)


MATCD = [str(k) for k in MAT["names"] if str(k).endswith("_cd")]


@pytest.mark.parametrize("key", MATCD)
def test_golden_coord_descent(key):
    """Two consecutive sweeps vs the reference (sequential Newton steps amplify rounding, hence 1e-8)."""
    import aqc_research_amd.core_op_matrix as com

    a = ansatz_from(MAT, key)
    circ = make_circ(a)
    th = MAT[f"{key}/thetas"].copy()
    target = MAT[f"{key}/target"].copy()
    f1 = com.coord_descent_single_sweep(circ, th, target, None)
    assert maxdiff(th, MAT[f"{key}/thetas_1"]) < 1e-9 and abs(f1 - float(MAT[f"{key}/fobj_1"])) < 1e-9
    f2 = com.coord_descent_single_sweep(circ, th, target, None)
    assert maxdiff(th, MAT[f"{key}/thetas_2"]) < 1e-8 and abs(f2 - float(MAT[f"{key}/fobj_2"])) < 1e-8
    assert f2 <= f1 + 1e-12  # descent


def test_coord_descent_rejects_cp():
    import aqc_research_amd.core_op_matrix as com
    from aqc_research_amd import ParametricCircuit

    circ = ParametricCircuit(2, "cp", np.array([[0], [1]]))
    with pytest.raises(NotImplementedError):
        com.coord_descent_single_sweep(circ, np.zeros(circ.num_thetas), np.eye(4, dtype=complex), None)


@pytest.mark.parametrize("kind", ["rand", "alt", "eigen"])
def test_sketched_objective_generators(kind):
    """Sketched AQC (k < d columns): the generators draw from np.random in the reference's order, so replaying the
    seed on the CPU gives the same (X, Y); objective and gradient must match the oracle on them."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.model_sketching.sk_core import SketchingObjectiveEx, skvecs_generator

    n, k = 5, 8
    rng = np.random.default_rng(8)
    circ = ParametricCircuit(n, "cz", orc.spin_blocks(n, 12))
    d = 1 << n
    u = np.ascontiguousarray(np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0])
    th = orc.rand_thetas(circ.num_thetas, rng)

    np.random.seed(321)
    gen = skvecs_generator(kind, k, u)
    f, g = SketchingObjectiveEx(circ, gen).objective_and_gradient(th)

    np.random.seed(321)                       # replay the draws with plain NumPy
    if kind == "rand":
        x = np.linalg.qr(np.random.rand(d, k) + 1j * np.random.rand(d, k))[0]
    elif kind == "alt":
        idx = np.random.permutation(d)[:k]
        x = np.zeros((d, k), complex); x[idx, np.arange(k)] = 1
    else:
        omega = np.random.randn(d, k) * 1j
        omega = omega + np.random.randn(d, k)
        x = np.linalg.qr(orc.v_dagger_mul_mat(circ, th, omega) - u.conj().T @ omega)[0]
    y = u @ x
    fr, gr = orc.sketching_objective_and_gradient(circ, th, x, y)
    assert abs(f - fr) < 1e-9 and maxdiff(g, gr) < 1e-9
    assert maxdiff(np.conj(x.T) @ x, np.eye(k)) < 1e-12


def test_zgemm_entry_point():
    from aqc_research_amd.engine import zgemm

    rng = np.random.default_rng(4)
    a = rng.standard_normal((70, 33)) + 1j * rng.standard_normal((70, 33))
    b = rng.standard_normal((33, 130)) + 1j * rng.standard_normal((33, 130))
    assert maxdiff(zgemm(a, b), a @ b) < 1e-12
    c = rng.standard_normal((70, 5)) + 1j * rng.standard_normal((70, 5))
    assert maxdiff(zgemm(a, c, conj_trans_a=True), a.conj().T @ c) < 1e-12


@pytest.mark.parametrize("ent", ["cx", "cz", "cp"])
def test_matrix_gradient_equals_parameter_shift(ent):
    """The reference's own check of grad_of_matrix_dot_product (test_core_op_matrix.py:114-140,305-336): every
    rotation angle obeys df/dt = (f(t + pi/2) - f(t - pi/2)) / 2 ... in the reference's form: shift pi, scale 1/4
    for half-angle rotations; the CPhase angle: shift pi/2, scale 1/2 -- evaluated with the HIP path itself."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd import core_op_matrix as com

    n, k = 4, 5
    rng = np.random.default_rng(31)
    blocks = np.stack([rng.permutation(n)[:2] for _ in range(6)], axis=1).astype(np.int64)
    circ = ParametricCircuit(n, ent, blocks)
    th = orc.rand_thetas(circ.num_thetas, rng)
    x = rng.standard_normal((1 << n, k)) + 1j * rng.standard_normal((1 << n, k))
    y = rng.standard_normal((1 << n, k)) + 1j * rng.standard_normal((1 << n, k))

    def f(t):   # <V(t) X | Y>_F
        return np.vdot(com.v_mul_mat(circ, t, x.copy()), y)

    g = com.grad_of_matrix_dot_product(circ, th, x.copy(), com.v_dagger_mul_mat(circ, th, y.copy()))
    tpb = 5 if ent == "cp" else 4
    for t in range(circ.num_thetas):
        is_cp_angle = ent == "cp" and t >= 3 * n and (t - 3 * n) % tpb == 4
        e = np.zeros_like(th)
        if is_cp_angle:     # f is a*e^{i t} + b in the CPhase angle: derivative = (f(t + pi/2) - f(t - pi/2)) / 2
            e[t] = np.pi / 2
            shift = (f(th + e) - f(th - e)) / 2
        else:               # half-angle rotation: f = a cos(t/2) + b sin(t/2): derivative = (f(t + pi) - f(t - pi)) / 4
            e[t] = np.pi
            shift = (f(th + e) - f(th - e)) / 4
        assert abs(shift - g[t]) < 1e-9 * max(1.0, abs(g[t]))
