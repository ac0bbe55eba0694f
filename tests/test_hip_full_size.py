"""GPU parity at the FULL sizes of BASELINE.json's configurations, through the front doors the reference offers:
config 1 (5-qubit cyclic_spin AQC, 180 blocks, T = 735), config 3 (16 qubits, 40 blocks, MPS inputs through
``fast_dot_gradient`` / ``v_dagger_mul_mps``, chi in {16, 64, 256}) and config 5 (10-qubit full-unitary AQC,
1024 x 1024 target, ``SketchingObjectiveEx``), each against the compiled CPU restatement (oracle/aqc_ref.c) on
identical inputs at the north-star tolerance 1e-10 (absolute, complex fp64), on every kernel family."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from oracle import aqc_ref as cref
from tests.helpers import FAMILIES, FAMILY_ENV, TOL, maxdiff

pytestmark = pytest.mark.gpu

def _family(monkeypatch, family):
    from aqc_research_amd.engine import HipContext

    monkeypatch.setenv("AQC_KERNEL_FAMILY", FAMILY_ENV[family])
    HipContext._cache.clear()   # the function-level entry points cache their workspace per structure


def _su(u):
    """U / det(U)^(1/d): the special-unitary form the reference compiles to (target_generator.py:269-288)."""
    d = u.shape[0]
    return u / np.exp(1j * np.angle(np.linalg.det(u)) / d)


@pytest.mark.parametrize("family", FAMILIES)
def test_config5_full_unitary_aqc_1024(family, monkeypatch):
    """core_op_matrix.py:645 / sk_core.py:167 at n = 10, d = k = 1024, spin layout L = 40 (T = 190)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx

    _family(monkeypatch, family)
    n, d = 10, 1024
    rng = np.random.default_rng(510)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
    assert circ.num_thetas == 190
    target = _su(np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0])
    objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(target))
    eye = np.eye(d, dtype=np.complex128)
    for _ in range(2):   # the second evaluation re-uses the resident X = I and target
        th = orc.rand_thetas(circ.num_thetas, rng)
        fobj, grad = objv.objective_and_gradient(th)
        vhy = cref.v_dagger_mul_mat(circ, th, target)
        cg = cref.grad_of_matrix_dot_product(circ, th, eye, vhy)
        assert abs(fobj - (1 - np.real(np.vdot(eye, vhy)) / d)) < TOL
        assert maxdiff(grad, -np.real(cg) / d) < TOL
    import aqc_research_amd.core_op_matrix as com

    vh = com.v_dagger_mul_mat(circ, th, target.copy(), None)   # function level, same size
    assert maxdiff(vh, vhy) < TOL
    g = com.grad_of_matrix_dot_product(circ, th, eye.copy(), vh, None)
    assert maxdiff(g / d, cg / d) < TOL


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("chi", [16, 64, 256])
def test_config3_mps_front_door_n16_l40(chi, family, monkeypatch):
    """mps_dot_objective.py:41 + mps_operations.py:349 at n = 16, 40 blocks (T = 208): |0> product-state lhs and a
    random Vidal-form target of bond dimension chi, trunc_thr = 1e-16, against the C oracle
    on the densified target -- the level the reference's own tests pin (test_mps_fast_dot_gradient.py:126-153)."""
    from aqc_research_amd import ParametricCircuit
    import aqc_research_amd.mps_operations as mpsop
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.mps_dot_objective import fast_dot_gradient

    _family(monkeypatch, family)
    n = 16
    rng = np.random.default_rng(3000 + chi)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 40))
    assert circ.num_thetas == 208
    th = orc.rand_thetas(circ.num_thetas, rng)
    zero = ([(np.ones((1, 1), complex), np.zeros((1, 1), complex)) for _ in range(n)], [np.ones(1) for _ in range(n - 1)])
    phi = orc.random_mps(n, chi, rng)
    dense = orc.mps_to_vector(phi)
    assert maxdiff(mpsop.mps_to_vector(phi), dense) < TOL
    vh_phi = mpsop.v_dagger_mul_mps(circ, th, phi, trunc_thr=1e-16)
    ref_vh = cref.v_dagger_mul_vec(circ, th, dense)
    assert maxdiff(mpsop.mps_to_vector(vh_phi), ref_vh) < TOL
    x = np.zeros(1 << n, complex)
    x[0] = 1
    assert abs(mpsop.mps_dot(zero, vh_phi) - ref_vh[0]) < TOL          # <0|V^H|phi>: the objective's overlap
    g = fast_dot_gradient(circ, th, zero, vh_phi, trunc_thr=1e-16)
    assert maxdiff(g, cref.grad_of_dot_product(circ, th, x, ref_vh)) < TOL
    br = (7, 31)
    g = fast_dot_gradient(circ, th, zero, vh_phi, trunc_thr=1e-16, block_range=br, front_layer=False)
    assert maxdiff(g, cref.grad_of_dot_product(circ, th, x, ref_vh, br, False)) < TOL


@pytest.mark.parametrize("family", FAMILIES)
def test_config1_cyclic_spin_180_blocks(family, monkeypatch):
    """docs/aqc.ipynb ansatz: n = 5, cyclic_spin, 180 blocks, T = 735, Haar-random SU(32) target."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx
    from scipy.stats import unitary_group

    _family(monkeypatch, family)
    n, d = 5, 32
    rng = np.random.default_rng(15)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "cyclic_spin", "full", 180))
    assert circ.num_thetas == 735
    target = _su(unitary_group.rvs(d, random_state=7))
    objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(target))
    eye = np.eye(d, dtype=np.complex128)
    for _ in range(3):
        th = orc.rand_thetas(circ.num_thetas, rng)
        fobj, grad = objv.objective_and_gradient(th)
        f0, g0 = orc.sketching_objective_and_gradient(circ, th, eye, target)
        vhy = cref.v_dagger_mul_mat(circ, th, target)
        cg = cref.grad_of_matrix_dot_product(circ, th, eye, vhy)
        assert abs(fobj - f0) < TOL and maxdiff(grad, g0) < TOL and maxdiff(grad, -np.real(cg) / d) < TOL


def test_sketching_objective_owns_its_workspace():
    """Function-level calls on the same ansatz (they share one cached workspace and overwrite X / Y / Z) and a second
    objective with another target must not disturb an objective's resident X = I and target."""
    from aqc_research_amd import ParametricCircuit
    import aqc_research_amd.core_op_matrix as com
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx

    n, d = 5, 32
    rng = np.random.default_rng(77)
    circ = ParametricCircuit(n, "cz", create_ansatz_structure(n, "spin", "full", 13))
    t1 = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
    t2 = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
    th = orc.rand_thetas(circ.num_thetas, rng)
    eye = np.eye(d, dtype=np.complex128)
    o1 = SketchingObjectiveEx(circ, FullRangeSketchingVectors(t1))
    f1, g1 = o1.objective_and_gradient(th)
    com.v_mul_mat(circ, th, eye.copy(), None)                     # clobbers the shared workspace's Y
    com.grad_of_matrix_dot_product(circ, th, t2.copy(), t1.copy(), None)   # ... and its X and Z
    o2 = SketchingObjectiveEx(circ, FullRangeSketchingVectors(t2))
    f2, g2 = o2.objective_and_gradient(th)
    f1b, g1b = o1.objective_and_gradient(th + 0.05)
    f2b, g2b = o2.objective_and_gradient(th + 0.05)
    for (f, g), tgt, t in (((f1, g1), t1, th), ((f2, g2), t2, th), ((f1b, g1b), t1, th + 0.05), ((f2b, g2b), t2, th + 0.05)):
        f0, g0 = orc.sketching_objective_and_gradient(circ, t, eye, tgt)
        assert abs(f - f0) < TOL and maxdiff(g, g0) < TOL


def test_sur_max_two_flips_beyond_64_states():
    """ThinStateHandler with max_flips = 2 at n = 11: 1 + 11 + 55 = 67 gathered amplitudes per lane
    (objective_base.py:42-255) -- more than the initial pinned staging holds."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n = 11
    rng = np.random.default_rng(211)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 14))
    user = dict(num_qubits=n, max_flips=2, enable_optim_stats=False, verbose=0)
    obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    target = orc.rand_state(n, rng)
    obj.set_target(target)
    assert obj.num_states == 67
    o = orc.SurMaxOracle(circ, target, 2, None, True)
    th = orc.rand_thetas(circ.num_thetas, rng)
    for _ in range(2):
        assert abs(obj.objective(th) - o.objective(th)) < TOL
        assert maxdiff(obj.gradient(th), o.gradient(th)) < TOL
        th = th + 0.1 * rng.standard_normal(th.size)


def test_eval_without_thetas_and_bad_download_buffer_fail_loudly():
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_Z, HipContext, Workspace

    circ = ParametricCircuit(4, "cx", create_ansatz_structure(4, "spin", "full", 3))
    ws = Workspace(HipContext.of(circ), batch=1)
    with pytest.raises(RuntimeError, match="thetas have not been uploaded"):
        ws.eval(None, vdag=True, gather=False, grad=False)
    with pytest.raises(ValueError):
        ws.download(BUF_Z, out=np.empty((1, 16), dtype=np.complex64))
    with pytest.raises(ValueError):
        ws.download(BUF_Z, out=np.empty((1, 32), dtype=np.complex128)[:, ::2])
    ws.close()


@pytest.mark.parametrize("family", FAMILIES)
def test_forced_family_is_the_one_that_runs(family, monkeypatch):
    """AQC_KERNEL_FAMILY must select the kernels that really execute (no silent fall-back between families)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    _family(monkeypatch, family)
    n = 12
    rng = np.random.default_rng(12)
    circ = ParametricCircuit(n, "cz", create_ansatz_structure(n, "spin", "full", 25))
    ws = Workspace(HipContext.of(circ), batch=3)
    want = int(FAMILY_ENV[family])
    assert [ws.kernel_family(w) for w in (0, 1, 2)] == [want] * 3
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(3)])
    y = np.stack([orc.rand_state(n, rng) for _ in range(3)])
    ws.set_thetas(th)
    ws.upload(BUF_Y, y)
    ws.set_basis(BUF_X, [0, 5, 4095])
    ws.apply(True, BUF_Y, BUF_Z)
    z = ws.download(BUF_Z)
    ws.grad((3, 20), False)
    g = ws.get_grads()
    for b, xi in enumerate((0, 5, 4095)):
        zr = cref.v_dagger_mul_vec(circ, th[b], y[b])
        x = np.zeros(1 << n, complex)
        x[xi] = 1
        assert maxdiff(z[b], zr) < TOL and maxdiff(g[b], cref.grad_of_dot_product(circ, th[b], x, zr, (3, 20), False)) < TOL
    ws.close()
