"""GPU parity added in round 3: the one-sweep form of the surrogate gradient (aqc_ws_set_combo: lhs state
conj(c_0)|state_0> + conj(c_max)|state_max>), and the sizes the earlier suite did not reach -- the headline at the
bench's 256 lanes, the deepest horizons of configs 2 and 4 (12-qubit 2nd-order Trotter ansatz with 12 layers, T = 1620;
20 qubits with 16 layers, T = 3708; user_options.py:65-80,114) and a config-4 style run_jobs whose every record is
replayed on the host with the compiled CPU restatement.  Tolerance 1e-10 absolute (complex fp64), as everywhere."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from oracle import aqc_ref as cref
from tests.helpers import TOL, maxdiff

pytestmark = pytest.mark.gpu


def _trotter(n, layers):
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz

    circ = trotter_ansatz(n, layers, True)
    base = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.6 * layers, delta=1.0)
    return circ, base, neel_state_index(n)


def _batched_reference(circ, idx, th, target, w, mx):
    """One lane of BatchedSurrogateObjective.value_and_grad(update_state=True) replayed with the compiled CPU restatement:
    hysteresis (objective_lhs_sur_max.py:113-117) and weight smoothing (:186) FIRST, then the value and the reference's
    two-sweep gradient (:147-175) under the new state.  Returns (f, g, w, max_no)."""
    n = circ.num_qubits
    z = cref.v_dagger_mul_vec(circ, th, target)
    hs = z[idx]
    hs2 = np.abs(hs) ** 2
    best = hs2[mx]
    for i in range(idx.size):
        if 1.1 * best < hs2[i]:
            best, mx = hs2[i], i
    f_old = 1.0 - (1.0 - w) * hs2[0] - w * hs2[mx]
    w = w + 0.1 * (np.sqrt(abs(f_old)) - w)
    e0 = np.zeros(1 << n, complex); e0[idx[0]] = 1
    g0 = cref.grad_of_dot_product(circ, th, e0, z, None, True)
    if mx == 0:
        g = (g0 * (-2 * np.conj(hs[0]))).real
    else:
        em = np.zeros(1 << n, complex); em[idx[mx]] = 1
        gm = cref.grad_of_dot_product(circ, th, em, z, None, True)
        g = (g0 * (-2 * (1 - w) * np.conj(hs[0]))).real + (gm * (-2 * w * np.conj(hs[mx]))).real
    return 1.0 - (1.0 - w) * hs2[0] - w * hs2[mx], g, w, mx


def test_set_combo_is_the_linear_combination_of_two_sweeps():
    """g(conj(c0) x0 + conj(c1) x1) == c0 g(x0) + c1 g(x1): conjugate-linearity of core_operations.py:823-1019 in x, lane-wise
    coefficients, one- and two-term lanes mixed, positions re-used across calls (only the previous ones are cleared)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_X2, BUF_Y, BUF_Z, HipContext, Workspace

    n, B = 13, 5
    rng = np.random.default_rng(313)
    circ = ParametricCircuit(n, "cp", create_ansatz_structure(n, "spin", "full", 19))
    ws = Workspace(HipContext.of(circ), batch=B)
    th = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(B)])
    y = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    ws.upload(BUF_Y, y)
    ws.set_thetas(th)
    ws.apply(True, BUF_Y, BUF_Z)
    for trial in range(3):
        idx = np.stack([rng.choice(1 << n, size=2, replace=False) for _ in range(B)]).astype(np.int64)
        idx[trial % B, 1] = -1                       # a one-term lane
        cf = rng.standard_normal((B, 2)) + 1j * rng.standard_normal((B, 2))
        ws.set_combo(BUF_X2, idx, cf)
        x = ws.download(BUF_X2)
        want = np.zeros((B, 1 << n), complex)
        for b in range(B):
            want[b, idx[b, 0]] = cf[b, 0]
            if idx[b, 1] >= 0:
                want[b, idx[b, 1]] = cf[b, 1]
        assert maxdiff(x, want) == 0.0               # exactly two non-zeros per lane, the old ones are gone
        _, g = ws.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=(2, 17), front_layer=True)
        for b in range(B):
            z = cref.v_dagger_mul_vec(circ, th[b], y[b])
            e0 = np.zeros(1 << n, complex); e0[idx[b, 0]] = 1
            ref = np.conj(cf[b, 0]) * cref.grad_of_dot_product(circ, th[b], e0, z, (2, 17), True)
            if idx[b, 1] >= 0:
                e1 = np.zeros(1 << n, complex); e1[idx[b, 1]] = 1
                ref = ref + np.conj(cf[b, 1]) * cref.grad_of_dot_product(circ, th[b], e1, z, (2, 17), True)
            assert maxdiff(g[b], ref) < 4 * TOL      # coefficients of modulus up to ~3
    with pytest.raises(RuntimeError):
        ws.set_combo(BUF_X2, np.tile([3, 3], (B, 1)), np.ones((B, 2)))   # the two states of a lane must differ
    ws.close()


@pytest.mark.parametrize("n,layers", [(12, 2), (16, 2)])
def test_surrogate_object_with_a_leading_flip_state(n, layers):
    """objective_lhs_sur_max.py:127-191 when the leading state is NOT |state_0> (the usual case on random targets): the
    objective object runs one combined sweep; values, gradients, hysteresis and weights must follow orc.SurMaxOracle, which
    runs the reference's two sweeps."""
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    circ, base, neel = _trotter(n, layers)
    rng = np.random.default_rng(7 * n)
    target = orc.rand_state(n, rng)
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0)
    obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    obj.set_target(target)
    o = orc.SurMaxOracle(circ, target, 1, None, True, base_index=neel)
    th = base + 0.1 * np.pi * (2 * rng.random(base.size) - 1)
    led = 0
    for step in range(4):
        f, fo = obj.objective(th), o.objective(th)
        assert abs(f - fo) < TOL and obj._max_no == o.max_no
        g, go = obj.gradient(th), o.gradient(th)
        assert maxdiff(g, go) < TOL and abs(obj._weight - o.weight) < 1e-13
        led += obj._max_no != 0
        th = th - 0.05 * g
    assert led >= 3   # random target: a flip state leads from the first evaluation on


def test_batched_surrogate_and_device_lbfgs_with_leading_flip_states():
    """The lane-batched surrogate (host assembly) and the device-resident L-BFGS on random targets, where flip states lead:
    value_and_grad against orc.SurMaxOracle lane by lane over a few descent steps, then minimize_on_device must decrease
    every lane's objective and leave the object's state (weight, leading state) equal to the device's."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective

    n, B = 12, 6
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(99)
    targets = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    th = base + 0.1 * np.pi * (2 * rng.random((B, base.size)) - 1)
    bo = BatchedSurrogateObjective(circ, targets, base_index=neel)
    idx = orc.flip_state_indices(n, 1, neel)
    w, mx = np.ones(B), np.zeros(B, dtype=int)          # the reference's state machine, replayed here lane by lane
    for step in range(3):
        f, g = bo.value_and_grad(th)                     # update_state=True: hysteresis + smoothing first, then f, g under the NEW state
        for b in range(B):
            fr, gr, w[b], mx[b] = _batched_reference(circ, idx, th[b], targets[b], w[b], mx[b])
            assert abs(f[b] - fr) < TOL and maxdiff(g[b], gr) < TOL and bo.max_no[b] == mx[b] and abs(bo.weight[b] - w[b]) < 1e-13
        th = th - 0.05 * g
    assert (bo.max_no != 0).any()
    bo.close()
    bo = BatchedSurrogateObjective(circ, targets, base_index=neel)
    f0, _ = bo.value_and_grad(th, update_state=False)
    res = bo.minimize_on_device(th, maxiter=8)
    assert (res["fun"] < f0 + 1e-12).all() and (res["fidelity"] >= 0).all()
    # the host object continues from the device's state: same value at the returned point under that state
    w, m = bo.weight.copy(), bo.max_no.copy()
    f1, _ = bo.value_and_grad(res["x"], update_state=False)
    assert maxdiff(f1, res["fun"]) < 1e-9 and (bo.max_no == m).all() and maxdiff(bo.weight, w) == 0.0
    bo.close()


def test_surrogate_eval_call_between_other_users_of_the_lhs_buffer():
    """aqc_ws_surrogate_eval (one native call per evaluation, replayed as a graph) keeps its two-hot lhs states in X2 next to
    set_combo and the device L-BFGS, which write the same buffer: after either of them the next evaluation must still match the
    host replay of the reference's state machine; trials (update_state=False) leave the state alone; bad states are refused."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
    from aqc_research_amd.engine import BUF_X2

    n, B = 13, 5
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(1313)
    targets = np.stack([orc.rand_state(n, rng) for _ in range(B)])
    idx = orc.flip_state_indices(n, 1, neel)
    bo = BatchedSurrogateObjective(circ, targets, base_index=neel)
    w, mx = np.ones(B), np.zeros(B, dtype=int)

    def check(th, update):
        f, g = bo.value_and_grad(th, update_state=update)
        for b in range(B):
            if update:
                fr, gr, w[b], mx[b] = _batched_reference(circ, idx, th[b], targets[b], w[b], mx[b])
            else:   # the frozen state: value and two-sweep gradient of the reference under (w, mx) as they are
                z = cref.v_dagger_mul_vec(circ, th[b], targets[b])
                hs = z[idx]
                fr = 1.0 - (1.0 - w[b]) * abs(hs[0]) ** 2 - w[b] * abs(hs[mx[b]]) ** 2
                e0 = np.zeros(1 << n, complex); e0[idx[0]] = 1
                g0 = cref.grad_of_dot_product(circ, th[b], e0, z, None, True)
                if mx[b] == 0:
                    gr = (g0 * (-2 * np.conj(hs[0]))).real
                else:
                    em = np.zeros(1 << n, complex); em[idx[mx[b]]] = 1
                    gm = cref.grad_of_dot_product(circ, th[b], em, z, None, True)
                    gr = (g0 * (-2 * (1 - w[b]) * np.conj(hs[0]))).real + (gm * (-2 * w[b] * np.conj(hs[mx[b]]))).real
            assert abs(f[b] - fr) < TOL and maxdiff(g[b], gr) < TOL
            assert bo.max_no[b] == mx[b] and abs(bo.weight[b] - w[b]) < 1e-13
        return g

    th = base + 0.1 * np.pi * (2 * rng.random((B, base.size)) - 1)
    g = check(th, True)
    g = check(th - 0.05 * g, False)                       # a trial: state untouched
    bo.ws.set_combo(BUF_X2, np.tile([3, 70], (B, 1)), np.ones((B, 2), dtype=complex))   # somebody else's pattern in X2
    g = check(th - 0.02 * g, True)
    res = bo.minimize_on_device(th, maxiter=2)            # the device L-BFGS rewrites X2 and the objective state
    w[:], mx[:] = bo.weight, bo.max_no
    check(res["x"], True)
    check(res["x"] + 0.01, False)
    bad = np.full(B, idx.size, dtype=np.int64)
    with pytest.raises(RuntimeError):
        bo.ws.surrogate_eval(th, np.ones(B), bad, True)
    bo.close()


def test_surrogate_eval_state_machine_on_crafted_amplitudes():
    """Hysteresis in both directions and the three update modes of aqc_ws_surrogate_eval.  With target = V(theta)|psi>,
    |psi> = sum_i c_i |state_i>, the amplitudes at theta are exactly the c_i, so the leading state can be steered: |state_0>
    -> a flip state (>10 % better), stays there (a better one within 10 %), -> another one, back to |state_0>.  Mode 1
    (hysteresis + weight smoothing) and mode 0 against the host replay; mode 2 = mode 1's leading state with the weight left
    alone and value / gradient under (old weight, new leading state) (objective_lhs_sur_max.py:110-186)."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
    from aqc_research_amd.engine import BUF_Y

    n, B = 12, 3
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(2024)
    idx = orc.flip_state_indices(n, 1, neel)
    S = idx.size
    th = base + 0.05 * np.pi * (2 * rng.random((B, base.size)) - 1)
    # moduli of (c_0, c_1, c_2) per step; the other states get small noise; phases random
    script = [(0.50, 0.60, 0.10), (0.60, 0.62, 0.64), (0.30, 0.55, 0.70), (0.70, 0.50, 0.60), (0.70, 0.72, 0.10)]
    expect = [1, 1, 2, 0, 0]          # leading state after each step: 0.36 > 1.1 * 0.25 | 0.41 < 1.1 * 0.38 | 0.49 > 1.1 * 0.30 | 0.49 > 1.1 * 0.36 | 0.52 < 1.1 * 0.49
    bo = BatchedSurrogateObjective(circ, np.tile(orc.rand_state(n, rng), (B, 1)), base_index=neel)
    w, mx = np.ones(B), np.zeros(B, dtype=int)
    for step, (m0, m1, m2) in enumerate(script):
        targets = np.empty((B, 1 << n), complex)
        for b in range(B):
            c = 0.01 * (rng.standard_normal(S) + 1j * rng.standard_normal(S))
            c[:3] = np.array([m0, m1, m2]) * np.exp(2j * np.pi * rng.random(3))
            psi = np.zeros(1 << n, complex)
            psi[idx] = c
            targets[b] = cref.v_mul_vec(circ, th[b], psi)      # not normalised: the objective is a polynomial in the amplitudes
        bo.ws.upload(BUF_Y, targets)
        # mode 2 first (it leaves the weight alone): leading state of mode 1, value and gradient under (old weight, that state)
        w2, m2_ = bo.weight.copy(), bo.max_no.copy()
        f2, _, hs2, g2 = bo.ws.surrogate_eval(th, w2, m2_, 2, None, True)
        assert (m2_ == expect[step]).all() and maxdiff(w2, bo.weight) == 0.0
        for b in range(B):
            h = hs2[b]
            assert maxdiff(np.abs(h[:3]), np.array(script[step])) < 1e-12
            k = expect[step]
            f_ref = 1.0 - (1.0 - w[b]) * abs(h[0]) ** 2 - w[b] * abs(h[k]) ** 2
            z = cref.v_dagger_mul_vec(circ, th[b], targets[b])
            e0 = np.zeros(1 << n, complex); e0[idx[0]] = 1
            g0 = cref.grad_of_dot_product(circ, th[b], e0, z, None, True)
            if k == 0:
                g_ref = (g0 * (-2 * np.conj(h[0]))).real
            else:
                ek = np.zeros(1 << n, complex); ek[idx[k]] = 1
                gk = cref.grad_of_dot_product(circ, th[b], ek, z, None, True)
                g_ref = (g0 * (-2 * (1 - w[b]) * np.conj(h[0]))).real + (gk * (-2 * w[b] * np.conj(h[k]))).real
            assert abs(f2[b] - f_ref) < TOL and maxdiff(g2[b].real, g_ref) < TOL
        # mode 1 through the objective: the host replay of the reference's state machine
        f, g = bo.value_and_grad(th, update_state=True)
        for b in range(B):
            fr, gr, w[b], mx[b] = _batched_reference(circ, idx, th[b], targets[b], w[b], mx[b])
            assert abs(f[b] - fr) < TOL and maxdiff(g[b], gr) < TOL and bo.max_no[b] == mx[b] == expect[step]
            assert abs(bo.weight[b] - w[b]) < 1e-13
    bo.close()


def test_surrogate_object_steered_through_its_state_machine():
    """SpSurrogateObjectiveMax, objective() + gradient() pairs, on the crafted amplitudes of the test above: |state_0> leads,
    a flip state takes over, holds against a slightly better one, loses to a much better one, |state_0> returns -- the pairs
    with a leading flip state are ONE device call (aqc_ws_surrogate_eval, mode 2); every value, gradient, leading state and
    weight against orc.SurMaxOracle (the reference's two sweeps)."""
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n = 12
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(4048)
    idx = orc.flip_state_indices(n, 1, neel)
    th = base + 0.05 * np.pi * (2 * rng.random(base.size) - 1)
    script = [(0.50, 0.60, 0.10), (0.60, 0.62, 0.64), (0.30, 0.55, 0.70), (0.70, 0.50, 0.60), (0.70, 0.72, 0.10), (0.40, 0.30, 0.60)]
    expect = [1, 1, 2, 0, 0, 2]
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0)
    obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    o = None
    for step, mods in enumerate(script):
        c = 0.01 * (rng.standard_normal(idx.size) + 1j * rng.standard_normal(idx.size))
        c[:3] = np.array(mods) * np.exp(2j * np.pi * rng.random(3))
        psi = np.zeros(1 << n, complex)
        psi[idx] = c
        target = cref.v_mul_vec(circ, th, psi)
        obj.set_target(target)
        if o is None:
            o = orc.SurMaxOracle(circ, target, 1, None, True, base_index=neel)
        o.target = target
        f, fo = obj.objective(th), o.objective(th)
        assert abs(f - fo) < TOL and obj._max_no == o.max_no == expect[step]
        g, go = obj.gradient(th), o.gradient(th)
        assert maxdiff(g, go) < TOL and abs(obj._weight - o.weight) < 1e-13
        th = th + 1e-3 * rng.standard_normal(th.size)    # a new point every pair (the amplitudes are re-crafted for it)


@pytest.mark.parametrize("B", [256, 1024])
def test_headline_all_lanes(B):
    """The bench's unit of work: 16 qubits, 40 blocks, 1024 lanes by default (64 items per persistent sweep workgroup; 256
    lanes: 16), every lane against the compiled CPU restatement."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    n, L = 16, 40
    rng = np.random.default_rng(16000 + B)
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, L))
    thetas = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(B)])
    y = orc.rand_state(n, rng)
    ws = Workspace(HipContext.of(ParametricCircuit(n, "cx", a.blocks)), batch=B)
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, 0)
    ws.gather_setup([0])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(a, thetas, y, 0, threads=16)
    assert maxdiff(hs[:, 0], hs_ref) < TOL and maxdiff(grads, g_ref) < TOL
    ws.close()


@pytest.mark.parametrize("n,L,B", [(13, 1, 300), (13, 2, 300), (14, 3, 150), (13, 6, 517)])
def test_shallow_circuits_on_the_persistent_sweep(n, L, B):
    """Stages with one or two sub-stages on 2^12 tiles, several items per persistent workgroup: the next item's operands are
    then requested after (single sub-stage) or inside the last sub-stage, not behind sub-stages 1-4 as at the headline depth."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    rng = np.random.default_rng(1000 * n + 10 * L + B)
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, L))
    thetas = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(B)])
    y = orc.rand_state(n, rng)
    ws = Workspace(HipContext.of(ParametricCircuit(n, "cx", a.blocks)), batch=B)
    assert ws.plan_info(1)[1] == 12
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, 5)
    ws.gather_setup([5])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(a, thetas, y, 5, threads=16)
    assert maxdiff(hs[:, 0], hs_ref) < TOL and maxdiff(grads, g_ref) < TOL
    ws.close()


def test_config1_size_at_the_bench_lane_count():
    """bench.py --workload mat5_cyc180 as it runs by default: the docs/aqc.ipynb ansatz (5 qubits, cyclic_spin, 180 blocks,
    T = 735) on 32-column targets, 1024 lanes (one 2^10 tile per lane): <I|V^H U> and the gradient of every 41st lane against
    the compiled CPU restatement (core_op_matrix.py:645-762)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace

    n, d, B = 5, 32, 1024
    rng = np.random.default_rng(51024)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "cyclic_spin", "full", 180))
    th = np.pi * (2 * rng.random((B, circ.num_thetas)) - 1)
    targets = np.stack([np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0] for _ in range(B)])
    ws = Workspace(HipContext.of(circ), batch=B, ncols=d)
    ws.upload(BUF_Y, targets)
    ws.set_identity(BUF_X)
    ws.set_thetas(th)
    ws.apply(True, BUF_Y, BUF_Z)
    tr = ws.vdot(BUF_X, BUF_Z)
    ws.grad(None, True)
    g = ws.get_grads()
    eye = np.eye(d, dtype=complex)
    for b in list(range(0, B, 41)) + [B - 1]:
        vhy = cref.v_dagger_mul_mat(circ, th[b], targets[b])
        assert abs(tr[b] - np.vdot(eye, vhy)) < TOL * d
        assert maxdiff(g[b], cref.grad_of_matrix_dot_product(circ, th[b], eye, vhy)) < TOL * d
    ws.close()


def test_config2_size_at_the_bench_lane_count():
    """bench.py --workload sv12_trotter2 as it runs by default: 12 qubits, 2nd-order Trotter ansatz of 2 layers, 1024 lanes
    (one 2^12 tile per lane, four items per persistent workgroup), each lane with its own target; every seventh lane against
    the compiled CPU restatement."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    n, B = 12, 1024
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(121024)
    thetas = base[None, :] + 0.2 * np.pi * (2 * rng.random((B, circ.num_thetas)) - 1)
    tg = rng.standard_normal((B, 1 << n)) + 1j * rng.standard_normal((B, 1 << n))
    tg /= np.linalg.norm(tg, axis=1, keepdims=True)
    ws = Workspace(HipContext.of(circ), batch=B)
    ws.upload(BUF_Y, tg)
    ws.set_basis(BUF_X, neel)
    ws.gather_setup([neel])
    hs, grads = ws.eval(thetas, gather=True)
    for b in list(range(0, B, 7)) + [B - 1]:   # (the restatement takes one target per call)
        h_ref, g_ref = cref.eval_batch(circ, thetas[b][None, :], tg[b], neel, threads=1)
        assert abs(hs[b, 0] - h_ref[0]) < TOL and maxdiff(grads[b], g_ref[0]) < TOL
    ws.close()


@pytest.mark.parametrize("n,layers,B,T", [(12, 12, 6, 1620), (20, 16, 2, 3708)])
def test_deepest_horizons_direct_parity(n, layers, B, T):
    """Config 2's last horizon (12 qubits, 12 layers) and config 4's (20 qubits, 16 layers: L = 912 blocks) through the
    batched workspace: V^H|target>, the flip-state amplitudes and the full gradient against the compiled CPU restatement."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    circ, base, neel = _trotter(n, layers)
    assert circ.num_thetas == T
    rng = np.random.default_rng(n * 100 + layers)
    thetas = base[None, :] + 0.1 * np.pi * (2 * rng.random((B, T)) - 1)
    y = orc.rand_state(n, rng)
    ws = Workspace(HipContext.of(circ), batch=B)
    ws.broadcast(BUF_Y, y)
    ws.set_basis(BUF_X, neel)
    ws.gather_setup([neel])
    hs, grads = ws.eval(thetas, gather=True)
    hs_ref, g_ref = cref.eval_batch(circ, thetas, y, neel, threads=8)
    assert maxdiff(hs[:, 0], hs_ref) < TOL and maxdiff(grads, g_ref) < TOL
    ws.close()


def test_cfg4_style_run_jobs_replayed_on_the_host():
    """run_jobs over (horizon, seeds) configurations as the config-4 bench drives it, at 14 qubits: every record's thetas
    must equal a host replay of the same fixed-step descent with the compiled CPU restatement (_batched_reference)."""
    import bench
    from aqc_research_amd.job_executor import run_jobs

    n, evals, seeds = 14, 3, [11, 12, 13]
    configs = [{"n": n, "horizon": h, "evals": evals, "device": 0, "seeds": [s + 100 * h for s in seeds]} for h in (1, 2)]
    results = run_jobs(configs, 1, bench._mix_job, records="fixed")
    assert len(results) == 2 and all(r["status"] == "ok" for r in results)
    for cfg, r in zip(configs, results):
        circ, base, neel = bench._mix_setup(n, cfg["horizon"])
        sd = cfg["seeds"][0]                          # the record carries lane 0's parameters
        target, th = bench._mix_target(n, sd), bench._mix_start(base, sd)
        idx = orc.flip_state_indices(n, 1, neel)
        w, mx = 1.0, 0
        for _ in range(evals):
            _, g, w, mx = _batched_reference(circ, idx, th, target, w, mx)
            th = th - 0.05 * g
        assert maxdiff(r["thetas"], th) < 1e-9      # three chained descent steps


def test_mps_batched_contraction_and_slot_cache():
    """aqc_ws_mps_to_vec_batch: all lanes' MPS -> dense in one launch chain (mps_operations.py:159-189 per lane), lanes that
    share a tuple share its resident copy, mixed bond dimensions run one chain per distinct dimension vector, the arrays of a
    cached tuple are read-only (a tuple whose owner modifies it all the same is uploaded again: fingerprint), and more distinct tuples than slots recycle the least recently used one."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_Y, HipContext, Workspace

    n, B = 10, 7
    rng = np.random.default_rng(1010)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 9))
    ws = Workspace(HipContext.of(circ), batch=B)
    a, b, c = orc.random_mps(n, 8, rng), orc.random_mps(n, 8, rng), orc.random_mps(n, 3, rng)
    lanes = [a, b, a, b, b, a, a]
    ws.mps_to_vec_batch(lanes, BUF_Y)
    got = ws.download(BUF_Y)
    for i, m in enumerate(lanes):
        assert maxdiff(got[i], orc.mps_to_vector(m)) < 1e-12
    assert len(ws._mps_cache) == 2
    mixed = [a, c, b, c, a, c, b]                      # two different sets of bond dimensions
    ws.mps_to_vec_batch(mixed, BUF_Y)
    got = ws.download(BUF_Y)
    for i, m in enumerate(mixed):
        assert maxdiff(got[i], orc.mps_to_vector(m)) < 1e-12
    assert all(g.flags.writeable for pair in a[0] for g in pair)   # the caller's arrays are left as they came
    a[0][3][0][0, 1] += 0.25                            # same tuple, new contents (any entry: the digest covers small tuples completely)
    ws.mps_to_vec_batch([a] * B, BUF_Y, lanes=np.arange(B)[::-1])
    assert maxdiff(ws.download(BUF_Y)[0], orc.mps_to_vector(a)) < 1e-12
    many = [orc.random_mps(n, 2, rng) for _ in range(70)]   # more than the 60 cached slots
    for k in range(0, 70, B):
        chunk = (many[k:k + B] + many[:B])[:B]
        ws.mps_to_vec_batch(chunk, BUF_Y)
        got = ws.download(BUF_Y)
        for i, m in enumerate(chunk):
            assert maxdiff(got[i], orc.mps_to_vector(m)) < 1e-12
    assert len(ws._mps_cache) <= 60
    ws.close()


@pytest.mark.parametrize("n,chi", [(2, 2), (3, 2), (5, 4), (9, 7), (13, 16)])
def test_mps_batched_contraction_shapes(n, chi):
    """The batched MPS -> dense chain on small and odd registers (left half of one site, right half of one or two), and its
    product-state shortcut (every bond dimension 1: |0>, a Neel state, random product states) -- each lane against
    orc.mps_to_vector, i.e. mps_operations.py:159-189."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace

    B = 4
    rng = np.random.default_rng(100 * n + chi)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", max(n - 1, 1)))
    ws = Workspace(HipContext.of(circ), batch=B)
    lanes = [orc.random_mps(n, chi, rng) for _ in range(B)]
    ws.mps_to_vec_batch(lanes, BUF_Y)
    got = ws.download(BUF_Y)
    for i, m in enumerate(lanes):
        assert maxdiff(got[i], orc.mps_to_vector(m)) < 1e-12

    def product(vecs):   # QiskitMPS of a product state: gamma_q = ([[v0]], [[v1]]), lambdas 1
        return ([(np.array([[v[0]]], complex), np.array([[v[1]]], complex)) for v in vecs], [np.ones(1) for _ in range(n - 1)])

    zero = product([(1.0, 0.0)] * n)
    neel = product([(1.0, 0.0) if q % 2 == 0 else (0.0, 1.0) for q in range(n)])
    rnd = [product(rng.standard_normal((n, 2)) + 1j * rng.standard_normal((n, 2))) for _ in range(2)]
    prods = [zero, neel] + rnd
    ws.mps_to_vec_batch(prods, BUF_X)
    got = ws.download(BUF_X)
    for i, m in enumerate(prods):
        assert maxdiff(got[i], orc.mps_to_vector(m)) < 1e-12
    assert got[0][0] == 1.0 and np.count_nonzero(got[0]) == 1 and np.count_nonzero(got[1]) == 1
    ws.close()


def test_batched_surrogate_objective_with_mps_targets():
    """Config 3 as a batched objective: the lanes' targets arrive as QiskitMPS tuples (set_mps_targets); values and gradients
    equal those of the objective built on the densified targets."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective

    n, B = 12, 5
    circ, base, neel = _trotter(n, 2)
    rng = np.random.default_rng(1205)
    mps = [orc.random_mps(n, 6, rng) for _ in range(B)]
    dense = np.stack([orc.mps_to_vector(m) for m in mps])
    th = base + 0.1 * np.pi * (2 * rng.random((B, base.size)) - 1)
    a = BatchedSurrogateObjective(circ, dense, base_index=neel)
    b = BatchedSurrogateObjective(circ, None, lanes=B, base_index=neel)
    b.set_mps_targets(mps)
    for _ in range(2):
        fa, ga = a.value_and_grad(th)
        fb, gb = b.value_and_grad(th)
        assert maxdiff(fa, fb) < TOL and maxdiff(ga, gb) < TOL and (a.max_no == b.max_no).all()
        th = th - 0.05 * ga
    with pytest.raises(ValueError):
        b.set_mps_targets(mps[:2])
    a.close(); b.close()


def test_mps_front_door_hands_the_state_over_on_the_device():
    """v_dagger_mul_mps -> fast_dot_gradient (mps_operations.py:349-371, mps_dot_objective.py:41-242): the returned MPS is a
    QiskitMPS tuple in canonical form (checked by contracting it back), and the gradient call that follows takes its dense
    state from the device; the same objects keep giving the right answer after the device copy has been overwritten."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_Z, HipContext
    from aqc_research_amd.mps_dot_objective import fast_dot_gradient
    from aqc_research_amd.mps_operations import DenseBackedMPS, check_mps, mps_dot, mps_to_vector, v_dagger_mul_mps

    n = 12
    rng = np.random.default_rng(1212)
    circ = ParametricCircuit(n, "cz", create_ansatz_structure(n, "spin", "full", 15))
    target = orc.random_mps(n, 8, rng)
    lvec = orc.random_mps(n, 2, rng)
    th = orc.rand_thetas(circ.num_thetas, rng)
    y, x = orc.mps_to_vector(target), orc.mps_to_vector(lvec)
    z_ref = cref.v_dagger_mul_vec(circ, th, y)
    g_ref = cref.grad_of_dot_product(circ, th, x, z_ref, None, True)
    scale = np.linalg.norm(x) * np.linalg.norm(y)
    vh = v_dagger_mul_mps(circ, th, target)
    ws = HipContext.of(circ).workspace(1, 1)
    assert isinstance(vh, DenseBackedMPS) and isinstance(vh, tuple) and vh.dense_on(ws, BUF_Z) and vh._mps is None
    g = fast_dot_gradient(circ, th, lvec, vh)
    assert vh._mps is None                              # nobody looked at the tensors: no SVD chain was run
    assert maxdiff(g, g_ref) < TOL * max(1.0, scale)
    assert check_mps(vh) and len(vh) == 2 and vh._mps is None
    gam, lam = vh
    assert check_mps((gam, lam))                        # the materialised tensors pass the structural check as a plain tuple
    assert len(gam) == n and all(np.all(np.diff(l) <= 1e-15) for l in lam) and max(l.size for l in lam) <= 1 << (n // 2)
    # (contracting it back runs on another workspace; at the no-truncation threshold only numerically zero Schmidt values go)
    assert maxdiff(mps_to_vector(vh), z_ref) < TOL and maxdiff(vh.dense_state, z_ref) < TOL
    other = v_dagger_mul_mps(circ, th + 0.1, target)    # overwrites Z of the circuit's workspace
    assert not vh.dense_on(ws, BUF_Z) and other.dense_on(ws, BUF_Z)
    assert maxdiff(fast_dot_gradient(circ, th, lvec, vh), g_ref) < TOL * max(1.0, scale)   # falls back to the host copy
    assert abs(mps_dot(lvec, vh) - np.vdot(x, z_ref)) < TOL * max(1.0, scale)
    # a truncated result is a smaller MPS whose infidelity is the discarded weight's order
    tr = v_dagger_mul_mps(circ, th, target, trunc_thr=1e-12)
    full = max(l.size for l in vh[1])
    small = DenseBackedMPS(tr.dense_state, 1e-3)
    w = orc.mps_to_vector((small[0], small[1]))
    infid = 1 - abs(np.vdot(w, z_ref)) ** 2 / (np.vdot(w, w).real * np.vdot(z_ref, z_ref).real)
    assert max(l.size for l in small[1]) < full and 0 <= infid < 2e-2


def test_non_canonical_mps_input_is_canonicalised_under_truncation():
    """The engine's discard rule is stated on Schmidt values (mps_operations.py:216-243 hands Aer's canonical states on): a
    hand-made, NON-canonical Vidal-form input (orc.random_mps) with a real truncation threshold is brought into canonical
    form on import, after which the truncated V^H obeys infidelity ~ discarded weight, as canonical inputs do; without the
    step the same call was off by 1e-2 (DESIGN 6b.5 of round 2)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.mps_engine import DeviceMPS, is_canonical, v_dagger_mul_mps

    n = 10
    rng = np.random.default_rng(77)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 27))
    th = orc.rand_thetas(circ.num_thetas, rng)
    raw = orc.random_mps(n, 8, rng)
    assert not is_canonical(raw)
    y = orc.mps_to_vector(raw)
    ref = cref.v_dagger_mul_vec(circ, th, y)
    m = DeviceMPS.from_qiskit(raw, trunc_thr=1e-6)            # canonicalised on import
    back = m.to_qiskit()
    assert is_canonical(back) and maxdiff(orc.mps_to_vector(back), y) < TOL    # same state, canonical gauge
    out = v_dagger_mul_mps(circ, th, m, trunc_thr=1e-6)
    got = orc.mps_to_vector(out.to_qiskit())
    infid = 1 - abs(np.vdot(got, ref)) ** 2 / (np.vdot(got, got).real * np.vdot(ref, ref).real)
    assert 0 <= infid < 20 * max(out.discarded_weight, 1e-9), (infid, out.discarded_weight)
    exact = DeviceMPS.from_qiskit(raw)                         # exact arithmetic: tensors taken as they are
    assert not is_canonical(exact.to_qiskit())
    assert maxdiff(orc.mps_to_vector(v_dagger_mul_mps(circ, th, exact).to_qiskit()), ref) < 10 * TOL
