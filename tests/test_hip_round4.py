"""GPU: round-4 additions -- the MPS objective at the reference's OWN default truncation threshold (1e-6), on both routes."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import TOL, maxdiff

pytestmark = pytest.mark.gpu


def _trotter_problem(n, layers, seed):
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_state
    from aqc_research_amd.mps_operations import vector_to_canonical_mps

    rng = np.random.default_rng(seed)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
    neel = neel_state_index(n)
    evol_time = 0.6 * layers
    th = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=evol_time, delta=1.0)
    th = th + 0.05 * rng.standard_normal(th.size)
    dense = trotter_state(n, evol_time=evol_time, num_steps=6 * layers, delta=1.0, second_order=True)
    target = vector_to_canonical_mps(dense, 1e-6)     # what Aer hands the reference at its default threshold
    return circ, th, neel, target


def test_mps_objective_at_the_reference_default_threshold_takes_the_dense_route():
    """user_options.py:55 -> time_evol_best_init.py:80 -> objective_lhs_sur_fast_mps_trotter.py:99: the reference's driver
    hands trunc_thr = 1e-6 to the MPS objective.  At n = 16 that call must run on the fused dense kernels (exact: the
    north-star tolerance against the dense oracle on the SAME target tensors), not on the truncated-SVD engine."""
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_fast_mps_trotter import SpSurrogateObjectiveFastMpsTrotter
    from aqc_research_amd import mps_dot_objective as mdo

    n = 16
    circ, th, neel, target = _trotter_problem(n, 2, 160)
    assert mdo.use_dense(n, 1e-6) and mdo.use_dense(24, 1e-3) and not mdo.use_dense(25, 1e-16)
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0, maxiter=5,
                trunc_thr=1e-6)
    o = SpSurrogateObjectiveFastMpsTrotter(user_parameters=user, circ=circ)
    o.set_target(target)
    assert not o._native_mps
    ref = orc.SurMaxOracle(circ, orc.mps_to_vector(target), 1, None, True, base_index=neel)
    for step in range(3):
        t = th + 0.03 * step
        f, g = o.objective(t), o.gradient(t)
        fr, gr = ref.objective(t), ref.gradient(t)
        assert abs(f - fr) < TOL and maxdiff(g, gr) < TOL


def test_mps_objective_truncated_engine_stays_within_the_discarded_weight_bound(monkeypatch):
    """The opt-in engine route (AQC_MPS_METHOD=mps) at trunc_thr = 1e-6, n = 16, against the dense oracle.  Bound: a state
    that went through K truncations discarding weights d_k differs from the exact one by eps <= sum_k sqrt(2 d_k) <=
    sqrt(2 K D), D = sum d_k (aqc_mps_discarded_weight).  |h| changes by <= eps_vh, f = 1 - |h0|^2 by <= 2 eps + eps^2; a
    gradient entry 0.5j <P w|z> by <= (eps_w + eps_z + eps_w eps_z) / 2, times |c| <= 2 of the surrogate's combination."""
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_fast_mps_trotter import SpSurrogateObjectiveFastMpsTrotter
    from aqc_research_amd.mps_engine import DeviceMPS, v_dagger_mul_mps, v_mul_mps

    n, thr = 16, 1e-6
    circ, th, neel, target = _trotter_problem(n, 2, 161)
    monkeypatch.setenv("AQC_MPS_METHOD", "mps")
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0, maxiter=5,
                trunc_thr=thr)
    o = SpSurrogateObjectiveFastMpsTrotter(user_parameters=user, circ=circ)
    o.set_target(target)
    assert o._native_mps
    f, g = o.objective(th), o.gradient(th)
    ref = orc.SurMaxOracle(circ, orc.mps_to_vector(target), 1, None, True, base_index=neel)
    fr, gr = ref.objective(th), ref.gradient(th)
    # discarded weights of the three walks the evaluation makes: V^H|target>, then V on both sweep operands
    K = int(circ.num_blocks + circ.half_layer_num_blocks)   # 2-qubit gates = truncation events of one walk
    tgt = DeviceMPS.from_qiskit(target, trunc_thr=thr)
    vh = v_dagger_mul_mps(circ, th, tgt, trunc_thr=thr)
    z = v_mul_mps(circ, th, vh, trunc_thr=thr)
    w = v_mul_mps(circ, th, DeviceMPS.basis_state(n, neel), trunc_thr=thr)
    d_vh, d_z, d_w = vh.discarded_weight, z.discarded_weight, w.discarded_weight
    for m in (tgt, vh, z, w):
        m.close()
    eps_vh, eps_z, eps_w = (float(np.sqrt(2 * K * d)) for d in (d_vh, d_z, d_w))
    bound_f = 2 * eps_vh + eps_vh ** 2
    bound_g = eps_w + eps_z + eps_w * eps_z
    assert d_vh <= K * thr * 1.01 and d_z <= 2 * K * thr * 1.01
    assert abs(f - fr) <= bound_f + TOL and maxdiff(g, gr) <= bound_g + TOL
    # and what is observed is first order in the discarded weight itself (infidelity ~ D), far inside the bound
    assert abs(f - fr) <= 50 * max(d_vh, 1e-12) + 1e-9, (abs(f - fr), d_vh)


def test_mps_to_vec_batch_with_mixed_bond_dimensions():
    """Lanes whose MPS operands differ in their bond dimensions (truncated canonical tensors do): the workspace pads the resident copies
    with zeros to a common shape (one contraction chain; the native call alone would run one chain per distinct shape) -- every lane
    must still receive ITS state (mps_operations.py:159-189 per lane); the same operands unpadded through the native grouping."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.engine import BUF_Y, HipContext, Workspace

    n = 9
    rng = np.random.default_rng(94)
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 4))
    a, b, c = orc.random_mps(n, 4, rng), orc.random_mps(n, 7, rng), orc.random_mps(n, 7, rng)
    zero = ([(np.ones((1, 1), complex), np.zeros((1, 1), complex)) for _ in range(n)], [np.ones(1) for _ in range(n - 1)])
    lanes = [b, a, zero, c, a, b, c]
    ws = Workspace(HipContext.of(circ), batch=len(lanes))
    for _ in range(2):                 # second round: resident copies and pointer tables are found again
        ws.mps_to_vec_batch(lanes, BUF_Y)
        out = ws.download(BUF_Y)
        for i, m in enumerate(lanes):
            assert maxdiff(out[i], orc.mps_to_vector(m)) < TOL
    # the native call on operands of different shapes (no padding: slots filled one by one)
    import ctypes

    distinct = [a, b, c, zero]
    for k, m in enumerate(distinct):
        ws.mps_upload(k, m)
    slots = np.array([next(k for k, d in enumerate(distinct) if d is m) for m in lanes], dtype=np.int32)
    ids = np.arange(len(lanes), dtype=np.int32)
    i32 = ctypes.POINTER(ctypes.c_int32)
    assert ws._L.aqc_ws_mps_to_vec_batch(ws.handle, len(lanes), slots.ctypes.data_as(i32), BUF_Y, ids.ctypes.data_as(i32)) == 0
    ws._touch(BUF_Y)
    out = ws.download(BUF_Y)
    for i, m in enumerate(lanes):
        assert maxdiff(out[i], orc.mps_to_vector(m)) < TOL
    ws.close()


@pytest.mark.parametrize("objective", ["sur_max", "sur_fast_mps_trotter"])
def test_horizon_driver_follows_the_reference_logic(objective):
    """time_evol_best_init.py:118-140,221-334 replayed with the oracle: ground-truth (10x steps) and reference targets, the
    threshold rule, the three fidelity figures of the record at the returned thetas, and the expansion loop."""
    from aqc_research_amd.model_sp_lhs import time_evol as te
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz

    n = 8
    opts = te.UserOptions(num_qubits=n, num_horizons=2, num_layers_inc=1, maxiter=15, objective=objective, fidelity_thr=0.97)
    assert list(opts.trotter_steps) == [3, 6] and list(opts.evol_times) == [1.2, 2.4] and opts.trunc_thr == 1e-6
    assert opts.use_mps == (objective != "sur_max") and opts.ini_state_index() == neel_state_index(n)
    res = te.run_simulation(opts)
    ini = np.zeros(1 << n, complex)
    ini[neel_state_index(n)] = 1

    def trotter_vec(steps, t):
        c = trotter_ansatz(n, steps, True)
        th = init_ansatz_to_trotter(c, np.zeros(c.num_thetas), evol_time=t, delta=1.0)
        return orc.v_mul_vec(c, th, ini)

    for i, r in enumerate(res):
        steps, t = int(opts.trotter_steps[i]), float(opts.evol_times[i])
        t1_gt, t1 = trotter_vec(10 * steps, t), trotter_vec(steps, t)
        f_t1 = abs(np.vdot(t1, t1_gt)) ** 2
        assert r["status"] == "ok" and r["num_trotter_steps"] == steps and r["evol_time1"] == t and r["block_reps"] == 3
        assert abs(r["fid_t1_vs_gt"] - f_t1) < TOL
        assert abs(r["fidelity_thr"] - max(f_t1, 0.97)) < TOL                      # _calc_fidelity_threshold
        a1 = orc.v_mul_vec(trotter_ansatz(n, r["num_layers"], True), r["thetas"], ini)
        assert abs(r["fid_a1_vs_gt"] - abs(np.vdot(a1, t1_gt)) ** 2) < TOL
        assert abs(r["fid_a1_vs_t1"] - abs(np.vdot(a1, t1)) ** 2) < TOL
        assert r["use_mps"] == opts.use_mps and r["entangler"] == "cx" and r["second_order_trotter"]
        assert r["fid_a1_vs_gt"] >= r["fidelity_trotter_init"] - 1e-9
        assert r["num_layers"] == i + 1 and r["expansions"] == 0
    # automatic threshold: 1.03 x fidelity(|t1>, |t1_gt>)
    tgt = te.generate_target(opts, 1)
    thr, f = te._calc_fidelity_threshold(tgt, None)
    assert abs(thr - 1.03 * f) < 1e-15 and abs(f - res[1]["fid_t1_vs_gt"]) < TOL
    # expansion loop: an unreachable threshold makes every allowed expansion happen, one more layer each
    hard = te.UserOptions(num_qubits=n, num_horizons=1, num_layers_inc=1, maxiter=3, objective=objective, fidelity_thr=1.0,
                          num_expansions=2)
    r = te.run_simulation(hard)[0]
    assert r["expansions"] == 2 and r["num_layers"] == 3 and r["thetas"].size == trotter_ansatz(n, 3, True).num_thetas


def test_state_prep_func_returning_a_circuit():
    """objective_base.py:298-303: a reference caller hands ``neel_init_state`` (a circuit of X gates) as ``state_prep_func``;
    the objective must behave exactly as with the equivalent bit mask, and refuse circuits with gates it does not know."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax
    from aqc_research_amd.model_sp_lhs.trotter import neel_init_state, neel_state_index

    n = 7
    rng = np.random.default_rng(7)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 1), second_order=True)
    target, th = orc.rand_state(n, rng), orc.rand_thetas(circ.num_thetas, rng)
    vals = []
    for prep in (neel_init_state, lambda k: neel_state_index(k)):
        user = dict(num_qubits=n, max_flips=1, state_prep_func=prep, enable_optim_stats=False, verbose=0, maxiter=5)
        o = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
        o.set_target(target)
        vals.append((o.objective(th), o.gradient(th)))
    assert vals[0][0] == vals[1][0] and maxdiff(vals[0][1], vals[1][1]) == 0.0
    ref = orc.SurMaxOracle(circ, target, 1, None, True, base_index=neel_state_index(n))
    assert abs(vals[0][0] - ref.objective(th)) < TOL and maxdiff(vals[0][1], ref.gradient(th)) < TOL

    class Odd:   # a circuit with a gate outside the set the state handlers know (round 5: general preparations such as H go through
        num_qubits = n   # GenericStateHandler, tests/test_hip_round5.py; only unknown gates are refused)

        class _I:
            class operation:
                name = "ccx"
                params = ()
            qubits = (0, 1, 2)
        data = [_I()]

    with pytest.raises(NotImplementedError):
        SpSurrogateObjectiveMax(user_parameters=dict(num_qubits=n, max_flips=1, state_prep_func=lambda k: Odd()), circ=circ)


def test_rccl_init_is_bounded_when_a_rank_never_joins():
    """ncclCommInitRank blocks until every rank has joined; aqc_comm_create runs it in a helper thread and gives up after
    AQC_COMM_INIT_TIMEOUT_S with an error naming the cause (a launch whose rank died before the rendezvous must
    fail, not hang).  Rank 0 of a 2-rank communicator whose rank 1 never starts; in a child process, which exits hard
    afterwards (the helper thread is still inside RCCL)."""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import ctypes, os, sys, time
        os.environ["AQC_COMM_INIT_TIMEOUT_S"] = "3"
        from aqc_research_amd import _lib
        L = _lib.lib()
        buf = ctypes.create_string_buffer(128)
        _lib.check(L.aqc_comm_unique_id(buf))
        h = ctypes.c_void_p()
        t0 = time.time()
        rc = L.aqc_comm_create(buf.raw, 2, 0, 0, ctypes.byref(h))
        dt = time.time() - t0
        msg = L.aqc_last_error().decode()
        print(rc, round(dt, 2), msg, flush=True)
        ok = rc != 0 and 2.5 < dt < 30 and "AQC_COMM_INIT_TIMEOUT_S" in msg and not h.value
        os._exit(0 if ok else 1)
    """)
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr


def _cd_problem(n, layout, depth, seed, ent="cx"):
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure

    rng = np.random.default_rng(seed)
    circ = ParametricCircuit(n, ent, create_ansatz_structure(n, layout, "full", depth))
    d = 1 << n
    u = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
    return circ, orc.rand_thetas(circ.num_thetas, rng), np.ascontiguousarray(u), rng


@pytest.mark.parametrize("n,ent,steps", [(5, "cx", 1), (5, "cx", 2), (5, "cx", 5), (5, "cx", 17), (5, "cz", 19), (3, "cx", 12), (6, "cx", 3), (2, "cz", 7)])
def test_coordinate_descent_single_steps_at_the_north_star_tolerance(n, ent, steps):
    """core_op_matrix.py:833-850 (_delta_theta) and :852-912: the walk stopped after a few parameter updates agrees with the
    oracle to 1e-10 on every theta -- the per-step arithmetic is exact; only the chain of ~T sequential Newton steps of a
    whole sweep amplifies rounding (tests/test_hip_parity_mat_obj.py checks full sweeps at 1e-9 / 1e-8)."""
    from aqc_research_amd.core_op_matrix import coord_descent_sweeps

    circ, th, u, _ = _cd_problem(n, "spin", 12, 500 + 10 * n + steps, ent)
    ref, f_ref = orc.coord_descent_single_sweep(circ, th, u, max_steps=steps)
    got = th.copy()
    f = coord_descent_sweeps(circ, got, u, 1, max_steps=steps)
    assert f.shape == (1, 1)
    assert int((np.abs(ref - th) > 0).sum()) == steps               # exactly `steps` parameters moved ...
    assert maxdiff(got, ref) < TOL and abs(f[0, 0] - f_ref) < TOL   # ... to the oracle's values


def test_coordinate_descent_one_launch_lanes_and_sweeps(monkeypatch):
    """aqc_ws_cd_sweeps: lanes = random restarts and their own targets, several sweeps in one launch (z = V^H U re-derived in
    the kernel at the start of each, core_op_matrix.py:806-810).  Every lane follows the oracle's consecutive single sweeps
    (full sweeps: rounding is amplified along ~T sequential steps, hence 1e-8 after the first and 1e-7 after the second),
    the reference-signature single-lane call gives the same numbers, and so does the launch chain it replaced."""
    from aqc_research_amd.core_op_matrix import coord_descent_single_sweep, coord_descent_sweeps

    n, lanes = 5, 5
    circ, th0, u0, rng = _cd_problem(n, "cyclic_spin", 30, 77)
    d = 1 << n
    ths = np.stack([th0] + [orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes - 1)])
    us = np.stack([u0] + [np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0] for _ in range(lanes - 1)])
    got = ths.copy()
    f = coord_descent_sweeps(circ, got, us, 2)
    assert f.shape == (lanes, 2)
    for b in range(lanes):
        t1, f1 = orc.coord_descent_single_sweep(circ, ths[b], us[b])
        t2, f2 = orc.coord_descent_single_sweep(circ, t1, us[b])
        assert abs(f[b, 0] - f1) < 1e-8 and abs(f[b, 1] - f2) < 1e-7 and maxdiff(got[b], t2) < 1e-7
        assert f2 < f1 < 1.0
    one = ths[1].copy()                                  # reference signature, one lane, one sweep at a time
    g1 = coord_descent_single_sweep(circ, one, us[1], None)
    g2 = coord_descent_single_sweep(circ, one, us[1], None)
    assert abs(g1 - f[1, 0]) < 1e-12 and abs(g2 - f[1, 1]) < 1e-10 and maxdiff(one, got[1]) < 1e-10
    monkeypatch.setenv("AQC_CD_CHAIN", "1")              # the launch chain (what problems beyond 6 qubits still use)
    chain = ths[1].copy()
    c1 = coord_descent_single_sweep(circ, chain, us[1], None)
    assert abs(c1 - f[1, 0]) < 1e-8
    monkeypatch.delenv("AQC_CD_CHAIN")
    # one target shared by all lanes; a 7-qubit problem does not fit one workgroup's LDS and says so
    shared = ths.copy()
    fs = coord_descent_sweeps(circ, shared, u0, 1)
    assert abs(fs[0, 0] - f[0, 0]) < 1e-12
    big, thb, ub, _ = _cd_problem(7, "spin", 8, 78)
    with pytest.raises(RuntimeError, match="do not fit"):
        coord_descent_sweeps(big, thb.copy(), ub, 1)
    fb = coord_descent_single_sweep(big, thb, ub, None)   # ... while the reference-signature call runs it on the chain
    assert 0.0 < fb < 1.0


def test_jacobi_svd_on_a_graded_two_site_tensor_that_did_not_converge():
    """tests/golden/svd_graded_26.npz: the 26 x 26 two-site matrix of a 32-qubit Trotter state (singular values from 0.86 down
    to 1e-22) on which the engine's SVD gave up after 60 sweeps -- rotating a numerically zero column over and over shrank it
    until its squared norm underflowed.  Columns below 1e-15 of the Frobenius norm are now left alone (all three kernels)."""
    import os

    from aqc_research_amd.mps_engine import svd
    from tests.helpers import GOLDEN

    m = np.load(os.path.join(GOLDEN, "svd_graded_26.npz"))["m"]
    ref = np.linalg.svd(m, compute_uv=False)
    for a in (m, m.conj().T, np.vstack([m, 1e-3 * m]), np.kron(np.diag([1.0, 1e-9, 1e-18]), m)):   # small / small / small / blocked kernel
        u, s, vh, sweeps = svd(a)
        assert 0 < sweeps < 30
        assert maxdiff((u * s) @ vh, a) < 1e-14
        good = s > 1e-13 * s[0]
        assert maxdiff(u[:, good].conj().T @ u[:, good], np.eye(int(good.sum()))) < 1e-11
    u, s, vh, _ = svd(m)
    assert maxdiff(s, ref) < 1e-15


def test_mps_engine_builds_a_32_qubit_trotter_target_without_truncation():
    """The walk that hit the non-converging SVD: V|neel> for a 32-qubit 2nd-order Trotter circuit of 6 layers at trunc_thr =
    1e-12, C-side loop and gate-by-gate calls; norm 1 and equal tensors' bonds on both routes."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index

    n = 32
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 6), second_order=True)
    th = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2, delta=1.0)
    basis = me.DeviceMPS.basis_state(n, neel_state_index(n))
    a = me.v_mul_mps(circ, th, basis, trunc_thr=1e-12, method="single")
    b = me._apply_circuit_gatewise(circ, th, basis.clone(), False, 1e-12, 0)
    assert abs(a.dot(a) - 1.0) < 1e-9 and abs(abs(a.dot(b)) - 1.0) < 1e-9
    assert a.bond_dims.max() <= 32 and list(a.bond_dims) == list(b.bond_dims)
    # the same circuit layer by layer on one lockstep lane (what v_mul_mps does by default while bonds stay <= 32), handed out as an MPS
    c = me.v_mul_mps(circ, th, basis, trunc_thr=1e-12, method="lockstep")
    assert list(c.bond_dims) == list(a.bond_dims) and abs(c.dot(a) - 1.0) < 1e-11 and abs(c.discarded_weight - a.discarded_weight) < 1e-12
    back = me.v_dagger_mul_mps(circ, th, c, trunc_thr=1e-12, method="lockstep")
    assert abs(abs(back.dot(basis)) - 1.0) < 1e-9
    for m in (a, b, c, back, basis):
        m.close()


def test_mps_engine_lanes_on_host_threads_match_single_lane_calls():
    """mps_engine.evaluate_lanes: V^H, <lhs|.> and the gate-by-gate gradient for several lanes at once (host threads, every MPS on
    its own stream, one synchronisation per 2-qubit gate) -- each lane equal to the same calls made one after the other, and, at
    trunc_thr -> 0 on a register that still fits, to the dense oracle (mps_dot_objective.py:41-242 per lane)."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import neel_state_index

    n, lanes = 10, 5
    rng = np.random.default_rng(105)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 2), second_order=True)
    ths = np.stack([0.4 * orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
    neel = neel_state_index(n)
    basis = me.DeviceMPS.basis_state(n, neel)
    tmps = [orc.random_mps(n, 4, rng) for _ in range(lanes)]
    targets = [me.DeviceMPS.from_qiskit(m) for m in tmps]
    h, g = me.evaluate_lanes(circ, ths, targets, basis, trunc_thr=0.0, method="threads")
    x = np.zeros(1 << n, complex)
    x[neel] = 1
    for b in range(lanes):
        vh = me.v_dagger_mul_mps(circ, ths[b], targets[b], method="single")
        assert abs(h[b] - basis.dot(vh)) < 1e-13
        assert maxdiff(g[b], me.fast_dot_gradient_mps(circ, ths[b], basis, vh, method="single")) < 1e-13
        vh.close()
        dense = orc.v_dagger_mul_vec(circ, ths[b], orc.mps_to_vector(tmps[b]))
        assert abs(h[b] - dense[neel]) < TOL
        assert maxdiff(g[b], orc.grad_of_dot_product(circ, ths[b], x, dense)) < TOL
    for m in targets + [basis]:
        m.close()


def _single_lane_reference(me, circ, th, target, lhs, **kw):
    grad_kw = dict(kw)
    vh = me.v_dagger_mul_mps(circ, th, target, trunc_thr=kw.get("trunc_thr", 0.0), max_bond=kw.get("max_bond", 0), method="single")
    try:
        apply_kw = {k: grad_kw[k] for k in ("trunc_thr", "max_bond") if k in grad_kw}
        return lhs.dot(vh), me.fast_dot_gradient_mps(circ, th, lhs, vh, method="single", **grad_kw), vh.discarded_weight, int(vh.bond_dims.max()), apply_kw
    finally:
        vh.close()


@pytest.mark.parametrize("entangler", ["cx", "cz", "cp"])
def test_lockstep_lanes_match_the_single_lane_engine_and_the_dense_oracle(entangler):
    """aqc_mpsb_eval: all lanes walk the ansatz together (one launch per step, one rank read-back per 2-qubit gate).  A generic
    ansatz with long-range blocks (swap routing), every lane its own target with its own bonds and its own lhs state, exact arithmetic:
    every lane equals the single-lane engine and the dense oracle (mps_dot_objective.py:41-242 per lane)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd import mps_engine as me

    n, lanes = 8, 6
    rng = np.random.default_rng(711)
    blocks = np.array([[0, 3, 5, 7, 2, 6, 1, 4], [1, 0, 2, 3, 6, 1, 7, 5]])
    circ = ParametricCircuit(n, entangler=entangler, blocks=blocks)
    ths = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
    tmps = [orc.random_mps(n, 1 + b % 4, rng) for b in range(lanes)]
    lmps = [orc.random_mps(n, 1 + (b + 1) % 3, rng) for b in range(lanes)]
    targets = [me.DeviceMPS.from_qiskit(m) for m in tmps]
    lhs = [me.DeviceMPS.from_qiskit(m) for m in lmps]
    h, g = me.evaluate_lanes(circ, ths, targets, lhs, method="lockstep")
    for b in range(lanes):
        h1, g1, *_ = _single_lane_reference(me, circ, ths[b], targets[b], lhs[b])
        assert abs(h[b] - h1) < 1e-12 and maxdiff(g[b], g1) < 1e-12
        x = orc.mps_to_vector(lmps[b])
        dense = orc.v_dagger_mul_vec(circ, ths[b], orc.mps_to_vector(tmps[b]))
        assert abs(h[b] - np.vdot(x, dense)) < TOL
        assert maxdiff(g[b], orc.grad_of_dot_product(circ, ths[b], x, dense)) < TOL
    # block range and front layer switched off, one shared target
    h2, g2 = me.evaluate_lanes(circ, ths, targets[3], lhs, block_range=(2, 5), front_layer=False, method="lockstep")
    for b in range(lanes):
        h1, g1, *_ = _single_lane_reference(me, circ, ths[b], targets[3], lhs[b], block_range=(2, 5), front_layer=False)
        assert abs(h2[b] - h1) < 1e-12 and maxdiff(g2[b], g1) < 1e-12
        tpb = 5 if entangler == "cp" else 4
        assert np.all(g2[b][: 3 * n] == 0) and np.all(g2[b][3 * n + 5 * tpb:] == 0) and np.all(g2[b][3 * n: 3 * n + 2 * tpb] == 0)
    # a state edited in place is copied into the lanes again (DeviceMPS.version), an unchanged one is not
    targets[0].gate1(np.array([[0, 1], [1, 0]], dtype=complex), 2)
    h3, g3 = me.evaluate_lanes(circ, ths, targets, lhs, method="lockstep")
    h1, g1, *_ = _single_lane_reference(me, circ, ths[0], targets[0], lhs[0])
    assert abs(h3[0] - h1) < 1e-12 and maxdiff(g3[0], g1) < 1e-12 and abs(h3[0] - h[0]) > 1e-6
    assert abs(h3[1] - h[1]) < 1e-14 and maxdiff(g3[1], g[1]) < 1e-14
    for m in targets + lhs:
        m.close()


def test_lockstep_lanes_truncate_like_the_single_lane_engine():
    """Truncation is decided per lane from the lane's own singular values by the single-lane rule (threshold on the discarded weight,
    bond cap): objective, gradient, discarded weight and largest bond of every lane equal the single-lane engine's at trunc_thr =
    1e-6 and with max_bond = 6; the 2nd-order Trotter ansatz of the ASP driver on the Neel state."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index

    n, lanes = 12, 4
    rng = np.random.default_rng(712)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 3), second_order=True)
    th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.9, delta=1.0)
    ths = np.stack([th0 + 0.05 * orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
    basis = me.DeviceMPS.basis_state(n, neel_state_index(n))
    target = me.v_mul_mps(circ, th0, basis, trunc_thr=1e-12)
    ls = me.LockstepLanes(n, lanes).set_targets(target).set_lhs(basis)
    for kw in (dict(trunc_thr=1e-6), dict(trunc_thr=1e-9, max_bond=6)):
        h, g, disc, bonds = ls.evaluate(circ, ths, details=True, **kw)
        for b in range(lanes):
            h1, g1, d1, b1, _ = _single_lane_reference(me, circ, ths[b], target, basis, **kw)
            assert abs(h[b] - h1) < 1e-12 and maxdiff(g[b], g1) < 1e-12
            assert abs(disc[b] - d1) < 1e-15 + 1e-9 * d1 and bonds[b] == b1
        if "max_bond" in kw:
            assert bonds.max() <= 6
    assert abs(h[0]) > 0.3                     # near the Trotter point the overlap is large: the lanes did real work
    ls.close()
    for m in (target, basis):
        m.close()


def test_lockstep_lanes_refuse_to_truncate_silently_and_auto_falls_back():
    """A lane whose bond would pass 32 fails the lockstep call loudly; method='auto' then answers from the single-lane engine."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd import mps_engine as me

    n, lanes = 14, 2
    rng = np.random.default_rng(713)
    blocks = np.array([[q % n for q in range(40)], [(q + 1) % n for q in range(40)]])
    circ = ParametricCircuit(n, entangler="cx", blocks=blocks)
    ths = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
    tmps = [orc.random_mps(n, 8, rng) for _ in range(lanes)]
    targets = [me.DeviceMPS.from_qiskit(m) for m in tmps]
    basis = me.DeviceMPS.basis_state(n, 0)
    with pytest.raises(RuntimeError, match="lockstep lanes"):
        me.evaluate_lanes(circ, ths, targets, basis, method="lockstep")
    h, g = me.evaluate_lanes(circ, ths, targets, basis, method="auto")
    for b in range(lanes):
        dense = orc.v_dagger_mul_vec(circ, ths[b], orc.mps_to_vector(tmps[b]))
        assert abs(h[b] - dense[0]) < TOL
    # with a bond cap inside the lockstep range the same batch runs in lockstep and equals the capped single-lane engine
    h, g = me.evaluate_lanes(circ, ths, targets, basis, max_bond=16, method="lockstep")
    for b in range(lanes):
        h1, g1, *_ = _single_lane_reference(me, circ, ths[b], targets[b], basis, max_bond=16)
        assert abs(h[b] - h1) < 1e-12 and maxdiff(g[b], g1) < 1e-12
    for m in targets + [basis]:
        m.close()


def test_batched_mps_surrogate_objective_matches_the_dense_batched_objective():
    """BatchedMpsSurrogateObjective (lockstep lanes of the MPS engine, two phases: V^H + flip amplitudes, then both gradient walks)
    against BatchedSurrogateObjective on the dense kernels (itself pinned by the oracle): values, gradients and the state machine of
    objective_lhs_sur_max.py:99-191 -- one lane starts next to a flip state, so the leading state changes on the way."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.batched_optimizer import BatchedMpsSurrogateObjective, BatchedSurrogateObjective
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index

    n, lanes = 10, 3
    rng = np.random.default_rng(731)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 2), second_order=True)
    neel = neel_state_index(n)
    th_star = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=0.8, delta=1.0)
    dense = []
    for b in range(lanes):   # lane b: V(th*)|neel> for b = 0, 2; V(th*) X_4|neel> for b = 1
        x = np.zeros(1 << n, complex)
        x[neel ^ (1 << 4) if b == 1 else neel] = 1
        v = orc.v_mul_vec(circ, th_star + 0.1 * b * orc.rand_thetas(circ.num_thetas, rng) / np.pi, x)
        dense.append(v / np.linalg.norm(v))
    from aqc_research_amd.mps_operations import vector_to_exact_mps

    targets = [me.DeviceMPS.from_qiskit(vector_to_exact_mps(v)) for v in dense]
    ref = BatchedSurrogateObjective(circ, np.stack(dense), base_index=neel)
    obj = BatchedMpsSurrogateObjective(circ, targets, base_index=neel, trunc_thr=0.0)
    th = np.stack([th_star + 0.03 * orc.rand_thetas(circ.num_thetas, rng) / np.pi for _ in range(lanes)])
    for step, upd in enumerate((True, False, True, True)):
        f0, g0 = ref.value_and_grad(th, update_state=upd)
        f1, g1 = obj.value_and_grad(th, update_state=upd)
        assert maxdiff(f1, f0) < TOL and maxdiff(g1, g0) < TOL
        assert list(obj.max_no) == list(ref.max_no) and maxdiff(obj.weight, ref.weight) < 1e-12
        th = th - 0.05 * g0
    assert obj.max_no[1] == 5 and obj.max_no[0] == 0      # lane 1 is led by the flip of qubit 4 (state 5), lane 0 by |state_0>
    ref.close()
    obj.close()
    for m in targets:
        m.close()


def test_seeded_horizons_of_the_mps_objective_run_on_the_lockstep_lanes(monkeypatch):
    """Random restarts of a horizon with objective = sur_fast_mps_trotter: up to 24 qubits on the dense batched objective (the
    target's dense state), beyond on the lockstep lanes of the MPS engine (BatchedMpsSurrogateObjective) -- forced here at 10
    qubits by lowering the dense limit; with trunc_thr -> 0 both routes and the state-vector objective give the same restarts."""
    from aqc_research_amd.model_sp_lhs import time_evol as te

    kw = dict(num_qubits=10, num_horizons=1, num_layers_inc=1, maxiter=8, fidelity_thr=0.999, num_seeds=3, vectorised_lbfgs=True, seed=5,
              trunc_thr=1e-14, trunc_thr_target=1e-14)
    ref = te.run_simulation(te.UserOptions(objective="sur_max", **kw))[0]
    dense = te.run_simulation(te.UserOptions(objective="sur_fast_mps_trotter", **kw))[0]
    monkeypatch.setattr(te, "_DENSE_MAX_QUBITS", 8)
    eng = te.run_simulation(te.UserOptions(objective="sur_fast_mps_trotter", **kw))[0]
    for r in (dense, eng):
        assert r["status"] == "ok" and len(r["fidelities"]) == 3
        assert maxdiff(np.array(r["fidelities"]), np.array(ref["fidelities"])) < 1e-6 and r["best_restart"] == ref["best_restart"]
    with pytest.raises(ValueError, match="vectorised"):
        te._seeded_horizon_job(0, {"opts": te.UserOptions(objective="sur_fast_mps_trotter", **dict(kw, vectorised_lbfgs=False)), "horizon": 1})


def test_horizon_driver_beyond_dense_reach():
    """One horizon of the ASP driver at 32 qubits with the MPS objective (time_evol_best_init.py:221-334): targets (10x-step ground truth and
    reference) built by the engine, the optimisation on the objective's two lockstep lanes, the record's fidelities from MPS inner products
    (no helper on the way may want a dense workspace of the register's size: mps_dot did, and failed beyond 30 qubits)."""
    from aqc_research_amd.model_sp_lhs import time_evol as te

    opts = te.UserOptions(num_qubits=32, num_horizons=1, num_layers_inc=1, maxiter=4, objective="sur_fast_mps_trotter", fidelity_thr=0.9999)
    r = te.run_simulation(opts)[0]
    assert r["status"] == "ok" and r["use_mps"] and r["num_layers"] == 1 and r["thetas"].size == 3 * 32 + 4 * 93
    assert 0.0 < r["fid_t1_vs_gt"] <= 1.0 + 1e-9 and 0.0 < r["fid_a1_vs_gt"] <= 1.0 + 1e-9
    assert r["fid_a1_vs_gt"] >= r["fidelity_trotter_init"] - 1e-6        # the optimisation does not lose what the Trotter point had


def test_seeded_horizon_falls_back_when_the_target_outgrows_the_lanes(monkeypatch):
    """A long evolution makes the untruncated target's bonds exceed the lockstep lanes' 32 (12 qubits: up to 64): the restarts of the
    horizon then run one after the other on the objective's single-lane route and the record says so."""
    from aqc_research_amd.model_sp_lhs import time_evol as te

    monkeypatch.setattr(te, "_DENSE_MAX_QUBITS", 8)
    opts = te.UserOptions(num_qubits=12, num_horizons=1, evol_time_step=6.0, trotter_steps_per_horizon=12, num_layers_inc=1, maxiter=2,
                          objective="sur_fast_mps_trotter", fidelity_thr=0.9999, num_seeds=2, vectorised_lbfgs=True, trunc_thr=1e-12)
    tgt = te.generate_target(opts, 0)
    assert max(np.shape(g0)[1] for g0, _ in tgt.t1_gt[0]) > 32
    r = te.run_simulation(opts)[0]
    assert r["status"] == "ok" and r["route"].startswith("single-lane") and len(r["fidelities"]) == 2
    assert all(0.0 <= f <= 1.0 + 1e-9 for f in r["fidelities"])


def test_function_level_mps_front_door_at_34_qubits():
    """The reference's own functions on QiskitMPS tuples beyond every dense limit (mps_operations.py:192-213,326-371;
    mps_dot_objective.py:41-242,245-516): v_mul_mps, v_dagger_mul_mps, mps_dot, fast_dot_gradient, a single-gate function and dot_z agree with
    each other -- the gradient with central differences of <V(theta) 0|phi>, dot_z with the gradient entry of its rotation."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_dot_objective as mdo
    from aqc_research_amd import mps_operations as mpo
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit

    n = 34
    rng = np.random.default_rng(734)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 1), second_order=False)
    th = 0.3 * orc.rand_thetas(circ.num_thetas, rng)
    zero = ([(np.ones((1, 1), complex), np.zeros((1, 1), complex)) for _ in range(n)], [np.ones(1) for _ in range(n - 1)])
    phi = mpo.v_mul_mps(circ, th + 0.05, zero, trunc_thr=1e-14)
    assert mpo.check_mps(phi) and mpo.mps_num_qubits(phi) == n and abs(mpo.mps_dot(phi, phi) - 1.0) < 1e-9
    vh = mpo.v_dagger_mul_mps(circ, th, phi, trunc_thr=1e-14)
    g = mdo.fast_dot_gradient(circ, th, zero, vh, trunc_thr=1e-14)
    assert g.shape == (circ.num_thetas,) and np.isfinite(g).all()
    for t in (2, 3 * n + 5):
        e = np.zeros_like(th); e[t] = 1e-5
        f = [mpo.mps_dot(mpo.v_mul_mps(circ, th + s * e, zero, trunc_thr=1e-14), phi) for s in (+1, -1)]
        assert abs((f[0] - f[1]) / 2e-5 - g[t]) < 1e-7
    # theta 2 is the first rotation applied to qubit 0 (Rz): its gradient entry is 0.5j <Z_0 0|V^H phi> = dot_z(0, |0>, vh)
    assert abs(mdo.dot_z(0, zero, vh) - g[2]) < 1e-9
    w = mdo.cx_mul_mps(0.0, 3, 4, mdo.rx_mul_mps(0.7, 3, zero), trunc_thr=1e-14)
    assert abs(mpo.mps_dot(w, w) - 1.0) < 1e-12 and max(np.shape(g0)[1] for g0, _ in w[0]) == 2


def test_lockstep_lanes_against_the_single_lane_engine_on_random_problems():
    """tools/lockstep_fuzz.py in small: random registers, entanglers, block layouts with long-range pairs, targets / lhs states of random
    bonds, truncation thresholds, bond caps, block ranges -- every lane of the lockstep walk equals the single-lane engine's (or the
    lockstep call refuses because a bond passes 32, never answers differently)."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd import mps_engine as me

    rng = np.random.default_rng(29)
    compared = 0
    for _ in range(24):
        n = int(rng.integers(3, 11))
        ent = ("cx", "cz", "cp")[int(rng.integers(0, 3))]
        nb = int(rng.integers(1, 3 * n))
        ctrl = rng.integers(0, n, nb)
        circ = ParametricCircuit(n, entangler=ent, blocks=np.stack([ctrl, (ctrl + rng.integers(1, n, nb)) % n]))
        lanes = int(rng.integers(1, 4))
        thr = (0.0, 1e-10, 1e-6, 1e-3)[int(rng.integers(0, 4))]
        cap = (0, 0, 3, 8)[int(rng.integers(0, 4))]
        ths = np.stack([orc.rand_thetas(circ.num_thetas, rng) for _ in range(lanes)])
        targets = [me.DeviceMPS.from_qiskit(orc.random_mps(n, int(rng.integers(1, 9)), rng), trunc_thr=thr) for _ in range(lanes)]
        lhs = [me.DeviceMPS.from_qiskit(orc.random_mps(n, int(rng.integers(1, 4)), rng), trunc_thr=thr) for _ in range(lanes)]
        lo = int(rng.integers(0, nb))
        kw = dict(trunc_thr=thr, max_bond=cap, block_range=None if rng.random() < 0.5 else (lo, int(rng.integers(lo, nb + 1))),
                  front_layer=bool(rng.random() < 0.7))
        try:
            h, g = me.evaluate_lanes(circ, ths, targets, lhs, method="lockstep", **kw)
        except RuntimeError as err:
            assert "lockstep lanes" in str(err)
            continue
        hs, gs = me.evaluate_lanes(circ, ths, targets, lhs, method="threads", **kw)
        assert maxdiff(h, hs) < 1e-11 and maxdiff(g, gs) < 1e-11, (n, ent, nb, lanes, thr, cap, kw)
        compared += 1
        for m in targets + lhs:
            m.close()
    assert compared >= 18
