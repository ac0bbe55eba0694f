"""GPU: jobs that share an ansatz advance in lockstep as lanes of one batched workspace and give the
results of one-at-a-time runs (aqc_research_amd/lockstep.py)."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import TOL, maxdiff

pytestmark = pytest.mark.gpu


def _setup(n=8, layers=2):
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz, trotter_state

    circ = trotter_ansatz(n, layers, True)
    neel = neel_state_index(n)
    cases = []
    rng = np.random.default_rng(99)
    for j in range(5):
        t = 0.6 + 0.3 * j
        target = trotter_state(n, evol_time=t, num_steps=6, delta=1.0, second_order=True)
        th0 = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=t, delta=1.0) + 0.05 * rng.standard_normal(circ.num_thetas)
        cases.append((target, th0))
    return circ, neel, cases


def _optimise(circ, neel, target, th0, workspace=None, maxiter=12):
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax
    from aqc_research_amd.optimizer import AqcOptimizer

    user = dict(num_qubits=circ.num_qubits, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False,
                verbose=0, maxiter=maxiter)
    if workspace is not None:
        user["workspace"] = workspace
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    f0 = objv.objective(th0)
    g0 = objv.gradient(th0)
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=maxiter).optimize(objv, circ, th0)
    return {"f0": f0, "g0": g0, "cost": res["cost"], "fidelity": res["fidelity"], "thetas": res["thetas"], "nfev": res["num_fun_ev"]}


def test_lockstep_equals_sequential():
    from aqc_research_amd.lockstep import LockstepBatch

    circ, neel, cases = _setup()
    seq = [_optimise(circ, neel, t, th) for t, th in cases]
    batch = LockstepBatch(circ, nlanes=len(cases))
    out = batch.run([(lambda view, t=t, th=th: _optimise(circ, neel, t, th, workspace=view)) for t, th in cases])
    batch.close()
    assert batch.native_calls < sum(r["nfev"] for r in seq) + 3 * len(cases)   # requests really were merged
    for a, b in zip(seq, out):
        assert not isinstance(b, BaseException), b
        assert abs(a["f0"] - b["f0"]) < TOL and maxdiff(a["g0"], b["g0"]) < TOL
        # same optimizer on (numerically) the same function: trajectories agree far below optimisation accuracy
        assert abs(a["cost"] - b["cost"]) < 1e-7 and abs(a["fidelity"] - b["fidelity"]) < 1e-7
        assert b["fidelity"] > 0.99


def test_lockstep_mixed_requests_against_oracle():
    """Lanes issue different call signatures in the same round and finish at different times."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.engine import BUF_X, BUF_X2, BUF_Y
    from aqc_research_amd.lockstep import LockstepBatch

    n = 7
    rng = np.random.default_rng(5)
    a = orc.Ansatz(n, "cz", orc.spin_blocks(n, 9))
    circ = ParametricCircuit(n, "cz", a.blocks)
    batch = LockstepBatch(circ, nlanes=4)
    idx = orc.flip_state_indices(n, 1)
    data = [(orc.rand_state(n, rng), [orc.rand_thetas(a.num_thetas, rng) for _ in range(3)]) for _ in range(4)]

    def job(view, lane):
        y, ths = data[lane]
        view.upload(BUF_Y, y)
        view.set_basis(BUF_X, int(idx[0]))
        view.gather_setup(idx)
        errs = []
        for r in range(lane + 1 if lane < 3 else 3):   # lanes run 1, 2, 3, 3 rounds
            th = ths[r]
            hs, g = view.eval(th, vdag=True, gather=True, grad=True, x_buf=BUF_X, block_range=(2, 7), front_layer=False)
            vh = orc.v_dagger_mul_vec(a, th, y)
            x = np.zeros(1 << n, complex); x[idx[0]] = 1
            errs.append(maxdiff(hs[0], vh[idx]))
            errs.append(maxdiff(g[0], orc.grad_of_dot_product(a, th, x, vh, (2, 7), False)))
            if lane % 2:   # odd lanes ask for a second sweep from another basis state: a different signature
                view.set_basis(BUF_X2, int(idx[1 + lane]))
                _, g2 = view.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=None, front_layer=True)
                x2 = np.zeros(1 << n, complex); x2[idx[1 + lane]] = 1
                errs.append(maxdiff(g2[0], orc.grad_of_dot_product(a, th, x2, vh)))
        return max(errs)

    out = batch.run([(lambda view, lane=lane: job(view, lane)) for lane in range(4)])
    batch.close()
    for e in out:
        assert not isinstance(e, BaseException), e
        assert e < TOL


def test_lockstep_objects_with_leading_flip_states():
    """Surrogate objects on lanes of a lockstep batch with random targets (a flip state leads from the first evaluation on):
    every lane's objective()/gradient() pair is one request served by aqc_ws_surrogate_eval for the whole batch; values,
    gradients, leading states and weights of every lane against orc.SurMaxOracle, lanes of different lengths, one lane mixing
    in plain eval requests that use the same lhs buffer."""
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.engine import BUF_X2
    from aqc_research_amd.lockstep import LockstepBatch
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n, lanes = 9, 4
    rng = np.random.default_rng(99)
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, 11))
    circ = ParametricCircuit(n, "cx", a.blocks)
    batch = LockstepBatch(circ, nlanes=lanes)
    idx = orc.flip_state_indices(n, 1)
    data = [(orc.rand_state(n, rng), orc.rand_thetas(a.num_thetas, rng)) for _ in range(lanes)]

    def job(view, lane):
        y, th = data[lane]
        user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: 0, enable_optim_stats=False, verbose=0, workspace=view)
        obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
        obj.set_target(y)
        o = orc.SurMaxOracle(a, y, 1, None, True)
        errs, led = [], 0
        for step in range(2 + lane):
            f, fo = obj.objective(th), o.objective(th)
            g, go = obj.gradient(th), o.gradient(th)
            errs += [abs(f - fo), maxdiff(g, go), abs(obj._weight - o.weight), float(obj._max_no != o.max_no)]
            led += obj._max_no != 0
            if lane == 1:   # somebody else's request on the lhs buffer between two pairs
                view.set_basis(BUF_X2, int(idx[2]))
                _, g2 = view.eval(None, vdag=False, gather=False, grad=True, x_buf=BUF_X2, block_range=None, front_layer=True)
                x2 = np.zeros(1 << n, complex); x2[idx[2]] = 1
                errs.append(maxdiff(g2[0], orc.grad_of_dot_product(a, th, x2, orc.v_dagger_mul_vec(a, th, y))))
            th = th - 0.05 * g
        return max(errs), led

    out = batch.run([(lambda view, lane=lane: job(view, lane)) for lane in range(lanes)])
    batch.close()
    for e in out:
        assert not isinstance(e, BaseException), e
        assert e[0] < TOL and e[1] >= 1


def test_run_jobs_lockstep_records():
    from aqc_research_amd.lockstep import run_jobs_lockstep

    circ, neel, cases = _setup(n=6)

    def job(j, cfg, ws):
        target, th0 = cases[j % len(cases)]
        th0 = th0 + 0.01 * cfg["rng"].standard_normal(th0.size)
        r = _optimise(circ, neel, target, th0, workspace=ws, maxiter=5)
        return {"cost": r["cost"], "fidelity": r["fidelity"]}

    recs = run_jobs_lockstep(circ, [{"tag": i} for i in range(7)], 1000, job, nlanes=3)
    assert [r["job_index"] for r in recs] == list(range(7))
    assert all(r["status"] == "ok" and r["seed"] == 1000 + 7 * (r["job_index"] + 1) for r in recs)
