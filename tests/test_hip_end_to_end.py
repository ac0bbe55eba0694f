"""GPU end-to-end: the objective objects driven by the optimizer wrapper and the job executor, the way
time_evol_best_init._model_function (time_evol_best_init.py:143-218) and aqc_sketching._full_aqc
(aqc_sketching.py:35-50) drive the reference."""
import numpy as np
import pytest

from oracle import aqc_oracle as orc

pytestmark = pytest.mark.gpu


def test_asp_lbfgs_recovers_ansatz_state():
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax
    from aqc_research_amd.optimizer import AqcOptimizer, EarlyStopper, TimeoutChecker

    n = 8
    rng = np.random.default_rng(11)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, 2), second_order=True)
    th_true = 0.25 * orc.rand_thetas(circ.num_thetas, rng)
    zero = np.zeros(1 << n, complex); zero[0] = 1
    target = orc.v_mul_vec(circ, th_true, zero)           # reachable target => fidelity -> 1
    user = dict(num_qubits=n, max_flips=1, enable_optim_stats=True, verbose=0, maxiter=200)
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(target)
    th0 = th_true + 0.05 * rng.standard_normal(th_true.size)
    f_start = objv.objective(th0)
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=200).optimize(
        objv, circ, th0, stopper=EarlyStopper(fidelity_thr=0.9999), timeout=TimeoutChecker(time_limit=600))
    assert res["fidelity"] > 0.999 and res["cost"] < f_start and not res["is_timeout"]
    assert res["stats"]["fobj"].size >= 1 and res["num_grad_ev"] >= 1
    # independent check of the returned parameters with the CPU oracle
    v = orc.v_mul_vec(circ, res["thetas"], zero)
    assert abs(np.vdot(target, v)) ** 2 > 0.999


def test_full_aqc_lbfgs_and_run_jobs():
    from aqc_research_amd import ParametricCircuit
    from aqc_research_amd.circuit_structures import create_ansatz_structure
    from aqc_research_amd.job_executor import run_jobs
    from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx
    from aqc_research_amd.optimizer import AqcOptimizer, SmallObjectiveStopper

    n = 3
    circ = ParametricCircuit(n, "cx", create_ansatz_structure(n, "spin", "full", 14))

    def job(index, cfg):
        rng = np.random.default_rng(cfg["seed"])
        th_true = orc.rand_thetas(circ.num_thetas, rng)
        target = np.ascontiguousarray(orc.v_mul_mat(circ, th_true, np.eye(1 << n, dtype=complex)))
        objv = SketchingObjectiveEx(circ, FullRangeSketchingVectors(target), stop_small_fobj=SmallObjectiveStopper(fobj_thr=1e-9))
        th0 = th_true + 0.1 * rng.standard_normal(th_true.size)
        res = AqcOptimizer(optimizer_name="lbfgs", maxiter=300).optimize(objv, circ, th0)
        v = orc.v_mul_mat(circ, res["thetas"], np.eye(1 << n, dtype=complex))
        return {"cost": res["cost"], "overlap": float(abs(np.vdot(v, target)) / (1 << n)), "nit": objv.num_iterations}

    results = run_jobs([{"seed": s} for s in (1, 2, 3)], 100, job)
    assert [r["status"] for r in results] == ["ok"] * 3
    for r in results:
        assert r["cost"] < 1e-6 and r["overlap"] > 1 - 1e-6


def test_asp_time_evolution_driver():
    """Two horizons of the ASP driver on 8 qubits: device-synthesised Trotter targets, Trotter initial point,
    L-BFGS on the HIP objective; the optimised ansatz must not be worse than its Trotter start."""
    from scipy.linalg import expm

    from aqc_research_amd.model_sp_lhs.time_evol import UserOptions, run_simulation
    from aqc_research_amd.model_sp_lhs.trotter import make_hamiltonian, neel_state_index, trotter_state

    n = 8
    # device Trotter state vs exact evolution
    ini = np.zeros(1 << n, complex); ini[neel_state_index(n)] = 1
    exact = expm(-1j * 1.2 * make_hamiltonian(n, 1.0)) @ ini
    tgt = trotter_state(n, evol_time=1.2, num_steps=12, with_global_phase=True)
    assert np.linalg.norm(tgt - exact) < 2e-3
    opts = UserOptions(num_qubits=n, num_horizons=2, num_layers_inc=1, trotter_steps_per_horizon=6, maxiter=30)
    res = run_simulation(opts)
    assert [r["status"] for r in res] == ["ok", "ok"] and [r["horizon"] for r in res] == [1, 2]
    for r in res:
        assert r["fidelity"] >= r["fidelity_trotter_init"] - 1e-9 and r["fidelity"] > 0.9
        assert r["thetas"].shape == (r["num_thetas"],)


def test_asp_driver_with_lockstep_restarts():
    """Random restarts of a horizon run as lockstep lanes; restart 0 is the plain Trotter start, so the best of
    the restarts can only match or beat the single-start driver."""
    from aqc_research_amd.model_sp_lhs.time_evol import UserOptions, run_simulation

    n = 8
    base = dict(num_qubits=n, num_horizons=2, num_layers_inc=1, trotter_steps_per_horizon=6, maxiter=20)
    single = run_simulation(UserOptions(**base))
    multi = run_simulation(UserOptions(num_seeds=5, theta_jitter=0.02, **base))
    assert [r["status"] for r in multi] == ["ok", "ok"]
    for s, m in zip(single, multi):
        assert len(m["fidelities"]) == 5 and m["thetas"].shape == (m["num_thetas"],)
        assert abs(m["fidelities"][0] - s["fidelity"]) < 1e-7       # restart 0 == the single-start run
        assert m["fidelity"] >= s["fidelity"] - 1e-9


def test_column_sharded_aqc_objective_two_ranks(tmp_path):
    """Full-unitary AQC objective with the sketching columns split over two ranks (gloo all-reduce of the
    (trace, gradient) record; both ranks share this box's one GPU): every rank must return the value and
    gradient of the unsharded evaluation, which itself is checked against the oracle."""
    import os
    import subprocess
    import sys
    import textwrap

    from tests.helpers import free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "shard.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {root!r})
        import numpy as np, torch.distributed as dist
        from oracle import aqc_oracle as orc
        from aqc_research_amd import ParametricCircuit
        from aqc_research_amd.circuit_structures import create_ansatz_structure
        from aqc_research_amd.model_sketching.sk_core import FullRangeSketchingVectors, SketchingObjectiveEx
        dist.init_process_group(backend="gloo")
        from tests.gloo_double import install; install(dist)
        n = 5
        circ = ParametricCircuit(n, "cz", create_ansatz_structure(n, "spin", "full", 11))
        rng = np.random.default_rng(42)
        target = np.linalg.qr(rng.standard_normal((32, 32)) + 1j * rng.standard_normal((32, 32)))[0]
        th = orc.rand_thetas(circ.num_thetas, rng)
        f1, g1 = SketchingObjectiveEx(circ, FullRangeSketchingVectors(target)).objective_and_gradient(th)
        sharded = SketchingObjectiveEx(circ, FullRangeSketchingVectors(target), column_shard=True)
        f2, g2 = sharded.objective_and_gradient(th)
        f3, g3 = sharded.objective_and_gradient(th + 0.01)          # second call re-uses the resident slab
        f0, g0 = orc.sketching_objective_and_gradient(circ, th, np.eye(32, dtype=complex), target)
        f4, g4 = orc.sketching_objective_and_gradient(circ, th + 0.01, np.eye(32, dtype=complex), target)
        err = max(abs(f1 - f0), abs(f2 - f0), abs(f3 - f4), np.abs(g1 - g0).max(), np.abs(g2 - g0).max(), np.abs(g3 - g4).max())
        print("RANK", dist.get_rank(), "ERR", err, flush=True)
        assert err < 1e-10 and sharded._shard is not None
        dist.barrier()
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ERR") == 2


def test_rccl_bound_directly_single_rank(tmp_path):
    """aqc_comm_*: librccl through the C ABI (no torch): unique id, communicator, all-gather, all-reduce, barrier on a
    one-rank group -- what a 1-GPU box can exercise of the multi-GPU transport."""
    from aqc_research_amd.comm import RcclCommunicator

    c = RcclCommunicator(0, 1, 0, str(tmp_path / "id"))
    assert c.rank == 0 and c.size == 1 and "rccl" in c.transport
    x = np.linspace(0, 1, 37)
    assert np.array_equal(c.allgather(x), x[None, :])
    y = x.copy()
    assert np.array_equal(c.allreduce(y, "sum"), x) and np.array_equal(c.allreduce(y, "max"), x)
    c.barrier()
    from aqc_research_amd import job_executor as jex

    recs = [{"job_index": j, "seed": 7 * j, "time": 0.1, "status": "ok", "cost": 0.5, "fidelity": 0.5, "num_iters": 3,
             "num_fun_ev": 3, "num_grad_ev": 3, "thetas": np.arange(4.0) + j} for j in range(3)]
    back = jex._gather_records(recs, c, 3, "fixed")
    assert [r["job_index"] for r in back] == [0, 1, 2] and np.array_equal(back[2]["thetas"], np.arange(4.0) + 2)
    c.close()
