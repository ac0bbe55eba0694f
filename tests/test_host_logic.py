"""CPU-only tests of host logic: job executor (incl. world_size-2 gloo), stoppers, optimizer wrapper."""
import os
import subprocess
import sys
import textwrap
import time

import numpy as np
import pytest

from oracle import aqc_oracle as orc
from tests.helpers import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _job(idx, cfg):
    if cfg.get("fail"):
        raise ValueError("boom")
    return {"value": cfg["a"] * 2 + np.random.rand(), "idx_echo": idx}


def test_run_jobs_serial_contract():
    from aqc_research_amd.job_executor import run_jobs

    cfgs = [{"a": i} for i in range(5)] + [{"a": 0, "fail": True}]
    res = run_jobs(cfgs, 100, _job)
    assert [r["job_index"] for r in res] == list(range(6))
    assert [r["seed"] for r in res] == [100 + 7 * (i + 1) for i in range(6)]
    assert all(r["status"] == "ok" for r in res[:5]) and "ValueError" in res[5]["status"] and res[5]["time"] == -1.0
    np.random.seed(107)
    assert res[0]["value"] == 0 + np.random.rand()  # per-job seeding (job_executor.py:64-65)
    assert len(run_jobs(cfgs, 100, _job, tolerate_failure=True)) == 5
    with pytest.raises(RuntimeError):
        run_jobs([{"a": 1, "fail": True}], 0, _job)
    with pytest.raises(ValueError):
        run_jobs([], 0, _job)


def test_run_jobs_gloo_world2(tmp_path):
    """Two ranks shard 7 jobs, gather the padded records; every rank ends with the same ordered list."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import numpy as np, torch.distributed as dist
        from aqc_research_amd.job_executor import run_jobs
        dist.init_process_group(backend="gloo")
        from tests.gloo_double import install; install(dist)
        def job(i, cfg):
            return {{"value": cfg["a"] + np.random.rand(), "rank": dist.get_rank(), "arr": np.arange(cfg["a"] + 1)}}
        res = run_jobs([{{"a": i}} for i in range(7)], 5, job)
        out = [(r["job_index"], r["seed"], r["rank"], float(r["value"]), int(r["arr"].sum())) for r in res]
        open(os.path.join({str(tmp_path)!r}, f"out{{dist.get_rank()}}.json"), "w").write(json.dumps(out))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    import json

    a = json.loads((tmp_path / "out0.json").read_text())
    b = json.loads((tmp_path / "out1.json").read_text())
    assert a == b and [r[0] for r in a] == list(range(7))
    assert [r[2] for r in a] == [i % 2 for i in range(7)]  # job j ran on rank j % world
    for i, r in enumerate(a):
        np.random.seed(5 + 7 * (i + 1))
        assert r[1] == 5 + 7 * (i + 1) and abs(r[3] - (i + np.random.rand())) < 1e-15 and r[4] == i * (i + 1) // 2


class _Quad:
    """Duck-typed objective: f = |x - c|^2."""

    def __init__(self, c):
        self.c, self.trackers, self.nf = np.asarray(c, float), None, 0

    def objective(self, x):
        self.nf += 1
        return float(np.sum((x - self.c) ** 2))

    def gradient(self, x):
        return 2 * (x - self.c)

    def set_status_trackers(self, timeout=None, stopper=None):
        self.trackers = (timeout, stopper)

    fidelity = property(lambda self: 0.5)
    statistics = property(lambda self: {"n": self.nf})


class _Circ:
    blocks = np.array([[0], [1]])
    entangler = "cx"


def test_optimizer_wrapper_and_stoppers():
    from aqc_research_amd import optimizer as opt

    obj = _Quad([1.0, -2.0, 0.5])
    res = opt.AqcOptimizer(optimizer_name="lbfgs", maxiter=50).optimize(obj, _Circ(), np.zeros(3))
    assert res["cost"] < 1e-10 and np.allclose(res["thetas"], obj.c, atol=1e-5)
    for key in ("num_iters", "num_fun_ev", "num_grad_ev", "ini_thetas", "blocks", "entangler", "stats", "is_timeout", "fidelity"):
        assert key in res
    assert res["is_timeout"] is False and res["stats"]["is_timeout"] is False and np.all(res["ini_thetas"] == 0)
    res = opt.AqcOptimizer(optimizer_name="adam", maxiter=2000, learn_rate=0.05).optimize(_Quad([0.3, 0.1]), _Circ(), np.zeros(2))
    assert res["cost"] < 1e-6

    # stoppers
    with pytest.raises(StopIteration):
        opt.SmallObjectiveStopper(fobj_thr=0.1).check(0.05)
    s = opt.NotImproveStopper(num_iters=2)
    s.check(1.0, 0); s.check(1.0, 1); s.check(1.0, 2)
    with pytest.raises(opt.StagnantOptimizationWarning):
        s.check(1.0, 3)
    es = opt.EarlyStopper(fidelity_thr=0.9)
    with pytest.raises(StopIteration):
        es.check(0.2, 0.95, np.zeros(2), 3, lambda f, t: {"cost": f, "thetas": t})
    assert es.optim_results["cost"] == 0.2
    tc = opt.TimeoutChecker(time_limit=-1)
    tc.check(1.0, np.zeros(1))  # no limit => no raise
    ga = opt.GradientAmplifier(history=3)
    assert ga.estimate(0.5) == 1.0 and ga.estimate(0.5) == 1.0 and ga.estimate(0.5) > 1.0

    # early stop propagates through the wrapper with the stored result
    class StopObj(_Quad):
        def gradient(self, x):
            raise StopIteration("stop")

        optim_results = property(lambda self: {"cost": 0.123, "thetas": np.ones(3)})

    res = opt.AqcOptimizer(maxiter=5).optimize(StopObj([0, 0, 0]), _Circ(), np.zeros(3))
    assert res["cost"] == 0.123 and res["is_timeout"] is False


def test_trotter_initialisation_matches_exact_evolution():
    """init_ansatz_to_trotter reproduces the XXZ evolution: the Trotter state (CPU oracle applies the
    ansatz) converges to exp(-iHt)|neel> with the 2nd-order rate (test_trotter.py:88-96 asks > 0.9)."""
    from scipy.linalg import expm

    from aqc_research_amd.model_sp_lhs.trotter import (init_ansatz_to_trotter, make_hamiltonian, neel_state_index,
                                                       trotter_alphas, trotter_ansatz, trotter_global_phase)
    from oracle import aqc_oracle as orc

    n, t, delta = 5, 1.0, 1.0
    ini = np.zeros(1 << n, complex); ini[neel_state_index(n)] = 1
    exact = expm(-1j * t * make_hamiltonian(n, delta)) @ ini
    errs = []
    for steps in (2, 4, 8):
        circ = trotter_ansatz(n, steps, True)
        th = init_ansatz_to_trotter(circ, np.ones(circ.num_thetas), evol_time=t, delta=delta)
        assert np.count_nonzero(th) <= 3 * (circ.num_blocks // 3) and np.all(th[: 3 * n] == 0)
        v = orc.v_mul_vec(circ, th, ini) * np.exp(1j * trotter_global_phase(n, steps, True))
        errs.append(np.linalg.norm(v - exact))
        assert abs(np.vdot(exact, v)) ** 2 > 0.99
    assert errs[1] < errs[0] / 3 and errs[2] < errs[1] / 3      # ~ dt^2
    a = trotter_alphas(0.3, 1.5)
    assert np.allclose(a, [np.pi / 2 - 0.225, 0.15 - np.pi / 2, np.pi / 2 - 0.15])
    # partial range leaves the other layers alone
    circ = trotter_ansatz(4, 3, False)
    th = init_ansatz_to_trotter(circ, np.full(circ.num_thetas, 7.0), evol_time=1.0, delta=1.0, layer_range=(1, 2))
    v2 = circ.subset2q(th).reshape(3, 3, 12)
    assert np.all(v2[0] == 7.0) and np.all(v2[2] == 7.0) and np.all(th[: 12] == 7.0) and np.count_nonzero(v2[1]) == 9


def test_lockstep_round_logic_with_stub_workspace(monkeypatch):
    """The lockstep coordinator (threads, rounds, call merging, early retirement) checked on the CPU: the
    batched workspace is replaced by a stub whose eval() is the oracle."""
    import threading

    from aqc_research_amd import ParametricCircuit, lockstep
    from aqc_research_amd.engine import BUF_X, BUF_Y

    n = 4
    a = orc.Ansatz(n, "cx", orc.spin_blocks(n, 5))
    circ = ParametricCircuit(n, "cx", a.blocks)
    calls = []

    class StubCtx:
        @staticmethod
        def of(c):
            return StubCtx()

    class StubWorkspace:
        def __init__(self, ctx, batch=1, ncols=1, device=0):
            self.batch, self.T, self.dim, self.ctx, self.device = batch, a.num_thetas, 1 << n, ctx, device
            self.y = np.zeros((batch, 1 << n), complex)
            self.basis = np.zeros(batch, dtype=np.int64)
            self.idx = None

        def upload(self, buf, data, lane=None):
            assert buf == BUF_Y
            self.y[lane] = data

        def set_basis(self, buf, index):
            self.basis = np.array(index)

        def gather_setup(self, idx):
            self.idx = np.array(idx)

        def eval(self, thetas, vdag=True, gather=False, grad=True, x_buf=BUF_X, block_range=None, front_layer=True):
            calls.append(threading.current_thread().name)
            hs = np.zeros((self.batch, len(self.idx)), complex)
            g = np.zeros((self.batch, self.T), complex)
            for b in range(self.batch):
                vh = orc.v_dagger_mul_vec(a, thetas[b], self.y[b])
                hs[b] = vh[self.idx]
                x = np.zeros(1 << n, complex)
                x[self.basis[b]] = 1
                g[b] = orc.grad_of_dot_product(a, thetas[b], x, vh, block_range, front_layer)
            return hs, g

        def close(self):
            pass

    monkeypatch.setattr(lockstep, "Workspace", StubWorkspace)
    monkeypatch.setattr(lockstep, "HipContext", StubCtx)
    rng = np.random.default_rng(8)
    ys = [orc.rand_state(n, rng) for _ in range(3)]
    ths = [[orc.rand_thetas(a.num_thetas, rng) for _ in range(4)] for _ in range(3)]
    batch = lockstep.LockstepBatch(circ, 3)

    def job(view, lane):
        view.upload(BUF_Y, ys[lane])
        view.set_basis(BUF_X, lane)
        view.gather_setup([0, 1, 2])
        out = []
        for r in range(2 + lane):     # lanes retire after 2, 3 and 4 rounds
            hs, g = view.eval(ths[lane][r], vdag=True, gather=True, grad=True)
            out.append((hs[0].copy(), g[0].copy()))
        return out

    res = batch.run([(lambda v, lane=lane: job(v, lane)) for lane in range(3)])
    assert batch.rounds == 4 and len(calls) == 4          # one merged native call per round
    for lane in range(3):
        assert not isinstance(res[lane], BaseException), res[lane]
        for r, (hs, g) in enumerate(res[lane]):
            vh = orc.v_dagger_mul_vec(a, ths[lane][r], ys[lane])
            x = np.zeros(1 << n, complex)
            x[lane] = 1
            assert np.allclose(hs, vh[:3], atol=1e-14) and np.allclose(g, orc.grad_of_dot_product(a, ths[lane][r], x, vh), atol=1e-14)
    with pytest.raises(ValueError):   # lanes must agree on the gathered amplitudes
        batch.lane(0).gather_setup([0, 1])
    # a failing job hands its exception back and does not dead-lock the others
    def bad(view):
        raise KeyError("boom")
    res = batch.run([bad, lambda v: job(v, 1)[0][0][0]])
    assert isinstance(res[0], KeyError) and isinstance(res[1], complex)


def test_lockstep_serves_whole_surrogate_evaluations_with_stub_workspace(monkeypatch):
    """SpSurrogateObjectiveMax objects on lockstep lanes, on the CPU: the batched workspace is a stub whose surrogate_eval()
    is the oracle (amplitudes, hysteresis, combined complex gradient c_0 g_0 + c_max g_max).  Every objective()/gradient()
    pair must be ONE request, a round ONE native call, and every lane must follow orc.SurMaxOracle."""
    from aqc_research_amd import ParametricCircuit, lockstep
    from aqc_research_amd.engine import BUF_Y
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax

    n = 4
    a = orc.Ansatz(n, "cz", orc.spin_blocks(n, 4))
    circ = ParametricCircuit(n, "cz", a.blocks)
    idx = orc.flip_state_indices(n, 1)
    calls = []

    class StubCtx:
        @staticmethod
        def of(c):
            return StubCtx()

    class StubWorkspace:
        def __init__(self, ctx, batch=1, ncols=1, device=0):
            self.batch, self.T, self.dim, self.ctx, self.device = batch, a.num_thetas, 1 << n, ctx, device
            self.y = np.zeros((batch, 1 << n), complex)

        def upload(self, buf, data, lane=None):
            assert buf == BUF_Y
            self.y[lane] = data

        def set_basis(self, buf, index):
            pass

        def gather_setup(self, index):
            assert np.array_equal(index, idx)

        def surrogate_eval(self, thetas, weight, max_no, update_state, block_range, front_layer):
            calls.append(int(update_state))
            B = self.batch
            f, fid = np.zeros(B), np.zeros(B)
            hs, gc = np.zeros((B, idx.size), complex), np.zeros((B, self.T), complex)
            for b in range(B):
                z = orc.v_dagger_mul_vec(a, thetas[b], self.y[b])
                h = z[idx]
                h2 = np.abs(h) ** 2
                if update_state:
                    best = h2[max_no[b]]
                    for i in range(idx.size):
                        if 1.1 * best < h2[i]:
                            best, max_no[b] = h2[i], i
                assert update_state == 2          # the objects ask for the hysteresis only: the weight moves in gradient()
                w, k = weight[b], int(max_no[b])
                f[b], fid[b], hs[b] = 1.0 - (1.0 - w) * h2[0] - w * h2[k], h2[0], h

                def sweep(i):
                    x = np.zeros(1 << n, complex)
                    x[idx[i]] = 1
                    return orc.grad_of_dot_product(a, thetas[b], x, z, block_range, front_layer)

                gc[b] = -2 * np.conj(h[0]) * sweep(0) if k == 0 else (-2 * (1 - w) * np.conj(h[0]) * sweep(0) - 2 * w * np.conj(h[k]) * sweep(k))
            return f, fid, hs, gc

        def close(self):
            pass

    monkeypatch.setattr(lockstep, "Workspace", StubWorkspace)
    monkeypatch.setattr(lockstep, "HipContext", StubCtx)
    monkeypatch.setattr(lockstep.LaneView, "prefers_surrogate_eval", True)
    rng = np.random.default_rng(44)
    data = [(orc.rand_state(n, rng), orc.rand_thetas(a.num_thetas, rng)) for _ in range(3)]
    batch = lockstep.LockstepBatch(circ, 3)

    def job(view, lane):
        y, th = data[lane]
        user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: 0, enable_optim_stats=False, verbose=0, workspace=view)
        obj = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
        obj.set_target(y)
        o = orc.SurMaxOracle(a, y, 1, None, True)
        err = 0.0
        for _ in range(2 + lane):
            f, fo = obj.objective(th), o.objective(th)
            g, go = obj.gradient(th), o.gradient(th)
            err = max(err, abs(f - fo), float(np.abs(g - go).max()), abs(obj._weight - o.weight), float(obj._max_no != o.max_no))
            th = th - 0.1 * g
        return err

    res = batch.run([(lambda v, lane=lane: job(v, lane)) for lane in range(3)])
    for r in res:
        assert not isinstance(r, BaseException), r
        assert r < 1e-12
    assert batch.rounds == 4 and len(calls) == 4 and batch.native_calls == 4   # lanes retire after 2, 3 and 4 pairs


def test_result_record_roundtrip():
    """Fixed-size record of run_jobs' final gather: {job_index, seed, ok, time, cost, fidelity, counts, thetas[T_max]}."""
    from aqc_research_amd import job_executor as jex

    th = np.linspace(-1, 1, 7)
    r = {"job_index": 5, "seed": 40, "time": 0.25, "status": "ok", "cost": 0.125, "fidelity": 0.875, "num_iters": 12,
         "num_fun_ev": 15, "num_grad_ev": 15, "thetas": th}
    back = jex.unpack_record(jex.pack_record(r, 10))
    assert back["job_index"] == 5 and back["seed"] == 40 and back["status"] == "ok" and back["num_iters"] == 12
    assert back["cost"] == 0.125 and back["fidelity"] == 0.875 and np.array_equal(back["thetas"], th)
    failed = jex.unpack_record(jex.pack_record({"job_index": 1, "seed": 2, "time": -1.0, "status": "Traceback ..."}, 3))
    assert failed["job_index"] == 1 and not failed["status"].startswith("ok") and "thetas" not in failed
    assert jex._fits_fixed_schema(r) and not jex._fits_fixed_schema(dict(r, extra=1))


def test_run_jobs_fixed_records_gloo_world2(tmp_path):
    """The shipped gather (fixed-size float64 records, one all-gather) over the gloo test double: 5 jobs on 2 ranks,
    ragged theta sizes, one failing job."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {ROOT!r})
        import numpy as np, torch.distributed as dist
        from aqc_research_amd.job_executor import run_jobs
        from aqc_research_amd import comm
        dist.init_process_group(backend="gloo")
        from tests.gloo_double import install; install(dist)
        def job(i, cfg):
            if cfg["a"] == 3:
                raise ValueError("boom")
            return {{"cost": 0.5 * cfg["a"], "fidelity": 1 - 0.1 * cfg["a"], "num_iters": cfg["a"], "num_fun_ev": 2 * cfg["a"],
                     "num_grad_ev": 2 * cfg["a"], "thetas": np.arange(cfg["a"] + 2, dtype=float) + np.random.rand()}}
        res = run_jobs([{{"a": i}} for i in range(5)], 11, job, records="auto")
        c = comm.from_environment()
        out = [(r["job_index"], r["seed"], r["status"].startswith("ok"), r.get("cost"), r.get("num_iters"),
                None if "thetas" not in r else [float(v) for v in r["thetas"]]) for r in res]
        open(os.path.join({str(tmp_path)!r}, f"out{{dist.get_rank()}}.json"), "w").write(json.dumps([out, c.transport, c.size]))
        dist.destroy_process_group()
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    import json

    a, ta, sa = json.loads((tmp_path / "out0.json").read_text())
    b, tb, sb = json.loads((tmp_path / "out1.json").read_text())
    assert a == b and sa == sb == 2 and "gloo" in ta
    assert [r[0] for r in a] == [0, 1, 2, 3, 4] and [r[2] for r in a] == [True, True, True, False, True]
    for i, r in enumerate(a):
        if i == 3:
            continue
        np.random.seed(11 + 7 * (i + 1))
        assert r[1] == 11 + 7 * (i + 1) and r[3] == 0.5 * i and r[4] == i
        assert np.allclose(r[5], np.arange(i + 2, dtype=float) + np.random.rand(), atol=0, rtol=0)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` started as ONE plain process must become N ranks (job_executor.py:136-143 fans out by
    itself too): the launcher spawns them with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / AQC_COMM_FILE set, relays rank 0's
    single JSON line and propagates failures.  --rank-echo keeps the ranks off the GPU (a gloo group on the CPU)."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rank-echo"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["world_size"] == 3 and out["ranks_seen"] == [0, 1, 2] and out["local_ranks"] == [0, 1, 2] and out["comm_files"] == 1
    # a rank that fails takes the launch down with a non-zero exit code (the others are terminated, nothing hangs)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-echo"], capture_output=True, text=True,
                       env=dict(env, AQC_BENCH_ECHO_FAIL_RANK="1"), timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_run_jobs_fails_instead_of_hanging_when_a_rank_dies(tmp_path):
    """A rank that dies before the final gather must fail the job list on the survivors (here: the gloo double's collective
    raises after its time-out), not leave them waiting forever (job_executor.py:149-159 reports failures, it never hangs)."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys, datetime
        sys.path.insert(0, {ROOT!r})
        import numpy as np, torch.distributed as dist
        from aqc_research_amd.job_executor import run_jobs
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=8))
        from tests.gloo_double import install; install(dist)
        def job(i, cfg):
            if dist.get_rank() == 1:
                os._exit(17)          # the rank dies inside its first job
            return {{"cost": 0.0, "fidelity": 1.0, "num_iters": 1, "num_fun_ev": 1, "num_grad_ev": 1, "thetas": np.zeros(3)}}
        try:
            run_jobs([{{"a": i}} for i in range(4)], 3, job, records="fixed")
        except Exception as ex:
            open(os.path.join({str(tmp_path)!r}, "failed.txt"), "w").write(type(ex).__name__)
            os._exit(9)
        os._exit(0)
    """))
    import time as _time

    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    t0 = _time.time()
    codes = [q.wait(timeout=120) for q in procs]
    assert codes[1] == 17 and codes[0] == 9, codes          # the survivor failed loudly ...
    assert _time.time() - t0 < 100 and (tmp_path / "failed.txt").exists()   # ... within the collective's time-out


def test_dense_backed_mps_is_a_lazy_canonical_qiskit_tuple():
    """What v_dagger_mul_mps hands back on the dense route (mps_operations.py:349-371 returns Aer's canonical MPS): a tuple of
    length 2 whose Vidal-form tensors are computed on first access, canonical, exact at the no-truncation threshold, smaller
    under a real one, and pickled as a plain tuple."""
    import pickle

    from aqc_research_amd.mps_engine import is_canonical
    from aqc_research_amd.mps_operations import DenseBackedMPS, check_mps, vector_to_canonical_mps

    rng = np.random.default_rng(31)
    v = orc.rand_state(7, rng)
    d = DenseBackedMPS(v, 1e-16)
    assert isinstance(d, tuple) and len(d) == 2 and check_mps(d) and d._mps is None        # nothing computed yet
    gam, lam = d
    assert d._mps is not None and len(gam) == 7 and len(lam) == 6 and check_mps((gam, lam)) and is_canonical((gam, lam))
    assert all(np.all(np.diff(l) <= 1e-15) for l in lam)
    assert np.abs(orc.mps_to_vector((gam, lam)) - v).max() < 1e-13
    assert np.abs(orc.mps_to_vector(d) - v).max() < 1e-13 and d[1][0].size == 2
    back = pickle.loads(pickle.dumps(d))
    assert type(back) is tuple and np.abs(orc.mps_to_vector(back) - v).max() < 1e-13
    cut = vector_to_canonical_mps(v, 1e-2)
    w = orc.mps_to_vector(cut)
    assert max(l.size for l in cut[1]) < max(l.size for l in lam)
    assert 0 < 1 - abs(np.vdot(w, v)) ** 2 / np.vdot(w, w).real < 5e-2
    assert not is_canonical(orc.random_mps(7, 4, rng))        # hand-made Vidal-form tensors are not canonical
    assert not d.dense_on(object(), 1)                        # no workspace behind this one


def test_basis_mask_of_duck_typed_preparation_circuits():
    """state_prep_func may return a circuit (trotter.py:381-410, consumed at objective_base.py:298-303): X-only circuits are
    turned into the bit mask of the basis state they prepare -- Qiskit-style instruction objects and old-style triples alike --
    anything else raises."""
    from aqc_research_amd.model_sp_lhs.objective_base import basis_mask_of_circuit
    from aqc_research_amd.model_sp_lhs.trotter import (half_zero_circuit, identity_circuit, neel_init_state,
                                                        neel_state_index)

    assert basis_mask_of_circuit(neel_init_state(7), 7) == neel_state_index(7) == 0b1010101
    assert basis_mask_of_circuit(identity_circuit(5), 5) == 0
    assert basis_mask_of_circuit(half_zero_circuit(6), 6) == 0b111000
    assert neel_init_state(6).basis_index == 0b010101

    class Bit:          # a Qiskit-like qubit object: resolved through circuit.find_bit(q).index
        def __init__(self, i):
            self._i = i

    class Loc:
        def __init__(self, i):
            self.index = i

    class Op:
        def __init__(self, name):
            self.name = name

    class Circ:
        num_qubits = 4

        def __init__(self, data):
            self.data = data

        def find_bit(self, q):
            return Loc(q._i)

    qs = [Bit(i) for i in range(4)]
    old_style = Circ([(Op("x"), [qs[1]], []), (Op("barrier"), qs, []), (Op("id"), [qs[0]], []), (Op("x"), [qs[3]], []), (Op("x"), [qs[3]], [])])
    assert basis_mask_of_circuit(old_style, 4) == 0b0010
    with pytest.raises(NotImplementedError):
        basis_mask_of_circuit(Circ([(Op("h"), [qs[0]], [])]), 4)
    with pytest.raises(NotImplementedError):
        basis_mask_of_circuit(Circ([(Op("cx"), [qs[0], qs[1]], [])]), 4)
    with pytest.raises(ValueError):
        basis_mask_of_circuit(Circ([]), 5)


def test_horizon_driver_options_follow_the_reference_defaults():
    """user_options.py:25-129: 6 horizons of 1.2 with 3 Trotter steps each, trunc_thr 1e-6, fidelity_thr 0.995, maxiter 40,
    2 layers per horizon, the Neel preparation; the threshold rule of time_evol_best_init.py:118-140."""
    from aqc_research_amd.model_sp_lhs import time_evol as te

    o = te.UserOptions()
    assert list(o.trotter_steps) == [3, 6, 9, 12, 15, 18] and list(o.evol_times) == [1.2, 2.4, 3.6, 4.8, 6.0, 7.2]
    assert (o.trunc_thr, o.fidelity_thr, o.maxiter, o.num_layers_inc, o.delta) == (1e-6, 0.995, 40, 2, 1.0)
    assert o.second_order_trotter and o.enable_grad_scaling and o.num_expansions == 0 and te.precise_multiplier() == 10
    assert not o.use_mps and te.UserOptions(objective="sur_fast_mps_trotter").use_mps
    assert te._initial_layers(o, 2) == 6 and te._initial_layers(te.UserOptions(manual_num_layers=[2, 4, 5]), 2) == 5
    v = np.zeros(4, complex); v[0] = 1
    w = np.array([np.sqrt(0.9), np.sqrt(0.1), 0, 0], complex)
    t = te.TargetState(num_qubits=2, num_trot_steps=3, evol_time=1.2, my_id=0, delta=1.0, second_order=True, t1_gt=v, t1=w)
    assert te._calc_fidelity_threshold(t, 0.995) == (0.995, pytest.approx(0.9))
    assert te._calc_fidelity_threshold(t, 0.5)[0] == pytest.approx(0.9)
    assert te._calc_fidelity_threshold(t, None)[0] == pytest.approx(1.03 * 0.9)
    with pytest.raises(ValueError):
        te.UserOptions(trotter_steps=[3, 6], evol_times=[1.2])


def test_rccl_id_file_of_another_launch_is_not_accepted(tmp_path):
    """A rank other than 0 must only take a unique id that carries ITS launch's tag: an id file left behind by a crashed
    launch with the same name (it used to be accepted when younger than 600 s and made ncclCommInitRank hang) times out
    with a clear error instead -- before any RCCL or HIP call is made."""
    from aqc_research_amd import comm

    f = tmp_path / "rccl_unique_id"
    f.write_bytes(b"\x01" * 128 + b"tag_of_a_crashed_launch")
    t0 = time.time()
    with pytest.raises(RuntimeError, match="timed out waiting for the unique id of launch 'this_launch'"):
        comm.RcclCommunicator(1, 2, 0, str(f), timeout=0.4, tag="this_launch")
    assert 0.3 < time.time() - t0 < 5
    f.write_bytes(b"\x01" * 128)                       # the tag-less format of earlier rounds is not accepted either
    with pytest.raises(RuntimeError, match="timed out"):
        comm.RcclCommunicator(1, 2, 0, str(f), timeout=0.2, tag="this_launch")


def test_early_stopper_rules_in_the_reference_order():
    """optimizer.py:228-336: objective threshold first, then stagnation -- which reports the BEST point seen, not the current
    one -- then the fidelity threshold; optimizer.py:36-65,158-225: limits <= 0 never expire, a passed limit raises TimeoutError
    after handing the current point to ``on_stop``."""
    from aqc_research_amd import optimizer as opt

    seen = []

    def on_stop(f, t):
        seen.append((f, t.copy()))
        return {"cost": f, "thetas": t.copy()}

    es = opt.EarlyStopper(num_iters=2)
    es.check(0.5, None, np.array([1.0, 1.0]), 0, on_stop)
    es.check(0.3, None, np.array([2.0, 2.0]), 1, on_stop)       # the best point
    es.check(0.4, None, np.array([3.0, 3.0]), 2, on_stop)
    es.check(0.35, None, np.array([4.0, 4.0]), 3, on_stop)
    with pytest.raises(StopIteration, match="no improvement"):
        es.check(0.45, None, np.array([5.0, 5.0]), 4, on_stop)
    assert es.optim_results["cost"] == 0.3 and np.all(es.optim_results["thetas"] == 2.0)
    es = opt.EarlyStopper(fobj_thr=0.1, fidelity_thr=0.5)
    with pytest.raises(StopIteration, match="fell below"):      # both rules fire: the objective threshold is tried first
        es.check(0.05, 0.9, np.zeros(2), 0, on_stop)
    with pytest.raises(ValueError):
        opt.EarlyStopper(fidelity_thr=1.5)
    opt.TimeoutStopper(time_limit=-1).check()
    tc = opt.TimeoutChecker(time_limit={"timeout": 1}, start_immediately=False)
    tc.check(1.0, np.zeros(1))                                   # not started: never expires
    tc._deadline._at = 0                                         # (a deadline in the past)
    with pytest.raises(TimeoutError):
        tc.check(0.7, np.ones(1), on_stop)
    assert tc.optim_results["cost"] == 0.7
