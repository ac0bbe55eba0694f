#!/usr/bin/env python3
"""
bench.py -- objective+gradient evaluations/sec of the HIP fidelity/gradient path.

One *step* = one full objective+gradient evaluation for each of the `batch`
independent (theta, target) lanes resident on the GPU:
    coefficient kernel (sin/cos of the lane's thetas)
  + V^H |target>                      (objective: hs = <state_i|V^H|target>)
  + gather of the n+1 flip-state amplitudes
  + forward w/z sweep with all inner products + fixed-order finalize  (gradient)
Thetas change every step (bank of K+W parameter sets resident in HBM), so nothing is
served from a cache.  Default workload = BASELINE.json's headline: 16 qubits, 40 blocks
(cx, spin layout, T = 208), state-vector path.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload NAME]
For N > 1 launch with torch.distributed.run (one rank per GPU); ranks process their own
lanes (weak scaling) and only the final result records are gathered (RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # fp64 vector peak = fp64 matrix (MFMA) peak on gfx950
PARITY_TOL = 1e-10       # north-star tolerance (complex fp64, absolute)

WORKLOADS = {
    # name: (n, layout, L, kind)
    "sv16_l40": dict(n=16, blocks=40, kind="generic", desc="16-qubit, 40-block cx spin ansatz, state-vector objective+gradient"),
    "sv12_trotter2": dict(n=12, layers=2, kind="trotter2", desc="12-qubit ASP, 2nd-order Trotter ansatz (2 layers), state-vector objective+gradient"),
    "sv20_l40": dict(n=20, blocks=40, kind="generic", desc="20-qubit, 40-block cx spin ansatz, state-vector objective+gradient"),
    "sv20_trotter2": dict(n=20, layers=2, kind="trotter2", desc="20-qubit ASP, 2nd-order Trotter ansatz (2 layers), state-vector objective+gradient"),
    "mat10_l40": dict(n=10, blocks=40, kind="generic", ncols=1024, desc="10-qubit full-unitary AQC (1024x1024 target), cx spin ansatz L=40, matrix objective+gradient"),
    # sketched AQC (aqc_sketching, docs/aqc.ipynb: 16 sketching vectors): the matrix path on k < d columns -- X = the sketching vectors,
    # Y = U X (sk_core.py:167-194), here random orthonormal d x 16 blocks
    "mat10_l40_k16": dict(n=10, blocks=40, kind="generic", ncols=16, desc="10-qubit sketched AQC (16 sketching vectors: 1024x16 matrices), cx spin ansatz L=40, matrix objective+gradient"),
    "mat5_cyc180": dict(n=5, blocks=180, kind="cyclic", ncols=32, desc="5-qubit full AQC (docs/aqc.ipynb ansatz: cyclic_spin, 180 blocks), matrix objective+gradient"),
    # config 3 through the MPS front door: every step re-uploads each lane's target as a QiskitMPS (host tensors),
    # contracts it to the dense state on the device (mps_to_vector chain) and runs V^H + gather + sweep on it
    # config 4: the ASP job mix -- 64 seeds x 8 time horizons of a 20-qubit 2nd-order Trotter ansatz (2h layers at horizon
    # h), 10 objective+gradient pairs per job -- sharded over the ranks by run_jobs (job j on rank j % world), results
    # gathered as fixed-size records.  One step = the whole mix.
    "cfg4_jobs": dict(n=20, kind="jobs", seeds=64, horizons=8, evals=10, desc="20-qubit ASP job mix: 64 seeds x 8 horizons (2nd-order Trotter ansatz, 2h layers), 10 objective+gradient pairs per job, sharded by run_jobs"),
    # config 4 as the reference's driver runs it (time_evol_best_init.py:221-382, job_executor.py:96-161): per horizon the
    # ground-truth / reference Trotter targets synthesised on the device, 64 random restarts of the 2h-layer ansatz optimised by
    # L-BFGS under the fidelity threshold, the reference's result record; horizons are the jobs of run_jobs.  One step = the run.
    "cfg4_driver": dict(n=20, kind="driver", seeds=64, horizons=8, maxiter=40, desc="20-qubit ASP run of the horizon driver: time_evol.run_simulation(UserOptions(num_qubits=20, num_horizons=8, num_seeds=64, objective='sur_max', vectorised_lbfgs=True)) -- Trotter targets synthesised on the device, 64 restarts per horizon as lanes of one batched surrogate objective, L-BFGS (maxiter 40) under the fidelity threshold, horizons sharded by run_jobs"),
    # config 2 as the reference's launcher runs it (run_time_evol.py defaults = user_options.py:25-129): 12 qubits, six horizons of
    # 2 .. 12 ansatz layers, ONE optimisation per horizon from the Trotter point (scipy L-BFGS under AqcOptimizer, maxiter 40)
    "cfg2_driver": dict(n=12, kind="driver", seeds=1, horizons=6, maxiter=40, desc="12-qubit ASP run of the horizon driver at the reference's defaults: time_evol.run_simulation(UserOptions(num_qubits=12, objective='sur_max')) -- six horizons (2..12 layers), one L-BFGS optimisation each (AqcOptimizer on SpSurrogateObjectiveMax, maxiter 40) under the fidelity threshold, Trotter targets synthesised on the device"),
    "mps16_l40_chi16": dict(n=16, blocks=40, kind="generic", chi=16, desc="16-qubit, 40-block cx spin ansatz, MPS-dot objective+gradient: QiskitMPS targets chi=16 (a different one per lane every step), contracted to dense on the device every evaluation"),
    "mps16_l40_chi64": dict(n=16, blocks=40, kind="generic", chi=64, desc="16-qubit, 40-block cx spin ansatz, MPS-dot objective+gradient: QiskitMPS targets chi=64 (a different one per lane every step), contracted to dense on the device every evaluation"),
    # the same front door at the threshold the reference's own driver hands over (user_options.py:55: trunc_thr = 1e-6): the targets
    # are canonical tensors truncated at 1e-6 and the calls take the dense route whatever the threshold (mps_dot_objective.use_dense)
    "mps16_l40_chi64_thr1e-6": dict(n=16, blocks=40, kind="generic", chi=64, trunc_thr=1e-6, desc="16-qubit, 40-block cx spin ansatz, MPS-dot objective+gradient at the reference's default trunc_thr = 1e-6: canonical QiskitMPS targets chi<=64 truncated at 1e-6 (a different one per lane every step), dense route"),
    "sv12_trotter12": dict(n=12, layers=12, kind="trotter2", desc="12-qubit ASP, 2nd-order Trotter ansatz (12 layers: the last horizon of run_time_evol.py's defaults), state-vector objective+gradient"),
    # coordinate descent (core_op_matrix.coord_descent_single_sweep, docs/aqc.ipynb: 1000 sweeps of the 5-qubit cyclic_spin ansatz):
    # one step = one Gauss-Seidel sweep over all 735 parameters for every lane (lane = random restart with its own target)
    "cd5_cyc180": dict(n=5, blocks=180, kind="cd", ncols=32, desc="5-qubit coordinate descent (docs/aqc.ipynb ansatz: cyclic_spin, 180 blocks, 735 parameters): one coord_descent_single_sweep per lane and step, lanes = random restarts with their own target unitary"),
    # beyond dense reach: the native MPS engine (truncated two-site SVDs on the device, no 2^n buffer anywhere), lanes on host threads
    "mps32_trotter2_opt": dict(n=32, layers=2, kind="mps_opt", trunc_thr=1e-6, lanes=64, maxiter=12, desc="32-qubit ASP horizon as the reference's driver runs it (time_evol_best_init.py:221-334 with objective sur_fast_mps_trotter): 64 restarts of a 2-layer 2nd-order Trotter ansatz (840 parameters) against a 6-layer Trotter target, surrogate objective on the native MPS engine at trunc_thr = 1e-6, all restarts optimised together by one vectorised L-BFGS"),
    "mps32_trotter2_engine": dict(n=32, layers=2, kind="mps_engine", trunc_thr=1e-6, lanes=1024, desc="32-qubit ASP, 2nd-order Trotter ansatz (2 layers, 840 parameters), MPS-dot objective+gradient on the native MPS engine at the reference's default trunc_thr = 1e-6 (V^H by truncated two-site SVDs, gate-by-gate gradient), targets = 6-layer Trotter states (bond <= 16)"),
    "mps16_l40_chi256": dict(n=16, blocks=40, kind="generic", chi=256, desc="16-qubit, 40-block cx spin ansatz, MPS-dot objective+gradient: QiskitMPS targets chi=256 (a different one per lane every step), contracted to dense on the device every evaluation"),
}


def build_circuit(w):
    from aqc_research_amd import ParametricCircuit, TrotterAnsatz
    from aqc_research_amd.circuit_structures import create_ansatz_structure, make_trotter_like_circuit

    if w["kind"] == "generic":
        return ParametricCircuit(w["n"], "cx", create_ansatz_structure(w["n"], "spin", "full", w["blocks"]))
    if w["kind"] == "cyclic":
        return ParametricCircuit(w["n"], "cx", create_ansatz_structure(w["n"], "cyclic_spin", "full", w["blocks"]))
    return TrotterAnsatz(w["n"], make_trotter_like_circuit(w["n"], w["layers"]), second_order=True)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one, else the affinity
    mask capped at the GPU box's per-GPU CPU share (16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def all_host_cores():
    """(logical CPUs, physical cores) this process may run on: the affinity mask, cut by the cgroup CPU quota when there is
    one; physical = distinct (package, core) pairs of /proc/cpuinfo inside the mask."""
    mask = os.sched_getaffinity(0)
    logical = len(mask)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            logical = min(logical, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    phys = set()
    try:
        cpu = pkg = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu, pkg, core = int(line.split(":")[1]), None, None
            elif line.startswith("physical id"):
                pkg = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip() and cpu is not None:
                if cpu in mask:
                    phys.add((pkg, core if core is not None else cpu))
                cpu = None
    except (OSError, ValueError):
        phys = set()
    physical = min(len(phys) or logical, logical)
    return logical, physical


def cpu_baseline(circ, ncols=1, seconds=8.0):
    """The reference algorithm (one pass per gate, one per inner product) timed on the host cores of
    this box on a bounded sample of the same workload: `value` is the compiled C restatement
    (oracle/aqc_ref.c) with one evaluation per core at a time -- the way the reference uses cores
    (job_executor.py:141) -- and the single-thread C and NumPy (oracle/aqc_oracle.py) rates ride along."""
    from oracle import aqc_oracle as orc
    from oracle import aqc_ref as cref

    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        limiter = None
    rng = np.random.default_rng(7)
    a = orc.as_ansatz(circ)
    cores = host_cores()

    def timed(fn, budget, cap=10000):
        fn()  # warm-up
        t0, count = time.perf_counter(), 0
        while count < 1 or (time.perf_counter() - t0 < budget and count < cap):
            count += fn()
        return count, time.perf_counter() - t0

    if ncols == 1:
        target = orc.rand_state(a.n, rng)
        x = np.zeros(a.dim, complex)
        x[0] = 1

        def numpy_one():
            th = orc.rand_thetas(a.num_thetas, rng)
            orc.grad_of_dot_product(a, th, x, orc.v_dagger_mul_vec(a, th, target))
            return 1

        def c_batch(threads):
            def run():
                th = np.stack([orc.rand_thetas(a.num_thetas, rng) for _ in range(threads)])
                cref.eval_batch(a, th, target, 0, threads)
                return threads
            return run

        logical, physical = all_host_cores()
        share = cores                                   # the per-GPU CPU share of the box (16)
        cores = max(physical, share)                    # the headline figure: one evaluation per physical core the process may use
        n_all, t_all = timed(c_batch(cores), seconds)
        n_share, t_share = (n_all, t_all) if share == cores else timed(c_batch(share), seconds / 2)
        n_one, t_one = timed(c_batch(1), seconds / 4)
    else:
        u = np.linalg.qr(rng.standard_normal((a.dim, ncols)) + 1j * rng.standard_normal((a.dim, ncols)))[0]
        eye = np.eye(a.dim, ncols, dtype=complex)

        def numpy_one():
            orc.sketching_objective_and_gradient(a, orc.rand_thetas(a.num_thetas, rng), eye, u)
            return 1

        def c_one():
            th = orc.rand_thetas(a.num_thetas, rng)
            cref.grad_of_matrix_dot_product(a, th, eye, cref.v_dagger_mul_mat(a, th, u))
            return 1

        cores = share = 1
        logical, physical = all_host_cores()
        n_all, t_all = timed(c_one, seconds)
        n_one, t_one = n_all, t_all
        n_share, t_share = n_all, t_all
    n_np, t_np = timed(numpy_one, seconds / 2, cap=200)
    if limiter is not None:
        limiter.restore_original_limits()
    return {
        "value": n_all / t_all,
        "unit": "evals/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_all} objective+gradient evaluations of the same ansatz in {t_all:.1f} s: C restatement of the "
                  f"reference algorithm, {cores} thread(s), one evaluation per thread",
        "per_gpu_share_16": {"value": n_share / t_share, "cores": share,
                             "note": "the same on the box's per-GPU CPU share (what an 8-GPU node leaves each rank)"},
        "affinity": {"logical_cpus": logical, "physical_cores": physical},
        "c_1thread_evals_per_s": n_one / t_one,
        "numpy_1thread_evals_per_s": n_np / t_np,
        "host": {"cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(), "blas_threads": 1},
    }


def reference_numpy_record(workload):
    """The reference's own NumPy path on this workload, as tools/ref_baseline.py measured it in the BUILD container (the
    reference never travels to the GPU box): read from the committed profiles/ref_baseline.json, labelled with its host."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "ref_baseline.json")))
        row = rec["configs"][workload]
    except Exception:
        return None
    return {"evals_per_s_1_process": row["single_process"]["rate"], "median_ms_1_process": row["single_process"]["median_ms"],
            "p10_ms": row["single_process"]["p10_ms"], "p90_ms": row["single_process"]["p90_ms"], "reps": row["single_process"]["reps"],
            "evals_per_s_parallel_processes": row["parallel_processes"]["rate"], "processes": row["parallel_processes"]["processes"],
            "measured": f"in the build container on {rec['host']['cpu_model']} ({rec['host']['os_cpu_count']} vCPU), {rec['date']}, "
                        "tools/ref_baseline.py; NOT on this GPU box"}


# ---- config 4: the ASP job mix ---------------------------------------------------------------------------------------
# What a horizon's jobs share stays resident between them (time_evol_best_init.py:337-382 builds everything per job): the
# ansatz context and its plans (HipContext cache), ONE batched objective per (horizon, lanes) whose lanes are the seeds of a
# job, and the synthetic targets, which live in a target bank on the device and reach an objective's lanes by
# device-to-device copies (in a real run they are synthesised on the device, model_sp_lhs/trotter).
_MIX = {"setup": {}, "pool": {}, "bank": {}, "objective": {}}


def _mix_setup(n, h):
    """(ansatz, Trotter-initialised parameters, Neel index) of horizon h: 2nd-order Trotter ansatz with 2h layers."""
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index, trotter_ansatz

    if (n, h) not in _MIX["setup"]:
        circ = trotter_ansatz(n, 2 * h, True)
        base = init_ansatz_to_trotter(circ, np.zeros(circ.num_thetas), evol_time=1.2 * h, delta=1.0)
        _MIX["setup"][(n, h)] = (circ, base, neel_state_index(n))
    return _MIX["setup"][(n, h)]


def _mix_pool(n):
    """8 synthetic targets, utils.rand_state(n) (uniform[0,1) + i uniform[0,1), normalised; utils.py:71-79), seed 0x696969."""
    from oracle import aqc_oracle as orc

    if n not in _MIX["pool"]:
        prng = np.random.default_rng(0x696969)
        _MIX["pool"][n] = [orc.rand_state(n, prng) for _ in range(8)]
    return _MIX["pool"][n]


def _mix_target(n, seed):
    return _mix_pool(n)[seed % 8]


def _mix_start(base, seed):
    return base + 0.1 * np.pi * (2.0 * np.random.default_rng(seed).random(base.size) - 1.0)


def _mix_job(job_index, cfg):
    """One entry of the config-4 mix: the seeds of the entry are the lanes of one batched objective of horizon h; `evals`
    objective+gradient pairs with a fixed-step descent between them (time_evol_best_init.py:197-208's inner loop with the
    optimizer's line search left out: the workload is the evaluations)."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
    from aqc_research_amd.engine import BUF_Y, HipContext, Workspace

    n, h, seeds, device = cfg["n"], cfg["horizon"], cfg["seeds"], cfg["device"]
    circ, base, neel = _mix_setup(n, h)
    if (n, device) not in _MIX["bank"]:   # the target bank: a workspace of the shallowest ansatz whose Y lanes hold the pool
        bank = Workspace(HipContext.of(_mix_setup(n, 1)[0]), batch=8, device=device)
        bank.upload(BUF_Y, np.stack(_mix_pool(n)))
        _MIX["bank"][(n, device)] = bank
    bank = _MIX["bank"][(n, device)]
    key = (n, h, len(seeds), device)
    if key not in _MIX["objective"]:
        _MIX["objective"][key] = BatchedSurrogateObjective(circ, None, lanes=len(seeds), base_index=neel, device=device)
    bo = _MIX["objective"][key]
    bo.reset_state()
    for lane, sd in enumerate(seeds):
        bo.target_from(lane, bank, BUF_Y, sd % 8)
    th = np.stack([_mix_start(base, sd) for sd in seeds])
    f = None
    for _ in range(cfg["evals"]):
        f, g = bo.value_and_grad(th)
        th = th - 0.05 * g
    return {"cost": float(np.mean(f)), "fidelity": float(np.mean(bo.fidelity)), "num_iters": cfg["evals"],
            "num_fun_ev": cfg["evals"] * len(seeds), "num_grad_ev": cfg["evals"] * len(seeds), "thetas": th[0]}


def _mix_release():
    for kind in ("objective", "bank"):
        for o in _MIX[kind].values():
            o.close()
        _MIX[kind].clear()


def run_job_mix(args, w, comm, rank, local_rank, n_gpus, comm_note, steps=None):
    """--workload cfg4_jobs: the whole mix through run_jobs (rank-sharded, fixed-size record gather)."""
    from aqc_research_amd.job_executor import run_jobs

    # one entry = (horizon, a chunk of the 64 seeds); entry index = horizon * chunks + chunk and run_jobs puts entry j on
    # rank j % world, so with chunks = world every rank works on every horizon (the horizons differ 8x in depth)
    nchunks = max(1, comm.size)
    while w["seeds"] % nchunks:
        nchunks += 1
    chunk = w["seeds"] // nchunks
    configs = [{"n": w["n"], "horizon": h, "evals": w["evals"], "device": local_rank,
                "seeds": [0x696969 + 7 * (c * chunk + s + 1) + 1000 * h for s in range(chunk)]}
               for h in range(1, w["horizons"] + 1) for c in range(nchunks)]
    K, W = max(1, min(args.steps if steps is None else steps, 3)), min(args.warmup, 1)
    for _ in range(W):   # warm-up: contexts, plans, objectives and first launches of every horizon (2 evaluation pairs each)
        run_jobs([dict(c, evals=2) for c in configs], 1, _mix_job, records="fixed")
    comm.barrier()
    t0 = time.perf_counter()
    results = None
    for _ in range(K):
        results = run_jobs(configs, 1, _mix_job, records="fixed")
    comm.barrier()
    wall = time.perf_counter() - t0
    if comm.size > 1:
        wall = float(comm.allreduce(np.array([wall]), "max")[0])
    _mix_release()
    ok = [r for r in results if r["status"].startswith("ok")]
    njobs = w["seeds"] * w["horizons"]
    evals = njobs * w["evals"]
    return {
        "metric": "objective+gradient evals/sec", "value": evals * K / wall, "unit": "evals/s", "n_gpus": n_gpus, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": w["desc"], "n_qubits": w["n"], "jobs": njobs, "jobs_per_s": njobs * K / wall, "evals_per_job": w["evals"],
                   "records_ok": len(ok), "records_total": len(results), "mean_fidelity": float(np.mean([r["fidelity"] for r in ok])),
                   "transport": comm.transport if comm_note is None else comm_note,
                   "lanes_per_entry": chunk, "entries": len(configs),
                   "sharding": "run_jobs: entry j = (horizon j // chunks, seed chunk j % chunks) on rank j % world; chunks = world, so "
                               "every rank holds every horizon; the seeds of an entry are the lanes of one batched objective",
                   "evaluation": "V^H + flip-state amplitudes + ONE sweep from the combined lhs state per lane (aqc_ws_set_combo)"},
    }


def run_driver(args, w, env, full):
    """--workload cfg4_driver: the horizon driver itself (model_sp_lhs/time_evol.run_simulation), horizons sharded by run_jobs.
    Short form (configs of the default line): 2 horizons.  After the run one restart is replayed against the C oracle: the
    objective's amplitude and gradient at the optimised thetas of horizon 1, recomputed on the device AND by the oracle."""
    from aqc_research_amd.engine import BUF_X, BUF_Y, HipContext, Workspace
    from aqc_research_amd.model_sp_lhs import time_evol
    from aqc_research_amd.model_sp_lhs.trotter import trotter_ansatz
    from oracle import aqc_ref as cref

    comm, rank, local_rank, n_gpus = env.comm, env.rank, env.local_rank, env.n_gpus
    H = w["horizons"] if full or w["seeds"] == 1 else 2
    mode = os.environ.get("AQC_BENCH_DRIVER_LBFGS", "vectorised")   # vectorised (host L-BFGS over (B, T) arrays) | device (aqc_ws_lbfgs)
    opts = time_evol.UserOptions(num_qubits=w["n"], num_horizons=H, num_seeds=w["seeds"], objective="sur_max", vectorised_lbfgs=True,
                                 device_lbfgs=(mode == "device"), maxiter=w["maxiter"], device=local_rank, seed=0x696969)
    warm = time_evol.UserOptions(num_qubits=w["n"], num_horizons=1, num_seeds=w["seeds"], objective="sur_max", vectorised_lbfgs=True,
                                 device_lbfgs=(mode == "device"), maxiter=2, device=local_rank, seed=1)
    time_evol.run_simulation(warm)     # library load, first launches (untimed)
    comm.barrier()
    t0 = time.perf_counter()
    recs = time_evol.run_simulation(opts)
    comm.barrier()
    wall = time.perf_counter() - t0
    if comm.size > 1:
        wall = float(comm.allreduce(np.array([wall]), "max")[0])
    evals = int(sum(r["num_fun_ev"] for r in recs))
    # parity replay of one restart (rank 0): horizon 1, the best restart's thetas
    parity = None
    if rank == 0:
        r1 = min(recs, key=lambda r: r["horizon"])
        circ = trotter_ansatz(w["n"], int(r1["num_layers"]), True)
        tgt = time_evol.generate_target(opts, int(r1["horizon"]) - 1).t1_gt
        th = np.asarray(r1["thetas"], dtype=np.float64)
        ini = opts.ini_state_index()
        ws = Workspace(HipContext.of(circ), batch=1, device=local_rank)
        ws.upload(BUF_Y, tgt)
        ws.set_basis(BUF_X, ini)
        ws.gather_setup([ini])
        hs, g = ws.eval(th, vdag=True, gather=True, grad=True)
        ws.close()
        h_ref, g_ref = cref.eval_batch(circ, th[None, :], tgt, ini, min(16, host_cores()))
        parity = max(abs(hs[0, 0] - h_ref[0]), float(np.abs(g[0] - g_ref[0]).max()))
        fid_gap = abs(abs(h_ref[0]) ** 2 - float(r1["fidelity"]))   # the record's fidelity against the oracle's at the record's thetas
        if not parity < PARITY_TOL:
            print(f"bench.py: cfg4_driver: the replayed restart deviates from the oracle by {parity:g}", file=sys.stderr)
            os._exit(3)
    return {
        "metric": "objective+gradient evals/sec", "value": evals / wall, "unit": "evals/s", "n_gpus": n_gpus, "steps": 1, "warmup": 1,
        "ms_per_step": wall * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "parity_maxerr": parity, "parity_lanes_checked": 1, "record_fidelity_vs_oracle": fid_gap if rank == 0 else None,
        "parity_note": "horizon 1, best restart: amplitude <ini|V^H|t1_gt>, complex gradient and the record's fidelity at the optimised "
                       "thetas, device vs C oracle (target = the driver's own ground-truth state)",
        "config": {"workload": w["desc"], "n_qubits": w["n"], "horizons": H, "restarts_per_horizon": w["seeds"], "lbfgs": mode if w["seeds"] > 1 else "scipy L-BFGS-B under AqcOptimizer (one objective object per horizon)",
                   "lbfgs_maxiter": w["maxiter"], "horizons_per_s": H / wall, "optimisations_per_s": H * w["seeds"] / wall,
                   "evaluations": evals, "lanes_per_entry": w["seeds"],
                   "fidelity_best_per_horizon": [float(r["fidelity"]) for r in sorted(recs, key=lambda r: r["horizon"])],
                   "fidelity_threshold_per_horizon": [float(r["fidelity_thr"]) for r in sorted(recs, key=lambda r: r["horizon"])],
                   "fidelity_min_over_restarts": [float(min(r.get("fidelities", [r["fidelity"]]))) for r in sorted(recs, key=lambda r: r["horizon"])],
                   "num_thetas_per_horizon": [int(r["num_thetas"]) for r in sorted(recs, key=lambda r: r["horizon"])],
                   "transport": comm.transport if env.comm_note is None else env.comm_note, "ranks_seen": env.ranks_seen,
                   "includes": "target synthesis (ground truth with 10x the Trotter steps + reference state) per horizon, initial "
                               "points, optimisation of all restarts, result records; contexts and plans are built inside the timed run"},
    }


def run_cd(args, w, env, full):
    """--workload cd5_cyc180: coordinate-descent sweeps (core_op_matrix.py:765-917).  One step = aqc_ws_cd_sweeps(nsweeps = 1)
    over all lanes: thetas host -> device, ONE persistent launch (a workgroup per lane keeps w and z in LDS for the whole walk,
    z = V^H U re-derived inside), thetas and objective values back to the host.  Metric: sweeps per second."""
    import ctypes

    from aqc_research_amd import _lib as L
    from aqc_research_amd.core_op_matrix import coord_descent_single_sweep
    from aqc_research_amd.engine import BUF_Y, HipContext, Workspace
    from oracle import aqc_ref as cref

    ent = ParametricCircuit_for(w)
    circ = ent
    n, T, d = circ.num_qubits, circ.num_thetas, circ.dimension
    B = args.batch if args.batch > 0 and full else 1536   # 3 workgroups of 50 KiB LDS per CU x 256 CUs = 768 lanes in flight: two full rounds
    K, W = (args.steps, args.warmup) if full else (max(1, min(args.steps, args.config_steps)), 2)
    rng = np.random.default_rng(4321 + 7 * (env.rank + 1))
    distinct = [np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0] for _ in range(16)]
    targets = np.stack([distinct[b % 16] for b in range(B)])
    th0 = np.pi * (2 * rng.random((B, T)) - 1)
    ws = Workspace(HipContext.of(circ), batch=B, ncols=d, device=env.local_rank)
    ws.upload(BUF_Y, targets)
    lib = L.lib()
    fobj = np.zeros((B, 1))

    def sweep(th):
        L.check(lib.aqc_ws_cd_sweeps(ws.handle, L.dptr(th), L.dptr(fobj), 1, -1))

    # the work is checked, not assumed: the first sweep of 8 lanes against the C restatement of the reference algorithm
    # (whole sweeps: ~T sequential Newton steps amplify rounding, hence 1e-8; single steps are pinned at 1e-10 in tests/)
    th = th0.copy()
    sweep(th)
    lanes = sorted(set(int(round(v)) for v in np.linspace(0, B - 1, min(B, 8))))
    t_ref, f_ref = cref.coord_descent_sweeps(circ, th0[lanes], targets[lanes], 1, threads=host_cores())
    parity = max(float(np.abs(th[lanes] - t_ref).max()), float(np.abs(fobj[lanes, 0] - f_ref[:, 0]).max()))
    if not parity < 1e-8:
        raise SystemExit(f"bench.py: cd sweep deviates from the oracle by {parity:g} (> 1e-8)")
    f_first = float(fobj[:, 0].mean())
    for _ in range(W):
        sweep(th)
    env.comm.barrier()
    t0 = time.perf_counter()
    ws.timer_start()
    for _ in range(K):
        sweep(th)
    ev_ms = ws.timer_stop()
    env.comm.barrier()
    wall = time.perf_counter() - t0
    if env.comm.size > 1:
        wall = float(env.comm.allreduce(np.array([wall]), "max")[0])
    f_last = float(fobj[:, 0].mean())
    ws.profile(True)
    for _ in range(3):
        sweep(th)
    from aqc_research_amd.engine import K_MISC

    launches, kern_ms = ws.profile_get(K_MISC)
    ws.profile(False)
    ws.close()
    # one lane: the persistent launch and the launch chain it replaced (2 launches per parameter + 1 per block)
    one, chain = None, None
    if env.rank == 0:
        th1 = th0[0].copy()
        coord_descent_single_sweep(circ, th1, targets[0], None)
        t1 = time.perf_counter()
        for _ in range(20):
            coord_descent_single_sweep(circ, th1, targets[0], None)
        one = (time.perf_counter() - t1) / 20
        os.environ["AQC_CD_CHAIN"] = "1"
        th1 = th0[0].copy()
        coord_descent_single_sweep(circ, th1, targets[0], None)
        t1 = time.perf_counter()
        for _ in range(5):
            coord_descent_single_sweep(circ, th1, targets[0], None)
        chain = (time.perf_counter() - t1) / 5
        del os.environ["AQC_CD_CHAIN"]
    if env.rank != 0:
        return None
    value = K * B * env.n_gpus / wall
    avg_launch_ms = kern_ms / max(launches, 1)
    # bytes the reference's algorithm moves per sweep and lane: per parameter one pass over w, z for the two inner products and
    # one read-modify-write pass for the two rotations, per block the entangler on both, plus the T gate passes of V^H on z
    N = d * d
    alg_bytes = 16.0 * N * (6 * T + 4 * circ.num_blocks + 2 * T)
    lds_peak = 256 * 128 * 2.4e9 / 1e9      # GB/s: 128 B per clock and CU
    achieved = alg_bytes * B / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    out = {
        "metric": "coordinate-descent sweeps/sec", "value": value, "unit": "sweeps/s", "n_gpus": env.n_gpus, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["desc"], "n_qubits": n, "num_thetas": T, "batch_per_gpu": B, "path": "matrix (core_op_matrix.coord_descent_single_sweep)",
                   "ranks_seen": env.ranks_seen, "transport": env.comm.transport,
                   "mean_fobj_after_first_sweep": f_first, "mean_fobj_after_last_sweep": f_last, "sweeps_per_lane": 1 + W + K},
        # the persistent kernel never goes to HBM between parameters: what it moves is LDS traffic
        "roofline": {"bound": "lds", "kernel": "cd_persistent_kernel", "achieved": achieved, "peak": lds_peak, "unit": "GB/s",
                     "frac": achieved / lds_peak, "traffic": None, "avg_launch_ms": avg_launch_ms,
                     "algorithmic_bytes_per_sweep_and_lane": alg_bytes,
                     "note": "algorithmic bytes = 16 B x d^2 x (8 T + 4 L) per sweep and lane (what the reference's pass-per-gate algorithm moves), "
                             "served from LDS: one workgroup per lane holds both operands for the whole walk; HBM sees the target once per sweep. "
                             "The walk is a chain of T dependent steps (two barriers and one fixed-order reduction each): latency-bound, not "
                             "bandwidth-bound",
                     "hbm_equivalent_frac_of_8TBps": achieved / HBM_PEAK_GBS},
        "parity_maxerr": parity, "parity_lanes_checked": len(lanes), "parity_tolerance": 1e-8,
        "single_lane": {"ms_per_sweep_one_launch": None if one is None else one * 1e3, "ms_per_sweep_launch_chain": None if chain is None else chain * 1e3,
                        "speedup_one_lane": None if not one else chain / one,
                        "speedup_all_lanes_vs_chain": None if not chain else value / env.n_gpus * chain},
        "device_ms_per_step_events": ev_ms / K,
    }
    if full and not args.no_cpu_baseline:
        cores = host_cores()
        nb = max(cores, 16)
        t1 = time.perf_counter()
        cref.coord_descent_sweeps(circ, th0[:nb], targets[:nb], 1, threads=cores)
        one_round = time.perf_counter() - t1
        rounds = max(1, int(8.0 / max(one_round, 1e-3)))
        t1 = time.perf_counter()
        cref.coord_descent_sweeps(circ, th0[:nb], targets[:nb], rounds, threads=cores)
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": nb * rounds / dt, "unit": "sweeps/s", "cores": cores, "kind": "port",
                               "sample": f"{nb} lanes x {rounds} sweeps of the same ansatz in {dt:.1f} s: C restatement of coord_descent_single_sweep "
                                         f"(oracle/aqc_ref.c), {cores} thread(s), one lane per thread",
                               "host": {"cpu_model": cpu_model(), "os_cpu_count": os.cpu_count()},
                               "reference_numpy": reference_numpy_record("cd5_cyc180")}
    return out


def run_mps_engine(args, w, env, full):
    """--workload mps32_trotter2_engine: registers beyond dense reach.  One step = one objective+gradient evaluation per lane on the
    native MPS engine (mps_engine.evaluate_lanes: V^H|target>, <neel|.>, gate-by-gate gradient; mps_dot_objective.py:41-242 with
    the arithmetic the reference hands to qiskit-aer), thetas changing every step.  Truncated arithmetic: parity-unpinned (qiskit-aer
    is absent); the check below is the engine against itself at trunc_thr -> 0 through a central difference of the objective."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index

    n, layers, thr = w["n"], w["layers"], float(w["trunc_thr"])
    B = args.batch if args.batch > 0 and full else w["lanes"]
    K, W = (max(1, min(args.steps, 10)), 1) if full else (2, 1)
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
    T = circ.num_thetas
    neel = neel_state_index(n)
    evol = 0.6 * layers
    th0 = init_ansatz_to_trotter(circ, np.zeros(T), evol_time=evol, delta=1.0)
    tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
    tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=evol, delta=1.0)
    rng = np.random.default_rng(99 + env.rank)
    basis = me.DeviceMPS.basis_state(n, neel, device=env.local_rank)
    distinct = []
    for b in range(min(B, 8)):   # eight different targets shared round-robin by the lanes: Trotter states of slightly different evolution times
        tb = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=evol * (1.0 + 0.01 * b), delta=1.0)
        distinct.append(me.v_mul_mps(tcirc, tb, basis, trunc_thr=1e-12))
    targets = [distinct[b % len(distinct)] for b in range(B)]
    bonds = int(max(t.bond_dims.max() for t in distinct))

    def thetas(i):
        return th0[None, :] + 0.02 * np.random.default_rng(1000 * i + env.rank).standard_normal((B, T))

    h, g = me.evaluate_lanes(circ, thetas(0), targets, basis, trunc_thr=thr, method="lockstep")
    # the lockstep lanes against the single-lane engine (same arithmetic, lane by lane) on a few lanes of the first step
    nchk = min(B, 4)
    hs, gs = me.evaluate_lanes(circ, thetas(0)[:nchk], targets[:nchk], basis, trunc_thr=thr, method="threads")
    lane_err = float(max(np.abs(h[:nchk] - hs).max(), np.abs(g[:nchk] - gs).max()))
    # consistency of value and gradient (engine against itself): d|h|^2/dtheta_k by a central difference on lane 0
    k, eps = 3 * n + 7, 1e-5
    tp, tm = thetas(0)[0].copy(), thetas(0)[0].copy()
    tp[k] += eps
    tm[k] -= eps
    hp, _ = me.evaluate_lanes(circ, tp[None, :], targets[:1], basis, trunc_thr=thr)
    hm, _ = me.evaluate_lanes(circ, tm[None, :], targets[:1], basis, trunc_thr=thr)
    fd = (abs(hp[0]) ** 2 - abs(hm[0]) ** 2) / (2 * eps)
    an = 2.0 * float(np.real(np.conj(h[0]) * g[0, k]))
    consistency = abs(fd - an)
    for i in range(W):
        me.evaluate_lanes(circ, thetas(1 + i), targets, basis, trunc_thr=thr, method="lockstep")
    env.comm.barrier()
    t0 = time.perf_counter()
    for i in range(K):
        h, g = me.evaluate_lanes(circ, thetas(10 + i), targets, basis, trunc_thr=thr, method="lockstep")
    env.comm.barrier()
    wall = time.perf_counter() - t0
    if env.comm.size > 1:
        wall = float(env.comm.allreduce(np.array([wall]), "max")[0])
    # roofline of the dominant kernel (lanes_gate2_kernel: a truncated 2-qubit gate = one-sided Jacobi SVD per lane in LDS): fp64 flops of
    # the rotations that really ran (counted on the device) / the summed duration of its launches (event pairs around each), two more steps
    roof = None
    ls = me._lockstep_for(n, B, env.local_rank, targets, basis)
    ls.gate2_stats(enable=True, reset=True)
    for i in range(2):
        me.evaluate_lanes(circ, thetas(30 + i), targets, basis, trunc_thr=thr, method="lockstep")
    st = ls.gate2_stats(enable=False)
    if st["launches_timed"] > 0 and st["launch_ms"] > 0:
        tfl = st["jacobi_flops"] / (st["launch_ms"] * 1e-3) / 1e12
        roof = {"bound": "valu fp64 (latency-bound in practice)", "kernel": "lanes_gate2_kernel", "achieved": tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tfl / FP64_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": st["launch_ms"] / st["launches_timed"],
                "flops_per_launch": st["jacobi_flops"] / st["launches_timed"],
                "flop_convention": "Jacobi rotations that ran: per column pair and sweep 36 flop per row of the work matrix (three inner products, "
                                   "rotation of the pair) + 20 per row of V; counted by the kernel",
                "svds_per_step": st["svds"] / 2, "sweeps_per_svd": st["sweeps"] / max(st["svds"], 1), "rotations_per_svd": st["rotations"] / max(st["svds"], 1),
                "share_of_step": (st["launch_ms"] / 2) / (wall / K * 1e3)}
    # pinned check: the same lanes at trunc_thr -> 0 on a register the C oracle reaches and whose bonds cannot outgrow the lanes (10 qubits, same ansatz family): amplitude and
    # complex gradient of 4 lanes against the dense restatement of the reference
    from oracle import aqc_ref as cref
    from oracle.aqc_oracle import mps_to_vector as orc_mps_to_vector

    n_small = 10   # (no bond of a 10-qubit register exceeds the lanes' 32: nothing is truncated at trunc_thr -> 0)
    circ_s = TrotterAnsatz(n_small, make_trotter_like_circuit(n_small, layers), second_order=True)
    ths = init_ansatz_to_trotter(circ_s, np.zeros(circ_s.num_thetas), evol_time=evol, delta=1.0)[None, :] + \
        0.02 * np.random.default_rng(7).standard_normal((4, circ_s.num_thetas))
    tcirc_s = TrotterAnsatz(n_small, make_trotter_like_circuit(n_small, 3 * layers), second_order=True)
    neel_s = neel_state_index(n_small)
    basis_s = me.DeviceMPS.basis_state(n_small, neel_s, device=env.local_rank)
    tgt_s = me.v_mul_mps(tcirc_s, init_ansatz_to_trotter(tcirc_s, np.zeros(tcirc_s.num_thetas), evol_time=evol, delta=1.0), basis_s, trunc_thr=1e-30)
    hs4, gs4 = me.evaluate_lanes(circ_s, ths, tgt_s, basis_s, trunc_thr=1e-30, method="lockstep")
    dense_t = orc_mps_to_vector(tgt_s.to_qiskit())
    parity = 0.0
    for b in range(4):
        h_ref, g_ref = cref.eval_batch(circ_s, ths[b][None, :], dense_t, neel_s, 1)
        parity = max(parity, abs(hs4[b] - h_ref[0]), float(np.abs(gs4[b] - g_ref[0]).max()))
    tgt_s.close(); basis_s.close()
    if not parity < PARITY_TOL:
        print(f"bench.py: {w['desc'][:40]}: lockstep lanes at trunc_thr -> 0 deviate from the oracle by {parity:g}", file=sys.stderr)
        os._exit(3)
    me.evaluate_lanes(circ, thetas(49)[:1], targets[:1], basis, trunc_thr=thr)   # (sizes the one-lane batch's launches)
    t1 = time.perf_counter()
    me.evaluate_lanes(circ, thetas(50)[:1], targets[:1], basis, trunc_thr=thr)
    one = time.perf_counter() - t1
    t1 = time.perf_counter()
    me.evaluate_lanes(circ, thetas(51)[:1], targets[:1], basis, trunc_thr=thr, method="threads")
    one_engine = time.perf_counter() - t1
    nthr = min(B, 16)   # the same batch shape on host threads (one single-lane engine per lane), for the record
    me.evaluate_lanes(circ, thetas(60)[:nthr], targets[:nthr], basis, trunc_thr=thr, method="threads")
    t2 = time.perf_counter()
    me.evaluate_lanes(circ, thetas(61)[:nthr], targets[:nthr], basis, trunc_thr=thr, method="threads")
    thr_rate = nthr / (time.perf_counter() - t2)
    # what an unmodified caller of the reference's objective gets at this size: SpSurrogateObjectiveFastMpsTrotter (objective() + gradient()
    # pairs as an optimizer issues them, objective_lhs_sur_fast_mps_trotter.py:99-227) -- two lockstep lanes inside
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_fast_mps_trotter import SpSurrogateObjectiveFastMpsTrotter

    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: neel, enable_optim_stats=False, verbose=0, maxiter=5, trunc_thr=thr,
                device=env.local_rank)
    objv = SpSurrogateObjectiveFastMpsTrotter(user_parameters=user, circ=circ)
    objv.set_target(distinct[0].to_qiskit())
    objv.objective(thetas(70)[0]); objv.gradient(thetas(70)[0])
    t3 = time.perf_counter()
    npairs = 3
    for i in range(npairs):
        objv.objective(thetas(71 + i)[0]); objv.gradient(thetas(71 + i)[0])
    pair = (time.perf_counter() - t3) / npairs
    pair_route = "two lockstep lanes" if getattr(objv, "_lk_live", False) else "single-lane engine"
    for m in distinct + [basis]:
        m.close()
    if env.rank != 0:
        return None
    return {
        "metric": "objective+gradient evals/sec", "value": K * B * env.n_gpus / wall, "unit": "evals/s", "n_gpus": env.n_gpus, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["desc"], "n_qubits": n, "num_thetas": T, "batch_per_gpu": B, "path": "MPS front door (mps_dot_objective), native MPS engine",
                   "mps_trunc_thr": thr, "target_max_bond": bonds, "mean_fidelity_term": float(np.mean(np.abs(h) ** 2)), "ranks_seen": env.ranks_seen,
                   "lanes": "lockstep, device-resident (aqc_mpsb_eval): one launch per step of the gate walk for all lanes, rank decisions on the device, bonds <= 32"},
        "roofline": roof,
        "parity_maxerr": parity, "parity_lanes_checked": 4,
        "parity_note": "parity_maxerr: the lockstep lanes at trunc_thr -> 0 on the 10-qubit member of the same ansatz family against the C oracle "
                       "(amplitude + complex gradient, 4 lanes); the TRUNCATED arithmetic of the timed 32-qubit workload stays parity-unpinned "
                       "(qiskit-aer absent); value / gradient consistency of the engine there: "
                       f"|central difference - analytic| = {consistency:.2e} on one parameter",
        "value_gradient_consistency": consistency,
        "lockstep_vs_single_lane_maxerr": lane_err, "lockstep_lanes_checked": nchk,
        "single_lane": {"ms_per_eval": one * 1e3, "evals_per_s": 1.0 / one, "route": "one lockstep lane",
                        "single_lane_engine_ms_per_eval": one_engine * 1e3},
        "host_thread_lanes": {"lanes": nthr, "evals_per_s": thr_rate, "route": "single-lane engine per lane, host threads"},
        "front_door_single_lane": {"object": "SpSurrogateObjectiveFastMpsTrotter.objective() + .gradient()", "route": pair_route, "ms_per_pair": pair * 1e3,
                                   "pairs_per_s": 1.0 / pair},
    }


def run_mps_opt(args, w, env, full):
    """--workload mps32_trotter2_opt: what the lockstep lanes are for.  One step = one whole optimisation of B restarts (random
    perturbations of the Trotter point) of a horizon beyond dense reach: BatchedMpsSurrogateObjective (the surrogate objective of
    objective_lhs_sur_fast_mps_trotter.py:99-227, V^H once per restart + both gradient walks on 2B lockstep lanes) under
    batched_lbfgs.  The reference runs one such optimisation per joblib process on qiskit-aer (job_executor.py:141)."""
    from aqc_research_amd import TrotterAnsatz
    from aqc_research_amd import mps_engine as me
    from aqc_research_amd.batched_optimizer import BatchedMpsSurrogateObjective, batched_lbfgs
    from aqc_research_amd.circuit_structures import make_trotter_like_circuit
    from aqc_research_amd.model_sp_lhs.trotter import init_ansatz_to_trotter, neel_state_index

    n, layers, thr, maxiter = w["n"], w["layers"], float(w["trunc_thr"]), int(w["maxiter"])
    B = args.batch if args.batch > 0 and full else w["lanes"]
    K = max(1, min(args.steps, 3)) if full else 1
    circ = TrotterAnsatz(n, make_trotter_like_circuit(n, layers), second_order=True)
    T = circ.num_thetas
    neel = neel_state_index(n)
    evol = 0.6 * layers
    th0 = init_ansatz_to_trotter(circ, np.zeros(T), evol_time=evol, delta=1.0)
    tcirc = TrotterAnsatz(n, make_trotter_like_circuit(n, 3 * layers), second_order=True)
    tth = init_ansatz_to_trotter(tcirc, np.zeros(tcirc.num_thetas), evol_time=evol, delta=1.0)
    basis = me.DeviceMPS.basis_state(n, neel, device=env.local_rank)
    target = me.v_mul_mps(tcirc, tth, basis, trunc_thr=1e-12)
    obj = BatchedMpsSurrogateObjective(circ, target, lanes=B, base_index=neel, trunc_thr=thr, device=env.local_rank)

    def start(i):
        return th0[None, :] + 0.05 * np.random.default_rng(500 * i + env.rank).standard_normal((B, T))

    obj.reset_state()
    obj.value_and_grad(start(0))            # warm-up: sizes the lanes' launches
    fid0 = obj.fidelity.copy()
    env.comm.barrier()
    t0 = time.perf_counter()
    evals = 0
    fid = None
    for i in range(K):
        obj.reset_state()
        before = obj.num_evals
        res = batched_lbfgs(obj.value_and_grad, start(i), maxiter=maxiter)
        evals += obj.num_evals - before
        fid = obj.fidelity.copy()
    env.comm.barrier()
    wall = time.perf_counter() - t0
    if env.comm.size > 1:
        wall = float(env.comm.allreduce(np.array([wall]), "max")[0])
    obj.close()
    for m in (target, basis):
        m.close()
    if env.rank != 0:
        return None
    return {
        "metric": "objective+gradient evals/sec", "value": evals * env.n_gpus / wall, "unit": "evals/s", "n_gpus": env.n_gpus, "steps": K, "warmup": 1,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["desc"], "n_qubits": n, "num_thetas": T, "batch_per_gpu": B, "path": "BatchedMpsSurrogateObjective under batched_lbfgs (lockstep lanes of the native MPS engine)",
                   "mps_trunc_thr": thr, "lbfgs_maxiter": maxiter, "evals_per_optimisation": evals / (K * B), "iterations": [int(res["nit"].min()), int(res["nit"].max())],
                   "fidelity_start_mean": float(np.mean(fid0)), "fidelity_end_mean": float(np.mean(fid)), "fidelity_end_min": float(np.min(fid)),
                   "ranks_seen": env.ranks_seen},
        "roofline": None, "parity_maxerr": None, "parity_lanes_checked": 0,
        "parity_note": "truncated MPS arithmetic is parity-unpinned (qiskit-aer absent); the objective is tested against the dense batched objective at "
                       "10 qubits (tests/test_hip_round4.py)",
        "optimisations_per_s": K * B * env.n_gpus / wall,
    }


def ParametricCircuit_for(w):
    return build_circuit(dict(w, kind="cyclic"))


def launch_ranks(n_ranks, argv):
    """`bench.py --gpus N` started as ONE plain process: it becomes the launcher -- N fresh child processes of this script,
    one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / AQC_COMM_FILE in their environment, exactly what
    torch.distributed.run would set), rank 0's JSON line relayed.  The launcher itself never touches the GPU (no HIP call,
    no library of the package loaded) and never replaces itself by another program.  Non-zero exit if any rank fails; the
    other ranks are terminated then (a dead rank would otherwise leave them waiting in a collective until its time-out)."""
    import shutil
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="aqc_bench_")
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AQC_COMM_FILE=os.path.join(tmp, "rccl_unique_id"),
                   AQC_COMM_TAG=os.path.basename(tmp), AQC_BENCH_LAUNCHER="self")   # the tag names THIS launch inside the id file
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc, out0 = 0, b""
    try:
        pending = set(range(n_ranks))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if code != 0:
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                    rc = code if code > 0 else 1
            if pending and rc == 0:
                if 0 in pending:   # keep rank 0's pipe drained (one line; it cannot fill, but do not rely on that)
                    pass
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=20)
            except subprocess.TimeoutExpired:
                q.kill()
        shutil.rmtree(tmp, ignore_errors=True)
    if rc == 0:
        sys.stdout.write(out0.decode())
        sys.stdout.flush()
    return rc


def objective_object_rates(circ, targets, rng, device):
    """Evaluations per second through the objective objects a user of the reference holds (one evaluation = one
    objective(theta) + gradient(theta) pair on the object, objective_lhs_sur_max.py:82-191, consumed by optimizer.py:579-590):
    the drop-in SpSurrogateObjectiveMax under AqcOptimizer(lbfgs) (one lane, host optimizer), the lane-batched surrogate's
    value_and_grad (64 lanes) and the device-resident multi-start L-BFGS on it."""
    from aqc_research_amd.batched_optimizer import BatchedSurrogateObjective
    from aqc_research_amd.model_sp_lhs.objective_lhs_sur_max import SpSurrogateObjectiveMax
    from aqc_research_amd.optimizer import AqcOptimizer

    n, T = circ.num_qubits, circ.num_thetas
    out = {}
    user = dict(num_qubits=n, max_flips=1, state_prep_func=lambda _n: 0, enable_optim_stats=False, verbose=0, maxiter=40, device=device)
    objv = SpSurrogateObjectiveMax(user_parameters=user, circ=circ, front_layer=True)
    objv.set_target(targets[0])
    th0 = 0.2 * np.pi * (2 * rng.random(T) - 1)
    AqcOptimizer(optimizer_name="lbfgs", maxiter=3).optimize(objv, circ, th0)   # warm-up (graph capture, first launches)
    t0 = time.perf_counter()
    res = AqcOptimizer(optimizer_name="lbfgs", maxiter=40).optimize(objv, circ, th0)
    dt = time.perf_counter() - t0
    out["SpSurrogateObjectiveMax_under_AqcOptimizer_lbfgs"] = {"lanes": 1, "evals_per_s": res["num_fun_ev"] / dt, "pairs": int(res["num_fun_ev"]),
                                                              "seconds": dt, "final_fidelity": float(res["fidelity"])}
    def value_and_grad_rate(lanes):
        bo = BatchedSurrogateObjective(circ, targets[:lanes], base_index=0, device=device)
        th = 0.2 * np.pi * (2 * rng.random((lanes, T)) - 1)
        f, g = bo.value_and_grad(th)
        t0 = time.perf_counter()
        calls = 0
        while calls < 20 or time.perf_counter() - t0 < 0.5:
            th = th - 0.05 * g
            f, g = bo.value_and_grad(th)
            calls += 1
        dt = time.perf_counter() - t0
        return bo, th, {"lanes": lanes, "evals_per_s": calls * lanes / dt, "calls": calls, "seconds": dt,
                        "lanes_led_by_a_flip_state": int((bo.max_no != 0).sum())}

    if targets.shape[0] > 64:   # the same object at the bench's own lane count (the headline's unit of work)
        bo, _, out["BatchedSurrogateObjective_value_and_grad_bench_lanes"] = value_and_grad_rate(targets.shape[0])
        bo.close()
    lanes = min(64, targets.shape[0])
    bo, th, out["BatchedSurrogateObjective_value_and_grad"] = value_and_grad_rate(lanes)
    bo.reset_state()
    bo.minimize_on_device(th, maxiter=2)   # warm-up
    bo.reset_state()
    t0 = time.perf_counter()
    dev = bo.minimize_on_device(th, maxiter=20)
    dt = time.perf_counter() - t0
    out["minimize_on_device"] = {"lanes": lanes, "evals_per_s": dev["nfev"] * lanes / dt, "batched_evaluations": dev["nfev"], "seconds": dt,
                                 "mean_fidelity": float(np.mean(dev["fidelity"]))}
    bo.close()
    return out


def rank_echo(args):
    """Launcher self-test (CPU, no HIP): every rank joins a gloo group from its environment and rank 0 prints who is there."""
    real_stdout = os.dup(1)   # one JSON line on stdout, library chatter to stderr (as in main)
    os.dup2(2, 1)
    import torch.distributed as dist

    if os.environ.get("AQC_BENCH_ECHO_FAIL_RANK") == os.environ.get("RANK"):
        os._exit(7)   # launcher self-test: this rank dies before the rendezvous
    dist.init_process_group(backend="gloo")
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, (int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), os.environ.get("AQC_COMM_FILE", "")))
    if dist.get_rank() == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"n_gpus": args.gpus, "world_size": dist.get_world_size(), "ranks_seen": sorted(r for r, _, _ in seen),
                          "local_ranks": sorted(l for _, l, _ in seen), "comm_files": len({f for _, _, f in seen})}), flush=True)
        os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()


def stage_launch_table(ws, prof_log, prof_steps, N, B, sparse_lhs=False):
    """Per stage launch of one step (matrix-core path): kernel, plan stage, sub-stages, share of the tiles it ran over, average
    duration (HIP events around every launch of `prof_steps` steps), EXECUTED MFMA flops (288 per amplitude and sub-stage for
    the sweep: 9 real 16x16x16 products per 256 amplitudes; 96 for V^H) and their rate.  None off the matrix-core path."""
    from aqc_research_amd._lib import K_APPLY, K_APPLY_LIST, K_APPLY_VIRTUAL, K_PROJECT, K_SWEEP, K_SWEEP_LIST, K_SWEEP_VIRTUAL

    if ws.kernel_family(1) != 3 or ws.kernel_family(0) != 3 or not prof_log or len(prof_log) % prof_steps:
        return None
    per = len(prof_log) // prof_steps
    kinds = [k for k, _ in prof_log[:per]]
    if any([k for k, _ in prof_log[i * per:(i + 1) * per]] != kinds for i in range(prof_steps)):
        return None
    avg = [sum(prof_log[i * per + j][1] for i in range(prof_steps)) / prof_steps for j in range(per)]
    sw_items, _, vd_items = ws.sparse_counts()
    rows, at = [], {0: 0, 1: 0}
    proj = ws.projected_info()   # the sweep's stages after the first on a virtual register (csrc/aqc_ws_project.cpp), or {}
    by_projection = bool(proj) and K_APPLY_VIRTUAL in kinds   # ... and V^H of the evaluation by projections of the target (same file)
    fused = by_projection and kinds.count(K_PROJECT) == 1        # both of them from one fetch (project_fused_kernel)
    vstage = vapply = nproj = napply_list = 0

    def mfma_row(kernel, plan, stage, nsub, tiles_frac, ms, flops, **extra):
        row = {"kernel": kernel, "plan": plan, "stage": stage, "substages": nsub, "last_substage_r_only": False, "tiles_frac": tiles_frac, "avg_ms": ms,
               "flops": flops, "flops_not_issued_zero_w": 0.0, "TFLOPs": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
               "frac_of_peak": flops / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if ms > 0 else 0.0}
        row.update(extra)
        return row

    for j, k in enumerate(kinds):
        if k == K_PROJECT and proj:   # one pass over z (the checkpoint of V^H) or over the target: 2^(touched + summed bits) elements per item, memory-bound
            cols = 16 * max(1, (1 << proj["shared_with_first_stage"]) // 16)
            elems = float(sw_items) * 2.0 ** (proj["touched_qubits"] + proj["summed_bits"])
            flops = 8.0 * elems * cols
            nbytes = 16.0 * elems
            second = by_projection and nproj == 1
            if fused:   # two products, three real MFMAs per K-step each
                flops = 2.0 * 6.0 * elems * cols
                rows.append(mfma_row("project_fused_kernel<2>", "lhs tile of (later stages)^H y AND projection of the target, one fetch of the target",
                                     None, None, None, avg[j], flops, bound="hbm", bytes_read=nbytes, GBps=nbytes / (avg[j] * 1e-3) / 1e9 if avg[j] > 0 else 0.0,
                                     frac_of_hbm_peak=nbytes / (avg[j] * 1e-3) / 1e9 / HBM_PEAK_GBS if avg[j] > 0 else 0.0))
                nproj += 1
                continue
            rows.append(mfma_row("project_kernel<4, 1>" if second else "project_staged_kernel",
                                 ("lhs tile of (later stages)^H y: sum over the touched qubits" if second else
                                  ("projection of the target onto the lhs subspace" if by_projection else "projection of z onto the lhs subspace")),
                                 None, None, None, avg[j], flops, bound="hbm", bytes_read=nbytes, GBps=nbytes / (avg[j] * 1e-3) / 1e9 if avg[j] > 0 else 0.0,
                                 frac_of_hbm_peak=nbytes / (avg[j] * 1e-3) / 1e9 / HBM_PEAK_GBS if avg[j] > 0 else 0.0))
            nproj += 1
            continue
        if k in (K_SWEEP_VIRTUAL, K_APPLY_VIRTUAL) and proj:
            sweep_kind = k == K_SWEEP_VIRTUAL
            per = proj["substages_per_stage"]
            if sweep_kind:
                st_v = vstage
                vstage += 1
            else:   # the virtual plan forwards (M_end), then backwards (Y_0)
                st_v = vapply if vapply < len(per) else 2 * len(per) - 1 - vapply
                vapply += 1
            nsub = per[st_v] if 0 <= st_v < len(per) else 0
            flops = (288.0 if sweep_kind else 96.0) * float(sw_items) * 2.0 ** proj["padded_qubits"] * nsub
            name = f"sweep_mfma_kernel<{proj['tile_bits']}, true, false, false>" if sweep_kind else f"apply_mfma_kernel<{proj['tile_bits']}, true>"
            what = (f"sweep, stages after the first on {proj['virtual_qubits']} virtual qubits" if sweep_kind else
                    (f"virtual lhs pattern through the later stages' gates" if vapply <= len(per) else "virtual z: (later stages)^H of the projection"))
            rows.append(mfma_row(name, what, st_v, nsub, None, avg[j], flops))
            continue
        if k == K_APPLY_LIST and by_projection:   # first: psi (the sweep's first stage applied to the lhs tiles); second: V^H's last stage on them
            stages, tile_bits, ntiles = ws.plan_info(1)
            nsub = ws.plan_stage(1, 0)[0]
            frac = sw_items / float(ntiles * B)
            rows.append(mfma_row(f"apply_mfma_kernel<{tile_bits}, true>", "first stage's gates on the lhs tiles (psi)" if napply_list == 0 else "V^H",
                                 0 if napply_list == 0 else stages - 1, nsub, frac, avg[j], 96.0 * N * B * frac * nsub))
            napply_list += 1
            continue
        if k not in (K_APPLY, K_APPLY_LIST, K_SWEEP, K_SWEEP_LIST):
            continue
        which = 1 if k in (K_SWEEP, K_SWEEP_LIST) else 0
        stages, tile_bits, ntiles = ws.plan_info(which)
        st = at[which]
        at[which] += 1
        if st >= stages:
            return None
        nsub = ws.plan_stage(which, st)[0]
        frac = 1.0
        if k == K_SWEEP_LIST:
            frac = sw_items / float(ntiles * B)
        elif k == K_APPLY_LIST:
            frac = vd_items / float(ntiles * B)
        flops = (288.0 if which else 96.0) * N * B * frac * nsub
        r_only = bool(which) and st == stages - 1 and ws.sweep_r_only_sub() >= 0   # its last sub-stage: R from the inputs, 12 of 36 MFMAs per group
        if which and k == K_SWEEP_LIST and by_projection and 2 <= nsub <= 12 and os.environ.get("AQC_R_ONLY_LAST", "1") != "0":
            r_only = True   # objective by projection: nobody reads what the first stage leaves -- ITS last sub-stage is the R-only one (grad_from_impl)
        if r_only:
            flops -= 192.0 * N * B * frac
        skipped = 0.0
        if which and sparse_lhs and k == K_SWEEP:   # dense launch of a sweep from basis states: zero groups / K-steps of w are not issued
            per_sub = [96.0 + 96.0 * 2.0 ** (g + kk) + 96.0 * 2.0 ** g for g, kk in ws.plan_skips(1, st)]
            skipped = flops - N * B * frac * sum(per_sub)
            flops -= skipped
        over_list = "true" if k in (K_SWEEP_LIST, K_APPLY_LIST) else "false"   # the names rocprofv3 shows: <tile bits, over a tile list[, zero-w skips]>
        rows.append({"kernel": (f"sweep_mfma_kernel<{tile_bits}, {over_list}, {'true' if which and sparse_lhs and k == K_SWEEP and any(g or kk for g, kk in ws.plan_skips(1, st)) else 'false'}, {'true' if r_only else 'false'}>"
                                if which else f"apply_mfma_kernel<{tile_bits}, {over_list}>"),
                     "plan": "sweep" if which else "V^H", "stage": st, "substages": nsub, "last_substage_r_only": r_only, "tiles_frac": frac, "avg_ms": avg[j],
                     "flops": flops, "flops_not_issued_zero_w": skipped, "TFLOPs": flops / (avg[j] * 1e-3) / 1e12 if avg[j] > 0 else 0.0,
                     "frac_of_peak": flops / (avg[j] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if avg[j] > 0 else 0.0})
    if not rows:
        return None
    executed = sum(r["flops"] for r in rows)
    dense = (288.0 * ws.plan_substages(1) + 96.0 * ws.plan_substages(0)) * N * B
    stage_ms = sum(r["avg_ms"] for r in rows)
    return {"launches": rows, "executed_flops_per_step": executed, "dense_route_flops_per_step": dense,
            "zero_products_avoided_frac": 1.0 - executed / dense, "stage_launch_ms_per_step": stage_ms,
            "all_launch_ms_per_step": sum(avg),
            "stage_launches_frac_of_peak": executed / (stage_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if stage_ms > 0 else 0.0,
            "note": "dense_route_flops = every stage over every tile (the sweep from a dense lhs state and a full V^H); the sparse-lhs "
                    "route runs the sweep's first stage over the tiles that hold the lhs basis states and V^H's last stage over the "
                    "tiles the evaluation reads -- products with exact zeros and amplitudes nobody reads are not executed; with the projected "
                    "route (projected_route: csrc/aqc_ws_project.cpp) the sweep's stages after the first run on a virtual register of "
                    "`virtual_qubits` qubits per lane after one pass over z; a one-call evaluation from one basis state replaces the stages of V^H "
                    "by two passes over the target (objective_by_projection)",
            "projected_route": proj or None, "objective_by_projection": by_projection}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=0, help="independent evaluations resident per GPU (default: 1024 state vectors up to 16 qubits, 64 beyond, 32-64 matrices)")
    ap.add_argument("--workload", default="sv16_l40", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--rank-echo", action="store_true", help=argparse.SUPPRESS)   # launcher self-test: no GPU work, see tests/
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the sustained-rate loop after the timed one (0 = off)")
    ap.add_argument("--no-objective-object", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="default workload only: skip the short runs of the other BASELINE configurations")
    ap.add_argument("--config-steps", type=int, default=10, help="timed steps of each short configuration run")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.rank_echo:
        return rank_echo(args)
    if os.environ.get("AQC_DEBUG_SKIP"):   # a work-skipping switch of tuning builds: never time with it
        raise SystemExit("bench.py: AQC_DEBUG_SKIP is set; refusing to time a run that may skip work")

    # stdout carries exactly one JSON line: anything libraries print there (RCCL prints a version banner when a
    # communicator is created) is diverted to stderr until the result is written
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Process group: one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, set by launch_ranks
    # above or by torch.distributed.run).  The transport is aqc_comm -- librccl bound directly through the C ABI, no torch
    # in the process; if it cannot initialise the run FAILS (non-zero exit).  AQC_BENCH_BACKEND=gloo is the rehearsal mode
    # for boxes with fewer GPUs than ranks: ranks share a device and the records travel over the test double of tests/.
    from aqc_research_amd import comm as aqc_comm

    comm = aqc_comm.Communicator()
    comm_note = None
    if world > 1 or os.environ.get("AQC_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AQC_BENCH_BACKEND", "rccl")
        if backend == "gloo":
            import torch.distributed as dist

            from tests.gloo_double import GlooDouble

            dist.init_process_group(backend="gloo")
            comm = GlooDouble(dist)
        elif backend == "rccl":
            try:
                if world > 1:
                    comm = aqc_comm.from_environment()
                else:   # AQC_BENCH_FORCE_DIST=1 on one rank: a one-rank RCCL communicator (init + collectives exercised)
                    import tempfile

                    comm = aqc_comm.RcclCommunicator(0, 1, local_rank, os.path.join(tempfile.gettempdir(), f"aqc_comm_id_bench_{os.getpid()}"))
            except Exception as exc:
                print(f"bench.py: rank {rank}: aqc_comm (direct RCCL) failed to initialise: {exc}", file=sys.stderr)
                os._exit(5)
        else:
            raise SystemExit(f"bench.py: unknown AQC_BENCH_BACKEND {backend!r} (rccl | gloo)")
        aqc_comm.use(comm)
    from aqc_research_amd import _lib as aqc_lib

    ndev = aqc_lib.lib().aqc_device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible; the path has no CPU fallback")
    if local_rank >= ndev:
        if os.environ.get("AQC_BENCH_BACKEND", "rccl") != "gloo":
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {ndev} are visible")
        local_rank %= ndev   # rehearsal: ranks share devices
    n_gpus = world
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    # who is really there: every rank's id through the communicator (one all-gather) -> config.ranks_seen
    ranks_seen = sorted(int(v) for v in comm.allgather(np.array([float(rank)]))[:, 0]) if comm.size > 1 else [0]
    if ranks_seen != list(range(world)):
        print(f"bench.py: rank {rank}: the process group holds ranks {ranks_seen}, expected 0..{world - 1}", file=sys.stderr)
        os._exit(6)
    import types

    env = types.SimpleNamespace(comm=comm, comm_note=comm_note, rank=rank, world=world, local_rank=local_rank, n_gpus=n_gpus,
                                ranks_seen=ranks_seen)
    t_start = time.perf_counter()
    out = measure(args.workload, args, env, full=True)
    # ---- every other BASELINE configuration in the SAME driver-run line: a few timed steps each, parity-checked against the
    # C oracle like the headline (single GPU, default workload only; --no-configs skips them) ---------------------------------
    if args.workload == "sv16_l40" and world == 1 and not args.no_configs:
        configs = {}
        for name in CONFIG_RUNS:
            t_cfg = time.perf_counter()
            try:
                configs[name] = brief(measure(name, args, env, full=False))
            except (SystemExit, Exception) as exc:   # a failed short run must not take the headline with it: it is reported as failed
                configs[name] = {"error": f"{type(exc).__name__}: {exc}"}
            configs[name]["wall_s_incl_setup"] = time.perf_counter() - t_cfg
            print(f"bench.py: config {name}: {json.dumps(configs[name])}", file=sys.stderr, flush=True)
        out["configs"] = configs
        out["configs_note"] = ("short runs of the other BASELINE.json configurations in this same process: --config-steps timed steps "
                               "after 3 warm-up steps each, inputs resident in HBM, thetas changing every step, the last step checked "
                               "against the C oracle (parity_maxerr); roofline_frac = executed MFMA flops of a sweep launch / its HIP-event "
                               "duration / 78.6 TFLOP/s")
        out["total_bench_wall_s"] = time.perf_counter() - t_start
    if rank == 0 and out is not None:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    comm.barrier()
    comm.close()


# the short configuration runs of the default invocation, in BASELINE.json's order (cfg 1, 2 first / last horizon, 3 through the
# MPS front door at the no-truncation and at the reference's default threshold, 4 sizes + job mix, 5)
CONFIG_RUNS = ["mat5_cyc180", "cd5_cyc180", "mat10_l40_k16", "sv12_trotter2", "sv12_trotter12", "mps16_l40_chi64", "mps16_l40_chi64_thr1e-6", "sv20_l40",
               "cfg2_driver", "sv20_trotter2", "cfg4_jobs", "cfg4_driver", "mat10_l40", "mps32_trotter2_engine", "mps32_trotter2_opt"]


def brief(o):
    """What a configs entry of the default line carries (the full record of a workload: --workload NAME)."""
    if o is None:
        return {"error": "no result"}
    r = o.get("roofline") or {}
    b = {("sweeps_per_s" if o.get("unit") == "sweeps/s" else "evals_per_s"): o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "lanes": o["config"].get("batch_per_gpu", o["config"].get("lanes_per_entry")),
         "workload": o["config"]["workload"], "roofline_frac": r.get("frac"), "roofline_kernel": r.get("kernel"),
         "sweep_avg_launch_ms": r.get("avg_launch_ms"), "parity_maxerr": o.get("parity_maxerr"),
         "parity_lanes_checked": o.get("parity_lanes_checked")}
    for k in ("front_door_single_lane", "jobs_per_s", "mean_fidelity", "single_lane", "host_thread_lanes", "lockstep_vs_single_lane_maxerr", "optimisations_per_s",
              "fidelity_start_mean", "fidelity_end_mean", "lbfgs_maxiter", "horizons_per_s", "horizons", "evaluations", "fidelity_best_per_horizon",
              "fidelity_threshold_per_horizon", "lbfgs",
              "value_gradient_consistency", "parity_note"):
        if k in o:
            b[k] = o[k]
        elif k in o["config"]:
            b[k] = o["config"][k]
    return b


def measure(workload, args, env, full):
    """One workload: set-up, warm-up, K timed steps between barriers, parity check of the last timed step, kernel profile.
    ``full`` adds the sustained loop, single-evaluation latency, objective objects and the CPU baseline (the headline run);
    the short configuration runs of the default line leave them out and use --config-steps steps."""
    comm, comm_note, rank, world, local_rank, n_gpus, ranks_seen = (env.comm, env.comm_note, env.rank, env.world, env.local_rank,
                                                                     env.n_gpus, env.ranks_seen)
    from aqc_research_amd._lib import K_APPLY_LIST, K_APPLY_VIRTUAL, K_PROJECT, K_SWEEP_LIST, K_SWEEP_VIRTUAL
    from aqc_research_amd.engine import BUF_X, BUF_Y, BUF_Z, K_APPLY, K_COEF, K_FINALIZE, K_MISC, K_SWEEP, HipContext, Workspace
    from oracle import aqc_oracle as orc

    w = WORKLOADS[workload]
    if w["kind"] == "cd":
        return run_cd(args, w, env, full)
    if w["kind"] == "mps_engine":
        return run_mps_engine(args, w, env, full)
    if w["kind"] == "mps_opt":
        return run_mps_opt(args, w, env, full)
    if w["kind"] == "driver":
        return run_driver(args, w, env, full)
    if w["kind"] == "jobs":
        out = run_job_mix(args, w, comm, rank, local_rank, n_gpus, comm_note, steps=(args.steps if full else 1))
        out["config"]["ranks_seen"] = ranks_seen
        return out
    circ = build_circuit(w)
    n, T = circ.num_qubits, circ.num_thetas
    ctx = HipContext.of(circ)
    G = ctx.num_gate_groups
    ncols = w.get("ncols", 1)
    N = (1 << n) * ncols          # complex128 elements per lane
    chi = w.get("chi", 0)
    # lanes per GPU: 1024 for problems of up to 2^16 amplitudes per lane (1 GiB per buffer at 16 qubits; a persistent sweep
    # workgroup then walks 64 tiles, so launch ramp, prologue and tail are amortised: 64 lanes give 156k evals/s at the
    # headline, 256 give 178k, 1024 give 183k -- DESIGN 6), 64 otherwise
    B = args.batch if args.batch > 0 and full else (64 if chi else (1024 if (ncols << n) <= (1 << 16) else (64 if ncols < 256 else 32)))
    K, W = (args.steps, args.warmup) if full else (max(1, min(args.steps, args.config_steps)), 3)
    thr = float(w.get("trunc_thr", 1e-16))

    rng = np.random.default_rng(1234 + 7 * (rank + 1))  # job_executor.py:64 seeding rule
    ws = Workspace(ctx, batch=B, ncols=ncols, device=local_rank)
    flip_idx = orc.flip_state_indices(n, 1)
    mps_targets = None
    if chi:   # a few distinct random Vidal-form targets (their normalisation is host SVD work); lane b of step i works on
        # target (i + b) mod D, so every lane's target CHANGES EVERY STEP.  The tuples keep resident device copies (slot cache
        # of the workspace, found again by identity + fingerprint); what a step pays is the contraction of all lanes' targets
        # and lhs states to dense vectors on the device -- one batched launch chain each (aqc_ws_mps_to_vec_batch).
        distinct = [orc.random_mps(n, chi, rng) for _ in range(min(B, 4))]
        D = len(distinct)
        if thr > 1e-12:   # the reference's default threshold: canonical tensors truncated at `thr`, as Aer hands them over
            from aqc_research_amd import mps_dot_objective as mdo
            from aqc_research_amd.mps_operations import vector_to_canonical_mps

            if not mdo.use_dense(n, thr):
                raise SystemExit(f"bench.py: {workload}: the MPS front door does not take the dense route at trunc_thr = {thr:g}")
            distinct = [vector_to_canonical_mps(orc.mps_to_vector(m) / np.linalg.norm(orc.mps_to_vector(m)), thr) for m in distinct]
        mps_targets = [[distinct[(i + b) % D] for b in range(B)] for i in range(D)]
        zero_mps = ([(np.ones((1, 1), complex), np.zeros((1, 1), complex)) for _ in range(n)], [np.ones(1) for _ in range(n - 1)])
        zero_list = [zero_mps] * B
        dense = np.stack([orc.mps_to_vector(m) for m in distinct])
        targets = dense[[(K + W - 1 + b) % D for b in range(B)]]      # the last timed step's assignment (parity check)
        ws.gather_setup(flip_idx)
    elif ncols == 1:
        targets = np.stack([orc.rand_state(n, rng) for _ in range(B)])
        ws.upload(BUF_Y, targets)
        ws.set_basis(BUF_X, 0)  # x = |0>
        ws.gather_setup(flip_idx)
    else:  # X = I, Y = random unitary per lane (FullRangeSketchingVectors, sk_core.py:317-326); k < d: X = the first k columns of I
        distinct = [np.linalg.qr(rng.standard_normal((1 << n, ncols)) + 1j * rng.standard_normal((1 << n, ncols)))[0] for _ in range(min(B, 8))]
        targets = np.stack([distinct[b % len(distinct)] * np.exp(2j * np.pi * b / max(B, 1)) for b in range(B)])   # (a lane's own phase)
        ws.upload(BUF_Y, targets)
        if ncols == 1 << n:
            ws.set_identity(BUF_X)
        else:
            ws.upload(BUF_X, np.ascontiguousarray(np.broadcast_to(np.eye(1 << n, ncols, dtype=complex), (B, 1 << n, ncols))))
    nsets = min(K + W, 64)       # thetas change every step (nothing is served from a cache); the bank is cycled
    bank = np.pi * (2 * rng.random((nsets, B, T)) - 1)
    ws.theta_bank(bank)

    def step(i):
        ws.use_theta_set(i % nsets)
        if mps_targets is not None:   # QiskitMPS operands arrive as host tuples on every evaluation (mps_dot_objective.py:41)
            ws.mps_to_vec_batch(zero_list, BUF_X)
            ws.mps_to_vec_batch(mps_targets[i % len(mps_targets)], BUF_Y)
        if ncols == 1:   # (the MPS front door too: its operands have just been contracted into X and Y)
            # Z = V^H Y where the evaluation reads it, hs = <state_i|V^H|target> for the registered flip states, the sweep from x:
            # one enqueue (the same launches as apply + gather_launch + grad, which the other branches spell out)
            ws.objective_launch(BUF_X, None, True)
        else:
            ws.apply(True, BUF_Y, BUF_Z)
            if ncols == 1:
                ws.gather_launch(BUF_Z)       # hs = <state_i|V^H|target>
            else:
                ws.vdot_launch(BUF_X, BUF_Z)  # <X|V^H Y>  (sk_core.py:192)
            ws.grad(None, True)
        ws.results_async()   # gradients + amplitudes of THIS step -> pinned host memory (what an optimizer reads every evaluation)

    def barrier():
        ws.sync()          # every launch of this rank's stream has completed (the path does not use torch's stream)
        comm.barrier()

    # Settle phase (untimed set-up, tools/hiccup_probe.py): within the first ~50 ms of sustained launches after start-up
    # this stack shows a one-off stall of 50-80 ms (HIP runtime / driver housekeeping); a short burst of the same
    # work followed by a pause gets it out of the way before the W warm-up and K timed steps below.
    for i in range(min(50, max(W, 1))):
        step(i)
    ws.sync()
    time.sleep(1.0 if full else 0.2)
    for i in range(W):
        step(i)
    barrier()
    t0 = time.perf_counter()
    ws.timer_start()
    for i in range(W, W + K):
        step(i)
    ev_ms = ws.timer_stop()
    barrier()
    wall = time.perf_counter() - t0
    if comm.size > 1:
        wall = float(comm.allreduce(np.array([wall]), "max")[0])

    # per-rank rates from each rank's own device clock (HIP events around its K timed steps): a SCALE record can be read
    # without a rerun (one more all-gather of a double per rank, outside the timed region)
    my_rate = K * B / (ev_ms * 1e-3) if ev_ms > 0 else 0.0
    rank_rates = [float(v) for v in comm.allgather(np.array([my_rate]))[:, 0]] if comm.size > 1 else [my_rate]
    # result record of the last step (checked + gathered: the only inter-GPU traffic)
    hs, grads = ws.results_fetch()   # the host copies the last timed step delivered
    hs = hs.reshape(B, -1)

    # sustained rate: the same step for >= --sustain-seconds of wall clock after the driver-sized loop (power / clock settling)
    sustained = None
    if full and args.sustain_seconds > 0:
        barrier()
        t1 = time.perf_counter()
        done_steps = 0
        while True:
            for i in range(50):
                step(W + K + done_steps + i)
            done_steps += 50
            ws.sync()
            flag = np.array([1.0 if time.perf_counter() - t1 < args.sustain_seconds else 0.0])
            if comm.size > 1:
                comm.allreduce(flag, "max")   # all ranks stop after the same number of steps
            if flag[0] == 0.0:
                break
        barrier()
        t_sus = time.perf_counter() - t1
        if comm.size > 1:
            t_sus = float(comm.allreduce(np.array([t_sus]), "max")[0])
        sustained = (done_steps * B * n_gpus / t_sus, t_sus, done_steps)
    # the shipped gather: one fixed-size record {cost, fidelity, counts, thetas[T]} per lane through run_jobs' own
    # packing + ONE all-gather over the communicator (job_executor._gather_records)
    from aqc_research_amd import job_executor as jex

    lane_records = [{"job_index": rank + comm.size * b, "seed": 0, "time": 0.0, "status": "ok", "cost": float(1.0 - abs(hs[b, 0]) ** 2),
                     "fidelity": float(abs(hs[b, 0]) ** 2), "num_iters": K, "num_fun_ev": K, "num_grad_ev": K,
                     "thetas": bank[(W + K - 1) % nsets][b]} for b in range(B)]
    gathered = jex._gather_records(lane_records, comm, B * comm.size, "fixed") if comm.size > 1 or comm.transport != "none" else lane_records
    if len(gathered) != B * comm.size:
        print(f"bench.py: rank {rank}: gathered {len(gathered)} records, expected {B * comm.size}", file=sys.stderr)
        os._exit(4)

    # ---- the timed work is checked, not assumed: the last step's (hs, gradient) of a few lanes against the C
    # restatement of the reference algorithm (oracle/aqc_ref.c) on the same (theta, target) --------------------
    last_th = bank[(W + K - 1) % nsets]
    # headline-sized problems: 32 lanes spread over the batch (the GPU suite checks all 1024 lanes of this exact workload,
    # tests/test_hip_round3.py::test_headline_all_lanes); 2^20-amplitude lanes: 2 (0.7 s of C oracle each)
    n_check = (32 if full else 8) if N <= (1 << 17) else 2
    check_lanes = sorted(set(int(round(v)) for v in np.linspace(0, B - 1, min(B, n_check))))
    parity = 0.0
    from oracle import aqc_ref as cref

    if ncols == 1:
        for b in check_lanes:   # EVERY delivered output of the lane: all gathered flip-state amplitudes <state_i|V^H|target> and the full gradient
            h_ref, g_ref = cref.eval_batch(circ, last_th[b][None, :], targets[b], 0, 1)
            vh_ref = cref.v_dagger_mul_vec(circ, last_th[b], targets[b])
            parity = max(parity, abs(hs[b, 0] - h_ref[0]), float(np.abs(hs[b] - vh_ref[np.asarray(flip_idx)]).max()),
                         float(np.abs(grads[b] - g_ref[0]).max()))
    else:
        eye = np.eye(1 << n, ncols, dtype=complex)
        for b in check_lanes[:1]:
            vhy = cref.v_dagger_mul_mat(circ, last_th[b], targets[b])
            g_ref = cref.grad_of_matrix_dot_product(circ, last_th[b], eye, vhy)
            parity = max(parity, abs(hs[b, 0] - np.vdot(eye, vhy)) / ncols, float(np.abs(grads[b] - g_ref).max()) / ncols)
    if not parity < PARITY_TOL:
        print(f"bench.py: rank {rank}: timed result deviates from the oracle by {parity:g} (> {PARITY_TOL:g})", file=sys.stderr)
        os._exit(3)

    out = None
    if rank == 0:
        # ---- per-kernel durations on this stream (HIP events around every launch) ----------
        # (the card idled through the CPU check above: half a second of the same steps first, so that the profiled launches run
        # at the clocks of the timed loop -- measured straight after the idle phase they come out 5-7 % longer than in the
        # rocprofv3 kernel trace of the same command)
        t_burst = time.perf_counter()
        while time.perf_counter() - t_burst < 0.5:
            for i in range(10):
                step(W + i)
            ws.sync()
        ws.profile(True)
        prof_steps = min(K, 10)
        for i in range(W, W + prof_steps):
            step(i)
        ws.sync()
        kinds = {"apply": K_APPLY, "sweep": K_SWEEP, "coef": K_COEF, "finalize": K_FINALIZE, "misc": K_MISC,
                 "sweep_tile_list": K_SWEEP_LIST, "apply_tile_list": K_APPLY_LIST, "project": K_PROJECT, "sweep_virtual": K_SWEEP_VIRTUAL,
                 "apply_virtual": K_APPLY_VIRTUAL}
        prof = {k: ws.profile_get(v) for k, v in kinds.items()}
        prof_log = ws.profile_log()
        ws.profile(False)
        stage_launches = stage_launch_table(ws, prof_log, prof_steps, N, B, sparse_lhs=(ncols == 1 and mps_targets is None) or (ncols == 1 and chi))
        sweep_launches, sweep_ms = prof["sweep"]
        apply_launches, apply_ms = prof["apply"]
        # algorithmic bytes: one gate group = read+write of each live vector
        sweep_bytes_per_step = 4 * 16 * N * G * B   # two vectors
        apply_bytes_per_step = 2 * 16 * N * G * B   # one vector
        sweep_avg_ms = sweep_ms / max(sweep_launches, 1)
        sweep_bytes_per_launch = sweep_bytes_per_step * prof_steps / max(sweep_launches, 1)
        achieved = sweep_bytes_per_launch / (sweep_avg_ms * 1e-3) / 1e9 if sweep_avg_ms > 0 else 0.0
        # useful fp64 work of the sweep (SURVEY 8d's unfused count per element: two operands 48 / block + 36 / front
        # qubit, inner products 32 + 24), spread over the sweep's launches of one step
        sweep_flops_per_step = float(N) * (80.0 * (G - n) + 60.0 * n) * B
        sweep_flops_per_launch = sweep_flops_per_step * prof_steps / max(sweep_launches, 1)
        sweep_tflops = sweep_flops_per_launch / (sweep_avg_ms * 1e-3) / 1e12 if sweep_avg_ms > 0 else 0.0
        mfma_tflops = None
        if ws.kernel_family(1) == 3 and sweep_ms > 0:
            mfma_tflops = 288.0 * N * B * ws.plan_substages(1) * prof_steps / (sweep_ms * 1e-3) / 1e12
        exec_tflops = mfma_tflops if mfma_tflops is not None else sweep_tflops
        exec_flops_per_launch = (288.0 * N * B * ws.plan_substages(1) * prof_steps / max(sweep_launches, 1)) if mfma_tflops is not None else sweep_flops_per_launch
        dominant = None
        if stage_launches:   # matrix-core path: every stage launch on its own (a sweep over a tile list is a different launch from a dense one)
            dominant = max(stage_launches["launches"], key=lambda r: r["avg_ms"])
            exec_tflops, exec_flops_per_launch, sweep_avg_ms = dominant["TFLOPs"], dominant["flops"], dominant["avg_ms"]
        stages_inv, k_inv, tiles_inv = ws.plan_info(0)
        stages_sw, k_sw, tiles_sw = ws.plan_info(1)

        # ---- single-evaluation latency (batch 1, host-visible result each call) ---------------
        latency = None
        latency_rounds = None
        if full and not args.no_latency and ncols == 1 and not chi:
            ws1 = Workspace(ctx, batch=1, device=local_rank)
            ws1.upload(BUF_Y, targets[0])
            ws1.set_basis(BUF_X, 0)
            ws1.gather_setup(flip_idx)
            ths = np.pi * (2 * rng.random((60, T)) - 1)
            for i in range(10):  # thetas from host, amplitudes + gradient back to host: one native call
                ws1.eval(ths[i], vdag=True, gather=True, grad=True)
            time.sleep(1.0)      # same one-off start-up stall as above
            rounds = []          # best of 6 rounds of 50 evaluations (all rounds are reported)
            for r in range(6):
                t1 = time.perf_counter()
                for i in range(10, 60):
                    ws1.eval(ths[i], vdag=True, gather=True, grad=True)
                rounds.append((time.perf_counter() - t1) / 50 * 1e3)
            latency = min(rounds)
            latency_rounds = rounds
            ws1.close()

        # ---- the metric as SURVEY 8(d) words it: objective(theta) + gradient(theta) on the objective OBJECT ----------------
        objective_object = None
        if full and not args.no_objective_object and ncols == 1 and not chi:
            objective_object = objective_object_rates(circ, targets, rng, local_rank)

        # measured HBM traffic of the dominant kernel (rocprofv3 --pmc passes, tools/pmc_summary.py), if the
        # committed profile was taken on this workload / batch
        traffic, traffic_source = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pj.get("workload") == workload and pj.get("batch_per_gpu") == B:
                want = dominant["kernel"] if dominant else "sweep_mfma_kernel"
                for kname, kv in pj["kernels"].items():
                    if want in kname or (not dominant and "sweep_stage_kernel" in kname):
                        traffic = kv["hbm_bytes_per_launch"]
                        traffic_source = ("NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_run5.py, "
                                          f"committed as profiles/pmc_traffic.json ({pj.get('date', 'undated')}, {pj.get('source', 'builder-run')})")
        except Exception:
            traffic, traffic_source = None, None
        # ---- the literal reference-signature calls, one lane: v_dagger_mul_mps + fast_dot_gradient with the workload's trunc_thr
        front_door = None
        if chi:
            from aqc_research_amd import mps_dot_objective as mdo
            from aqc_research_amd import mps_operations as mpo

            th1 = np.pi * (2 * rng.random((40, T)) - 1)
            for i in range(5):
                mdo.fast_dot_gradient(circ, th1[i], zero_mps, mpo.v_dagger_mul_mps(circ, th1[i], distinct[i % D], trunc_thr=thr), trunc_thr=thr)
            t1 = time.perf_counter()
            for i in range(5, 40):
                vh1 = mpo.v_dagger_mul_mps(circ, th1[i], distinct[i % D], trunc_thr=thr)
                g1 = mdo.fast_dot_gradient(circ, th1[i], zero_mps, vh1, trunc_thr=thr)
            dt1 = time.perf_counter() - t1
            a1 = orc.as_ansatz(circ)
            x1 = np.zeros(1 << n, complex)
            x1[0] = 1
            g_ref = orc.grad_of_dot_product(a1, th1[39], x1, orc.v_dagger_mul_vec(a1, th1[39], orc.mps_to_vector(distinct[39 % D])))
            front_door = {"calls": "mps_operations.v_dagger_mul_mps + mps_dot_objective.fast_dot_gradient, trunc_thr = %g" % thr,
                          "evals_per_s": 35 / dt1, "ms_per_eval": dt1 / 35 * 1e3, "route": "dense" if mdo.use_dense(n, thr) else "mps engine",
                          "parity_maxerr": float(np.abs(g1 - g_ref).max())}
        evals = K * B * n_gpus
        value = evals / wall
        out = {
            "metric": "objective+gradient evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": n_gpus,
            "steps": K,
            "warmup": W,
            "ms_per_step": wall / K * 1e3,
            "ms_per_eval": wall / (K * B) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": w["desc"],
                "n_qubits": n,
                "num_thetas": T,
                "gate_groups": G,
                "batch_per_gpu": B,
                "path": ("MPS front door (mps_dot_objective), dense route" if chi else "state-vector (core_operations)") if ncols == 1 else "matrix (core_op_matrix)",
                "mps_bond_dimension": chi or None,
                "mps_trunc_thr": thr if chi else None,
                "columns": ncols,
                "transport": comm.transport if comm_note is None else comm_note,
                "records_gathered": len(gathered),
                "ranks_seen": ranks_seen,
                "per_rank_evals_per_s": {"min": min(rank_rates), "max": max(rank_rates), "mean": float(np.mean(rank_rates)),
                                         "all": rank_rates, "clock": "HIP events on each rank's own stream over its timed steps"},
                "launcher": os.environ.get("AQC_BENCH_LAUNCHER", "external" if world > 1 else "none"),
                "tile_bits": {"vdag": k_inv, "sweep": k_sw},
                "launches_per_eval_step": {"vdag": stages_inv, "sweep": stages_sw},
            },
            # The sweep launches fuse many gate groups per HBM round trip, so what bounds them is fp64 arithmetic on the matrix
            # cores, not HBM.  `achieved` = flops the matrix pipe EXECUTED in one sweep launch (288 real flops per amplitude
            # and sub-stage: 9 real 16x16x16 products per 256 amplitudes) / its average duration (HIP events on the
            # workspace stream), against the 78.6 TFLOP/s fp64 MFMA peak: a true bound, frac <= 1 on every workload.
            # SURVEY 8(d)'s gate-by-gate flop count (what the reference's algorithm would spend) rides along as
            # `algorithmic_*`; it is NOT a bound for a kernel that fuses whole gate groups into one 16 x 16 unitary.
            "roofline": {
                "bound": dominant.get("bound", "mfma") if dominant else "mfma",
                "kernel": dominant["kernel"] if dominant else ws.sweep_kernel_name(),
                "launch": None if not dominant else (dominant["plan"] if dominant["stage"] is None or dominant["tiles_frac"] is None else
                                                     f"{dominant['plan']} stage {dominant['stage']} ({dominant['substages']} sub-stages, "
                                                     f"{dominant['tiles_frac']:.4g} of the tiles)"),
                "step": None if not stage_launches else {k: v for k, v in stage_launches.items() if k != "launches"},
                "launches": None if not stage_launches else stage_launches["launches"],
                # (a memory-bound dominant launch -- the projected route's pass over z -- is priced in bytes)
                "achieved": dominant["GBps"] if dominant and dominant.get("bound") == "hbm" else exec_tflops,
                "peak": HBM_PEAK_GBS if dominant and dominant.get("bound") == "hbm" else FP64_PEAK_TFLOPS,
                "unit": "GB/s" if dominant and dominant.get("bound") == "hbm" else "TFLOP/s",
                "frac": dominant["frac_of_hbm_peak"] if dominant and dominant.get("bound") == "hbm" else exec_tflops / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "avg_launch_ms": sweep_avg_ms,
                "flops_per_launch": exec_flops_per_launch,
                "flop_convention": "executed MFMA flops (288 per amplitude and sub-stage)" if mfma_tflops is not None else
                                   "SURVEY 8d gate-by-gate flops (VALU kernel family: no fused unitaries)",
                "algorithmic_TFLOPs": sweep_tflops,
                "algorithmic_frac": sweep_tflops / FP64_PEAK_TFLOPS,
                "algorithmic_note": "SURVEY 8d unfused flop count / launch time; not a bound (exceeds 1 when one unitary absorbs many gates)",
                "substages": {"vdag": ws.plan_substages(0), "sweep": ws.plan_substages(1)},
            },
            # byte view of the same launches (information only: SURVEY 8d's per-gate-group byte model is not a lower
            # bound for a fused launch; `traffic_bytes_per_launch` is the rocprofv3 PMC measurement)
            "hbm": {
                "traffic_bytes_per_launch": traffic,
                "achieved_GBps": (traffic / (sweep_avg_ms * 1e-3) / 1e9) if traffic and sweep_avg_ms > 0 else None,
                "frac_of_8TBps": (traffic / (sweep_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and sweep_avg_ms > 0 else None,
                "algorithmic_bytes_per_launch": sweep_bytes_per_launch,
                "algorithmic_GBps": achieved,
            },
            "parity_maxerr": parity,
            "parity_lanes_checked": len(check_lanes) if ncols == 1 else 1,
            "debug_skip": os.environ.get("AQC_DEBUG_SKIP", ""),
            # The fused launches are bound by the fp64 vector ALU, not by HBM: SURVEY 8(d)'s unfused flop count
            # (V^H 28 / block + 14 / front qubit, sweep 48 + 36, inner products 32 + 24 per element) against
            # the 78.6 TFLOP/s fp64 peak of MI355X_MICROARCH.md, over the whole evaluation (wall clock).
            "fp64_whole_eval": {
                "achieved": N * (108.0 * (G - n) + 74.0 * n) * value / n_gpus / 1e12,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": N * (108.0 * (G - n) + 74.0 * n) * value / n_gpus / 1e12 / FP64_PEAK_TFLOPS,
                "flops_per_eval": N * (108.0 * (G - n) + 74.0 * n),
            },
            "sustained_evals_per_s": None if sustained is None else sustained[0],
            "sustained": None if sustained is None else {"seconds": sustained[1], "steps": sustained[2]},
            "results_to_host": "every timed step ends with asynchronous copies of its gradients (batch x T complex128) and amplitudes "
                               "into pinned host memory (aqc_ws_results_async: a second stream, overlapped with the next step's kernels; the "
                               "kernels that overwrite the results wait for the copies); the last step's copies are the checked ones",
            "objective_object": objective_object,
            "kernel_ms_per_step": {k: v[1] / prof_steps for k, v in prof.items()},
            "device_ms_per_step_events": ev_ms / K,
            "latency_batch1_ms": latency,
            "latency_batch1_rounds_ms": latency_rounds,
            "algorithmic_GBps_whole_eval": (sweep_bytes_per_step + apply_bytes_per_step) * K / wall / 1e9,
        }
        if front_door is not None:
            out["front_door_single_lane"] = front_door
        if full and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(circ, ncols)
            out["cpu_baseline"]["reference_numpy"] = reference_numpy_record(workload)
    ws.close()
    comm.barrier()
    return out


if __name__ == "__main__":
    main()
