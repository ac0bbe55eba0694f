#!/usr/bin/env python3
"""
Baseline of record: the REFERENCE itself (qiskit-community/aqc-research v0.1.0, imported read-only from
/root/reference exactly as tests/golden/make_golden.py does) timed on this container's host cores, on every
configuration of BASELINE.json:

  * one evaluation = v_dagger_mul_vec + grad_of_dot_product (state vector), or
    v_dagger_mul_mat + grad_of_matrix_dot_product on X = I, Y = U (matrix);
    coordinate descent: one coord_descent_single_sweep;
  * 5 warm-ups, then >= 20 timed repetitions (a time cap trims the slowest configurations, never below 20);
    median, p10 and p90 of the per-evaluation time;
  * (a) one process, BLAS threads = all cores; (b) P single-threaded processes at once -- how the reference itself
    uses cores (job_executor.py:141) -- aggregate rate = P / median.

Runs in the BUILD container only: the reference never travels to the GPU box.  Writes profiles/ref_baseline.json (the table
BASELINE.md section 2 quotes and bench.py prints as cpu_baseline.reference_numpy).

Usage:  python tools/ref_baseline.py [--reps 20] [--procs 8] [--only NAME,...]
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("AQC_REFERENCE", "/root/reference")

CONFIGS = [
    # name, BASELINE.json config, kind, parameters
    ("mat5_cyc180", "cfg 1: 5-qubit AQC, cyclic_spin L=180 (docs/aqc.ipynb ansatz)", "matrix", dict(n=5, layout="cyclic_spin", blocks=180)),
    ("cd5_cyc180", "cfg 1, coordinate descent: one coord_descent_single_sweep", "cd", dict(n=5, layout="cyclic_spin", blocks=180)),
    ("sv12_trotter2", "cfg 2: 12-qubit ASP, 2nd-order Trotter, 2 layers (first horizon)", "trotter", dict(n=12, layers=2)),
    ("sv12_trotter12", "cfg 2: 12-qubit ASP, 2nd-order Trotter, 12 layers (last horizon)", "trotter", dict(n=12, layers=12)),
    ("sv16_l40", "cfg 3 geometry on the dense path: 16 qubits, 40 blocks (the headline)", "vector", dict(n=16, layout="spin", blocks=40)),
    ("sv20_l40", "cfg 4 size: 20 qubits, 40 blocks", "vector", dict(n=20, layout="spin", blocks=40)),
    ("sv20_trotter2", "cfg 4 job: 20-qubit 2nd-order Trotter, 2 layers", "trotter", dict(n=20, layers=2)),
    ("mat10_l40", "cfg 5: 10-qubit full unitary, d=1024, spin L=40", "matrix", dict(n=10, layout="spin", blocks=40)),
]


def _import_reference():
    """The reference's NumPy modules; qiskit names are empty placeholders that are never executed."""
    import numpy as np

    sys.dont_write_bytecode = True
    np.cfloat = np.complex128  # NumPy >= 2 removed the alias the reference uses

    class _Missing:
        def __init__(self, *a, **k):
            raise RuntimeError("qiskit is not available in this container")

    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    placeholder("qiskit", QuantumCircuit=_Missing)
    placeholder("qiskit.quantum_info", Operator=_Missing, Statevector=_Missing)
    placeholder("qiskit.circuit")
    placeholder("qiskit.circuit.library", QFT=_Missing)
    placeholder("qiskit_aer", AerSimulator=_Missing)
    sys.modules["qiskit"].quantum_info = sys.modules["qiskit.quantum_info"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import aqc_research.core_op_matrix as com
    import aqc_research.core_operations as cop
    from aqc_research.circuit_structures import create_ansatz_structure, make_trotter_like_circuit
    from aqc_research.parametric_circuit import ParametricCircuit, TrotterAnsatz

    return types.SimpleNamespace(np=np, cop=cop, com=com, create_ansatz_structure=create_ansatz_structure,
                                 make_trotter_like_circuit=make_trotter_like_circuit, ParametricCircuit=ParametricCircuit,
                                 TrotterAnsatz=TrotterAnsatz)


def make_eval(R, kind, p):
    """A closure that runs ONE evaluation of the reference on fresh random thetas (seed 1234: BASELINE.md section 2)."""
    np = R.np
    rng = np.random.default_rng(1234)
    n = p["n"]
    d = 1 << n
    if kind == "trotter":
        circ = R.TrotterAnsatz(n, R.make_trotter_like_circuit(n, p["layers"]), second_order=True)
    else:
        circ = R.ParametricCircuit(n, "cx", R.create_ansatz_structure(n, p["layout"], "full", p["blocks"]))
    T = circ.num_thetas

    def thetas():
        return np.pi * (2.0 * rng.random(T) - 1.0)

    if kind in ("vector", "trotter"):
        y = rng.standard_normal(d) + 1j * rng.standard_normal(d)
        y /= np.linalg.norm(y)
        x = np.zeros(d, dtype=np.complex128)
        x[0] = 1.0
        vhy = np.zeros(d, dtype=np.complex128)
        wsp = np.zeros((4, d), dtype=np.complex128)

        def one():
            th = thetas()
            R.cop.v_dagger_mul_vec(circ, th, y, vhy, wsp)
            R.cop.grad_of_dot_product(circ, th, x, vhy, wsp)

    elif kind == "matrix":
        u = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
        wsp = np.zeros((4, d, d), dtype=np.complex128)

        def one():
            th = thetas()
            vhy = R.com.v_dagger_mul_mat(circ, th, u.copy(), wsp)
            R.com.grad_of_matrix_dot_product(circ, th, np.eye(d, dtype=np.complex128), vhy, wsp)

    elif kind == "cd":
        u = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))[0]
        wsp = np.zeros((4, d, d), dtype=np.complex128)
        th = thetas()

        def one():
            R.com.coord_descent_single_sweep(circ, th, u, wsp)

    else:
        raise ValueError(kind)
    return one, T, circ.num_blocks


def time_config(name, reps, warm, cap_s, out=None, barrier=None):
    R = _import_reference()
    cfg = next(c for c in CONFIGS if c[0] == name)
    one, T, L = make_eval(R, cfg[2], cfg[3])
    for _ in range(warm):
        one()
    if barrier is not None:
        barrier.wait()
    times = []
    t_all = time.perf_counter()
    while len(times) < reps or (len(times) < 4 * reps and time.perf_counter() - t_all < cap_s):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    res = dict(times=times, T=int(T), L=int(L))
    if out is not None:
        out.put(res)
    return res


def stats(times):
    import numpy as np

    t = np.sort(np.asarray(times))
    return dict(reps=int(t.size), median_ms=float(np.median(t) * 1e3), p10_ms=float(np.percentile(t, 10) * 1e3),
                p90_ms=float(np.percentile(t, 90) * 1e3))


def _worker(name, reps, warm, cap_s, out, barrier):
    time_config(name, reps, warm, cap_s, out, barrier)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=5)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--cap-seconds", type=float, default=20.0, help="keep repeating (up to 4x reps) while under this many seconds")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "ref_baseline.json"))
    args = ap.parse_args()
    names = [c[0] for c in CONFIGS if not args.only or c[0] in args.only.split(",")]

    cpu = "unknown"
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    import numpy as np

    table = {}
    ctx = mp.get_context("spawn")
    for name in names:
        cfg = next(c for c in CONFIGS if c[0] == name)
        # (a) one process, BLAS threads = all cores (set before NumPy loads in the child)
        os.environ["OPENBLAS_NUM_THREADS"] = str(args.procs)
        os.environ["OMP_NUM_THREADS"] = str(args.procs)
        q = ctx.Queue()
        pr = ctx.Process(target=_worker, args=(name, args.reps, args.warm, args.cap_seconds, q, None))
        pr.start()
        single = q.get()
        pr.join()
        # (b) P single-threaded processes at the same time
        os.environ["OPENBLAS_NUM_THREADS"] = "1"
        os.environ["OMP_NUM_THREADS"] = "1"
        q = ctx.Queue()
        bar = ctx.Barrier(args.procs)
        prs = [ctx.Process(target=_worker, args=(name, args.reps, args.warm, args.cap_seconds, q, bar)) for _ in range(args.procs)]
        for pr in prs:
            pr.start()
        multi = [q.get() for _ in prs]
        for pr in prs:
            pr.join()
        s1 = stats(single["times"])
        sm = stats([t for m in multi for t in m["times"]])
        unit = "sweeps/s" if cfg[2] == "cd" else "evals/s"
        table[name] = dict(config=cfg[1], kind=cfg[2], num_thetas=single["T"], num_blocks=single["L"], unit=unit,
                           single_process=dict(s1, blas_threads=args.procs, rate=1e3 / s1["median_ms"]),
                           parallel_processes=dict(sm, processes=args.procs, blas_threads=1, rate=args.procs * 1e3 / sm["median_ms"]))
        print(f"{name:16s} 1 proc: {s1['median_ms']:9.2f} ms ({1e3 / s1['median_ms']:8.2f}/s)   {args.procs} procs: "
              f"{sm['median_ms']:9.2f} ms each ({args.procs * 1e3 / sm['median_ms']:8.2f}/s aggregate)", flush=True)
    rec = dict(what="the reference's own NumPy path (qiskit-community/aqc-research v0.1.0, /root/reference) timed by tools/ref_baseline.py",
               where="build container (the reference never travels to the GPU box)",
               host=dict(cpu_model=cpu, os_cpu_count=os.cpu_count(), python=platform.python_version(), numpy=np.__version__),
               date=time.strftime("%Y-%m-%d"), warmups=args.warm, min_reps=args.reps, configs=table)
    if os.path.exists(args.out) and args.only:   # partial refresh keeps the other rows
        old = json.load(open(args.out))
        old["configs"].update(table)
        rec["configs"] = old["configs"]
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
