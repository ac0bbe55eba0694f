"""
Batch executor with the reference's ``run_jobs`` contract (job_executor.py:39-161), sharded
over GPUs instead of joblib processes: one process per GPU (torch.distributed.run), job j runs on
rank ``j % world_size`` with seed ``seed + 7*(j+1)`` (job_executor.py:64), and the only
communication is the final gather of the result records: ONE all-gather of fixed-size float64 records
{cost, fidelity, iteration counts, thetas[T_max]} over RCCL bound directly through the C ABI (``aqc_comm_*``; no
torch), or over the gloo test double on CPU.  Without a process group jobs run serially.
"""
import pickle
import sys
import traceback
from time import perf_counter
from typing import Callable, Dict, List

import numpy as np


def _job_function_wrapper(job_index: int, config: Dict, seed: int, job_function: Callable[[int, Dict], Dict]) -> Dict:
    seed = seed + 7 * (job_index + 1)
    try:
        if not isinstance(config, dict) or not callable(job_function):
            raise TypeError("config must be a dict and job_function callable")
        np.random.seed(seed)
        tic = perf_counter()
        result = job_function(job_index, config)
        result.update({"time": perf_counter() - tic, "status": "ok", "job_index": job_index, "seed": seed})
    except Exception:
        status = traceback.format_exc()
        # printed HERE, on the rank that ran the job: a fixed-size record of the final gather cannot carry the text
        print(f"exception in job={job_index}:\n{status}", flush=True)
        result = {"time": float(-1), "status": status, "job_index": job_index, "seed": seed}
    finally:
        sys.stderr.flush()
        sys.stdout.flush()
    return result


def local_device() -> int:
    """GPU of this rank (LOCAL_RANK under a one-process-per-GPU launcher, folded onto the visible devices), 0 otherwise."""
    from .engine import default_device

    return default_device()


# ---- result records ---------------------------------------------------------------------------------------------
# The final gather of run_jobs is the only inter-GPU traffic of the path: one fixed-size float64 record
# per job, {job_index, seed, ok, time, cost, fidelity, num_iters, num_fun_ev, num_grad_ev, T, thetas[T_max]}.
_FIXED = ("cost", "fidelity", "num_iters", "num_fun_ev", "num_grad_ev")
_BOOKKEEPING = ("time", "status", "job_index", "seed")
_HEAD = 4 + len(_FIXED) + 1


def _fits_fixed_schema(result: Dict) -> bool:
    if not result["status"].startswith("ok"):
        return True
    extra = set(result) - set(_FIXED) - set(_BOOKKEEPING) - {"thetas"}
    return not extra and "thetas" in result and np.ndim(result["thetas"]) == 1


def pack_record(result: Dict, t_max: int) -> np.ndarray:
    rec = np.full(_HEAD + t_max, np.nan)
    ok = result["status"].startswith("ok")
    rec[0:4] = (result["job_index"], result["seed"], 1.0 if ok else 0.0, result["time"])
    if ok:
        th = np.asarray(result["thetas"], dtype=np.float64).ravel()
        rec[4 : 4 + len(_FIXED)] = [float(result.get(k, np.nan)) for k in _FIXED]
        rec[_HEAD - 1] = th.size
        rec[_HEAD : _HEAD + th.size] = th
    return rec


def unpack_record(rec: np.ndarray) -> Dict:
    ok = rec[2] == 1.0
    out = {"job_index": int(rec[0]), "seed": int(rec[1]), "time": float(rec[3]),
           "status": "ok" if ok else "failed on its rank (the traceback was printed there)"}
    if ok:
        for k, v in zip(_FIXED, rec[4 : 4 + len(_FIXED)]):
            if not np.isnan(v):
                out[k] = int(v) if k.startswith("num_") else float(v)
        out["thetas"] = rec[_HEAD : _HEAD + int(rec[_HEAD - 1])].copy()
    return out


def _gather_records(local: List[Dict], comm, njobs: int, records: str) -> List[Dict]:
    """All ranks end with every job's result.  "fixed": one float64 record of the schema above per job (what the
    optimisation drivers produce); "pickle": arbitrary result dictionaries as padded byte payloads; "auto" picks
    "fixed" when every rank's results fit the schema.  Both travel as float64 words through ONE all-gather (plus a
    3-word header exchange), over RCCL (aqc_comm) or the gloo test double."""
    fits = all(_fits_fixed_schema(r) for r in local)
    t_max = max([np.size(r["thetas"]) for r in local if r["status"].startswith("ok") and "thetas" in r] + [0])
    payload = pickle.dumps(local) if records != "fixed" else b""
    head = comm.allgather(np.array([1.0 if fits else 0.0, float(t_max), float(len(payload))]))
    use_fixed = records == "fixed" or (records == "auto" and bool(np.all(head[:, 0] == 1.0)))
    per_rank = -(-njobs // comm.size)   # jobs of the busiest rank
    if use_fixed:
        t_all = int(head[:, 1].max())
        send = np.full((per_rank, _HEAD + t_all), np.nan)
        send[:, 0] = -1.0                 # unused slots
        for i, r in enumerate(local):
            send[i] = pack_record(r, t_all)
        got = comm.allgather(send.ravel()).reshape(comm.size * per_rank, _HEAD + t_all)
        return [unpack_record(rec) for rec in got if rec[0] >= 0]
    words = int(-(-int(head[:, 2].max()) // 8)) + 1
    buf = np.zeros(words * 8, dtype=np.uint8)
    buf[: len(payload)] = np.frombuffer(payload, dtype=np.uint8)
    got = comm.allgather(buf.view(np.float64)).view(np.uint8).reshape(comm.size, words * 8)
    out: List[Dict] = []
    for r in range(comm.size):
        out.extend(pickle.loads(got[r, : int(head[r, 2])].tobytes()))
    return out


def run_jobs(
    configs: List[Dict],
    seed: int,
    job_function: Callable[[int, Dict], Dict],
    *,
    tolerate_failure: bool = False,
    num_jobs: int = -1,
    records: str = "auto",
) -> List[Dict]:
    """Runs every configuration once; returns the list of result dicts ordered by job index,
    each augmented with ``time``, ``status``, ``job_index``, ``seed`` (job_executor.py:96-161).
    ``num_jobs`` is accepted for compatibility; parallelism comes from the process group (one process per GPU,
    job j on rank j % world).  ``records``: "fixed" | "pickle" | "auto" -- see ``_gather_records``."""
    if not (isinstance(configs, list) and len(configs) > 0 and isinstance(configs[0], dict)):
        raise ValueError("configs must be a non-empty list of dictionaries")
    if not callable(job_function):
        raise TypeError("job_function must be callable")
    if not (isinstance(num_jobs, int) and (num_jobs == -1 or num_jobs >= 1)):
        raise ValueError("num_jobs must be -1 or a positive integer")

    if records not in ("auto", "fixed", "pickle"):
        raise ValueError("records must be 'auto', 'fixed' or 'pickle'")
    from .comm import from_environment

    comm = from_environment()
    rank, world = comm.rank, comm.size
    results = [_job_function_wrapper(i, c, seed, job_function) for i, c in enumerate(configs) if i % world == rank]
    if world > 1:
        results = _gather_records(results, comm, len(configs), records)
    results.sort(key=lambda r: r["job_index"])

    print("")
    for r in results:
        if not r["status"].startswith("ok") and rank == 0:
            print(f"Simulation {r['job_index']} failed:\n\n{r['status']}\n{'-' * 80}\n\n")
    if not any(r["status"].startswith("ok") for r in results):
        raise RuntimeError("there is no valid simulation results")
    if tolerate_failure:
        results = [r for r in results if r["status"].startswith("ok")]
    return results
