"""
Batch executor with the reference's ``run_jobs`` contract (job_executor.py:39-161), sharded
over GPUs instead of joblib processes: one process per GPU (torch.distributed.run), job j runs on
rank ``j % world_size`` with seed ``seed + 7*(j+1)`` (job_executor.py:64), and the only
communication is the final gather of the result records (all_gather of padded byte records:
RCCL over xGMI with the nccl backend, gloo on CPU).  Without a process group jobs run serially.
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
        print(f"exception in job={job_index},\n", flush=True)
        result = {"time": float(-1), "status": traceback.format_exc(), "job_index": job_index, "seed": seed}
    finally:
        sys.stderr.flush()
        sys.stdout.flush()
    return result


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # torch is optional plumbing
        return None
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def local_device() -> int:
    """GPU of this rank (LOCAL_RANK under torch.distributed.run), 0 otherwise."""
    import os

    return int(os.environ.get("LOCAL_RANK", "0"))


def _gather_records(local: List[Dict], dist) -> List[Dict]:
    """all_gather of fixed-size (padded) records; returns every rank's results on every rank."""
    import torch

    device = torch.device("cuda", local_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    payload = pickle.dumps(local)
    size = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(size) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, size)
    cap = int(max(int(s.item()) for s in sizes))
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, buf)
    out: List[Dict] = []
    for b, s in zip(bufs, sizes):
        out.extend(pickle.loads(bytes(b[: int(s.item())].cpu().numpy())))
    return out


def run_jobs(
    configs: List[Dict],
    seed: int,
    job_function: Callable[[int, Dict], Dict],
    *,
    tolerate_failure: bool = False,
    num_jobs: int = -1,
) -> List[Dict]:
    """Runs every configuration once; returns the list of result dicts ordered by job index,
    each augmented with ``time``, ``status``, ``job_index``, ``seed`` (job_executor.py:96-161).
    ``num_jobs`` is accepted for compatibility; parallelism comes from the process group."""
    if not (isinstance(configs, list) and len(configs) > 0 and isinstance(configs[0], dict)):
        raise ValueError("configs must be a non-empty list of dictionaries")
    if not callable(job_function):
        raise TypeError("job_function must be callable")
    if not (isinstance(num_jobs, int) and (num_jobs == -1 or num_jobs >= 1)):
        raise ValueError("num_jobs must be -1 or a positive integer")

    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
    results = [_job_function_wrapper(i, c, seed, job_function) for i, c in enumerate(configs) if i % world == rank]
    if dist:
        results = _gather_records(results, dist)
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
