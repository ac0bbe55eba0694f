"""
CPU test double of the job-sharded path's process group (aqc_research_amd.comm.Communicator): the same interface over an
already initialised ``torch.distributed`` group (gloo here; the rehearsal mode of bench.py may hand in any backend).
Test infrastructure: the product package binds librccl directly and never imports torch.
"""
import os

import numpy as np

from aqc_research_amd import comm as aqc_comm


class GlooDouble(aqc_comm.Communicator):
    def __init__(self, dist):
        self._dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.transport = f"torch.distributed ({dist.get_backend()})"

    def _device(self):
        import torch

        if self._dist.get_backend() == "nccl":
            return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        return torch.device("cpu")

    def allgather(self, send: np.ndarray) -> np.ndarray:
        import torch

        s = torch.from_numpy(np.ascontiguousarray(send, dtype=np.float64).ravel().copy()).to(self._device())
        outs = [torch.empty_like(s) for _ in range(self.size)]
        self._dist.all_gather(outs, s)
        return np.stack([o.cpu().numpy() for o in outs])

    def allreduce(self, data: np.ndarray, op: str = "sum") -> np.ndarray:
        import torch

        t = torch.from_numpy(data).to(self._device())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM if op == "sum" else self._dist.ReduceOp.MAX)
        np.copyto(data, t.cpu().numpy())
        return data

    def barrier(self) -> None:
        self._dist.barrier()


def install(dist) -> GlooDouble:
    """Wrap the initialised group and make it the process group of the package (comm.use)."""
    double = GlooDouble(dist)
    aqc_comm.use(double)
    return double
