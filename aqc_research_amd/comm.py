"""
Process-group layer of the job-sharded path: one process per GPU, fixed-size float64 records, two collectives
(all-gather of result records, all-reduce of small sums).  The production transport is ``aqc_comm_*`` of the C ABI --
librccl (RCCL over xGMI) bound directly, no torch.  ``GlooDouble`` is the CPU test double used by the world_size-2
tests in this repository and by rehearsals on boxes with fewer GPUs than ranks; it speaks the same interface over
``torch.distributed`` with the gloo backend.

Rendezvous ("unique id passed via file/env, no MPI"): ranks are told their place by the environment that
``torch.distributed.run`` / any launcher sets (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_PORT); rank 0 writes the 128-byte
RCCL unique id to a file, the others poll for it.
"""
import ctypes
import os
import tempfile
import time
from typing import Optional

import numpy as np

from . import _lib

__all__ = ["Communicator", "RcclCommunicator", "GlooDouble", "from_environment"]


class Communicator:
    """Interface: rank, size, allgather(f64[count]) -> f64[size, count], allreduce(f64[n], op) in place, barrier."""

    rank = 0
    size = 1
    transport = "none"

    def allgather(self, send: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(send, dtype=np.float64).reshape(1, -1).copy()

    def allreduce(self, data: np.ndarray, op: str = "sum") -> np.ndarray:
        return data

    def barrier(self) -> None:
        pass

    def close(self) -> None:
        pass


class RcclCommunicator(Communicator):
    """``aqc_comm_*``: RCCL bound directly through the C ABI (include/aqc_hip.h)."""

    transport = "rccl (aqc_comm)"

    def __init__(self, rank: int, size: int, device: int, id_file: str, timeout: float = 120.0):
        L = _lib.lib()
        self._L, self.rank, self.size, self.device = L, int(rank), int(size), int(device)
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(L.aqc_comm_unique_id(buf))
            tmp = id_file + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(buf.raw)
            os.replace(tmp, id_file)   # atomic: readers never see a partial id
            uid = buf.raw
        else:
            t0 = time.time()

            def fresh():   # a complete id written during THIS launch (a crashed run may have left one behind)
                try:
                    st = os.stat(id_file)
                except OSError:
                    return False
                return st.st_size == 128 and st.st_mtime >= t0 - 600.0

            while not fresh():
                if time.time() - t0 > timeout:
                    raise RuntimeError(f"aqc_comm: rank {rank} timed out waiting for the unique id in {id_file}")
                time.sleep(0.02)
            with open(id_file, "rb") as f:
                uid = f.read()
        handle = ctypes.c_void_p()
        _lib.check(L.aqc_comm_create(uid, size, rank, device, ctypes.byref(handle)))
        self.handle = handle
        self._id_file = id_file

    def allgather(self, send: np.ndarray) -> np.ndarray:
        s = np.ascontiguousarray(send, dtype=np.float64).ravel()
        out = np.empty((self.size, s.size), dtype=np.float64)
        _lib.check(self._L.aqc_comm_allgather(self.handle, _lib.dptr(s), _lib.dptr(out), s.size))
        return out

    def allreduce(self, data: np.ndarray, op: str = "sum") -> np.ndarray:
        if not (isinstance(data, np.ndarray) and data.dtype == np.float64 and data.flags.c_contiguous):
            raise ValueError("allreduce works in place on a C-contiguous float64 array")
        _lib.check(self._L.aqc_comm_allreduce(self.handle, _lib.dptr(data), data.size, {"sum": 0, "max": 2}[op]))
        return data

    def barrier(self) -> None:
        _lib.check(self._L.aqc_comm_barrier(self.handle))

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._L.aqc_comm_destroy(self.handle)
            self.handle = None
            if self.rank == 0:
                try:
                    os.remove(self._id_file)
                except OSError:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GlooDouble(Communicator):
    """CPU test double with the same interface over an already initialised torch.distributed group."""

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


_current: Optional[Communicator] = None


def _initialised_torch_group():
    import sys

    dist = sys.modules.get("torch.distributed")   # never import torch on behalf of the caller
    if dist is None:
        return None
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def from_environment(prefer: str = "auto") -> Communicator:
    """The process group of this run: an already initialised torch.distributed group is wrapped (that is how the CPU
    tests and rehearsals drive the path); otherwise, under a launcher (WORLD_SIZE > 1), RCCL is bound directly;
    a single process gets the trivial communicator.  The result is cached per process."""
    global _current
    if _current is not None:
        return _current
    dist = _initialised_torch_group()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if dist is not None and prefer != "rccl":
        _current = GlooDouble(dist)
    elif world > 1:
        rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        # all ranks of one launch share the launcher as parent process: port + parent pid name the launch
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('AQC_COMM_TAG', '0')}"
        id_file = os.environ.get("AQC_COMM_FILE", os.path.join(tempfile.gettempdir(), f"aqc_comm_id_{tag}"))
        _current = RcclCommunicator(rank, world, local % max(1, _lib.lib().aqc_device_count()), id_file)
    else:
        _current = Communicator()
    return _current


def reset() -> None:
    """Forget (and close) the cached communicator -- tests only."""
    global _current
    if _current is not None:
        _current.close()
    _current = None
