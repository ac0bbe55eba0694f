"""
Process-group layer of the job-sharded path: one process per GPU, fixed-size float64 records, two collectives
(all-gather of result records, all-reduce of small sums).  The production transport is ``aqc_comm_*`` of the C ABI --
librccl (RCCL over xGMI) bound directly, no torch.  (The CPU tests and rehearsals of this repository install a test
double with the same interface over ``torch.distributed``/gloo through ``use()``: ``tests/gloo_double.py``.)

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

__all__ = ["Communicator", "RcclCommunicator", "from_environment", "use", "reset"]


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

    def __init__(self, rank: int, size: int, device: int, id_file: str, timeout: float = 120.0, tag: Optional[str] = None):
        """``tag`` names the launch (default ``launch_tag()``): rank 0 writes it behind the 128 id bytes and the other ranks
        only accept a file that carries THEIR tag, so an id left behind by a crashed launch that used the same file name
        is never taken for this launch's (it made ``ncclCommInitRank`` hang instead of fail)."""
        L = _lib.lib()
        self._L, self.rank, self.size, self.device = L, int(rank), int(size), int(device)
        nonce = (launch_tag() if tag is None else str(tag)).encode()
        if rank == 0:
            try:
                os.remove(id_file)   # a crashed earlier launch may have left one behind
            except OSError:
                pass
            buf = ctypes.create_string_buffer(128)
            _lib.check(L.aqc_comm_unique_id(buf))
            tmp = id_file + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(buf.raw + nonce)
            os.replace(tmp, id_file)   # atomic: readers never see a partial id
            uid = buf.raw
        else:
            t0 = time.time()

            def read_id():   # the id of THIS launch: 128 bytes followed by this launch's tag (files are replaced atomically)
                try:
                    with open(id_file, "rb") as f:
                        raw = f.read()
                except OSError:
                    return None
                return raw[:128] if len(raw) >= 128 and raw[128:] == nonce else None

            uid = read_id()
            while uid is None:
                if time.time() - t0 > timeout:
                    raise RuntimeError(f"aqc_comm: rank {rank} timed out waiting for the unique id of launch {nonce.decode()!r} in {id_file}")
                time.sleep(0.02)
                uid = read_id()
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


_current: Optional[Communicator] = None


def use(communicator: Optional[Communicator]) -> None:
    """Install the process group of this run (tests and rehearsals hand in their double; None forgets it)."""
    global _current
    _current = communicator


def launch_tag() -> str:
    """Names one launch from values EVERY rank of it sees alike, on any node: the rendezvous address and port, the launcher's run
    id / restart count (a restarted attempt of an elastic launcher gets a new name), ``AQC_COMM_TAG`` (``bench.launch_ranks``
    sets a fresh one per launch).  This is the nonce written behind the id bytes: with ``AQC_COMM_FILE`` on a shared file
    system the ranks of other nodes must arrive at the same string (the parent pid, which differs from node to node, is only
    part of the DEFAULT file name, which is node-local anyway)."""
    env = os.environ
    return "_".join([env.get("MASTER_ADDR", "local"), env.get("MASTER_PORT", "0"), env.get("AQC_COMM_TAG", "0"),
                     env.get("TORCHELASTIC_RUN_ID", "0"), env.get("TORCHELASTIC_RESTART_COUNT", "0")])


def from_environment() -> Communicator:
    """The process group of this run: whatever ``use()`` installed; otherwise, under a launcher (WORLD_SIZE > 1), RCCL
    bound directly; a single process gets the trivial communicator.  The result is cached per process."""
    global _current
    if _current is not None:
        return _current
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        id_file = os.environ.get("AQC_COMM_FILE", os.path.join(tempfile.gettempdir(), f"aqc_comm_id_{launch_tag()}_{os.getppid()}"))
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
