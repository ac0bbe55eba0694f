"""
Lockstep batching of independent optimisations on one GPU.

The reference parallelises seeds x horizons over CPU processes, one objective per process
(job_executor.py:136-143).  On the GPU one evaluation at a time leaves the device mostly idle
(one 16-qubit evaluation is ~0.2 ms, 64 of them in one batched launch sequence are ~0.9 ms), so
jobs that share an ansatz are run as *lanes* of one B-lane workspace: every job keeps its own
optimizer, objective object, stoppers and state machine -- unchanged host code, one thread per job
-- but its native calls go through a ``LaneView``.  A view does not launch anything itself: it
files a request and blocks; when every running job has filed one, the last arrival executes the
requests of the round as batched native calls (one per distinct call signature) and wakes the
others.  Results are those of the single-lane calls (each lane sees only its own thetas, target and
lhs state); only the order of floating-point operations inside the kernels may differ, because the
batched workspace may pick the throughput kernel family.
"""
import os
import threading
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .engine import BUF_X, BUF_X2, HipContext, Workspace

__all__ = ["LockstepBatch", "LaneView", "run_jobs_lockstep"]


class _Request:
    __slots__ = ("sig", "thetas", "weight", "max_no")

    def __init__(self, sig, thetas, weight=1.0, max_no=0):
        self.sig, self.thetas, self.weight, self.max_no = sig, thetas, weight, max_no


class LaneView:
    """The subset of the ``Workspace`` interface the state-preparation objectives use, bound to one
    lane of a ``LockstepBatch``.  Shapes are those of a one-lane workspace."""

    batch = 1
    ncols = 1

    def __init__(self, owner: "LockstepBatch", lane: int):
        self._owner, self.lane = owner, lane
        self.T, self.dim, self.ctx, self.device = owner.ws.T, owner.ws.dim, owner.ws.ctx, owner.ws.device
        self._mps: Dict[int, Any] = {}

    # -- data movement (immediate, serialised with the batched calls) ---------------------------
    def upload(self, buf: int, data, lane: Optional[int] = None) -> None:
        with self._owner._cv:
            self._owner.ws.upload(buf, data, lane=self.lane)

    def download(self, buf: int, lane: Optional[int] = None) -> np.ndarray:
        with self._owner._cv:
            return self._owner.ws.download(buf, lane=self.lane)[None, :]

    def set_basis(self, buf: int, index) -> None:
        with self._owner._cv:
            self._owner._combo_idx[buf][self.lane] = (int(np.asarray(index).ravel()[0]), -1)
            self._owner._combo_coef[buf][self.lane] = (1.0, 0.0)
            self._owner._basis_dirty[buf] = True

    def set_combo(self, buf: int, index, coef) -> None:
        with self._owner._cv:
            self._owner._combo_idx[buf][self.lane] = np.asarray(index, dtype=np.int64).ravel()[:2]
            self._owner._combo_coef[buf][self.lane] = np.asarray(coef, dtype=np.complex128).ravel()[:2]
            self._owner._basis_dirty[buf] = True

    def gather_setup(self, index) -> None:
        idx = np.ascontiguousarray(index, dtype=np.int64).ravel()
        with self._owner._cv:
            if self._owner._gather_idx is None:
                self._owner._gather_idx = idx.copy()
                self._owner.ws.gather_setup(idx)
            elif not np.array_equal(self._owner._gather_idx, idx):
                raise ValueError("all lanes of a lockstep batch must gather the same amplitudes")

    def mps_upload(self, slot: int, mps) -> None:
        self._mps[slot] = mps

    def mps_to_vec(self, slot: int, buf: int, lane: int = 0) -> None:
        with self._owner._cv:   # upload + contraction are one atomic step: the MPS slots are shared
            self._owner.ws.mps_upload(slot, self._mps[slot])
            self._owner.ws.mps_to_vec(slot, buf, self.lane)

    # -- compute (deferred to the round) --------------------------------------------------------
    def eval(self, thetas=None, vdag: bool = True, gather: bool = False, grad: bool = True, x_buf: int = BUF_X,
             block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True):
        th = None
        if thetas is not None:
            th = np.array(thetas, dtype=np.float64).ravel()
            if th.size != self.T:
                raise ValueError(f"expected {self.T} thetas, got {th.size}")
        br = None if block_range is None else (int(block_range[0]), int(block_range[1]))
        sig = (bool(vdag), bool(gather), bool(grad), int(x_buf), br, bool(front_layer))
        return self._owner._submit(self.lane, _Request(sig, th))

    # A lane's objective()/gradient() pair as ONE request (Workspace.surrogate_eval): with it every lane of a round files the
    # same kind of request whatever state leads, so a round is one native call.
    prefers_surrogate_eval = os.environ.get("AQC_LOCKSTEP_SURROGATE_EVAL", "1") != "0"   # 0: the per-call requests of round 2 (cross-check)

    def surrogate_eval(self, thetas, weight: np.ndarray, max_no: np.ndarray, update_state=True,
                       block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True):
        th = np.array(thetas, dtype=np.float64).ravel()
        if th.size != self.T:
            raise ValueError(f"expected {self.T} thetas, got {th.size}")
        br = None if block_range is None else (int(block_range[0]), int(block_range[1]))
        sig = ("sur", int(update_state), br, bool(front_layer))
        f, fid, hs, gc, w_new, mx_new = self._owner._submit(self.lane, _Request(sig, th, float(weight[0]), int(max_no[0])))
        if update_state:
            weight[0], max_no[0] = w_new, mx_new
        return f, (fid if update_state else None), hs, gc

    def close(self) -> None:
        pass

    def __getattr__(self, name):  # anything else of Workspace is a whole-batch operation
        raise AttributeError(f"Workspace.{name} is not available through a lockstep lane")


class LockstepBatch:
    """``nlanes`` independent clients of one ``nlanes``-lane workspace, served round by round."""

    def __init__(self, circ, nlanes: int, device: Optional[int] = None):
        if nlanes < 1:
            raise ValueError("nlanes must be positive")
        self.ws = Workspace(HipContext.of(circ), batch=int(nlanes), ncols=1, device=device)
        self.nlanes = int(nlanes)
        self._cv = threading.Condition(threading.RLock())
        self._pending: Dict[int, _Request] = {}
        self._results: Dict[int, Any] = {}
        self._active = 0
        self._thetas = np.zeros((self.nlanes, self.ws.T))
        # lhs state of every lane as a two-term combination of basis states (one term: second index -1)
        self._combo_idx = {b: np.tile(np.array([0, -1], dtype=np.int64), (self.nlanes, 1)) for b in (BUF_X, BUF_X2)}
        self._combo_coef = {b: np.tile(np.array([1.0, 0.0], dtype=np.complex128), (self.nlanes, 1)) for b in (BUF_X, BUF_X2)}
        self._basis_dirty = {BUF_X: True, BUF_X2: True}
        self._gather_idx: Optional[np.ndarray] = None
        self._error: Optional[BaseException] = None
        self.rounds = 0
        self.native_calls = 0

    def lane(self, i: int) -> LaneView:
        if not 0 <= i < self.nlanes:
            raise IndexError("lane out of range")
        return LaneView(self, i)

    def close(self) -> None:
        self.ws.close()

    # -- the round ------------------------------------------------------------------------------
    def _submit(self, lane: int, req: _Request):
        with self._cv:
            if self._error is not None:
                raise RuntimeError("lockstep batch failed") from self._error
            self._pending[lane] = req
            if len(self._pending) >= self._active:
                self._serve()
            else:
                while lane not in self._results and self._error is None:
                    self._cv.wait()
            if self._error is not None:
                raise RuntimeError("lockstep batch failed") from self._error
            return self._results.pop(lane)

    def _serve(self) -> None:
        """Runs with the lock held by the last lane to arrive (everybody else is waiting)."""
        try:
            groups: Dict[Tuple, List[int]] = {}
            for lane, req in self._pending.items():
                if req.thetas is not None:
                    self._thetas[lane] = req.thetas
                groups.setdefault(req.sig, []).append(lane)
            # calls that recompute Z = V^H Y first, so that sweeps of the same round see current data
            for sig in sorted(groups, key=lambda s: (not s[0], repr(s))):
                if sig[0] == "sur":   # whole evaluations: V^H, amplitudes, state update and one sweep per lane, on the device
                    _, mode, br, front = sig
                    w = np.ones(self.nlanes)
                    mx = np.zeros(self.nlanes, dtype=np.int64)
                    for lane in groups[sig]:
                        w[lane], mx[lane] = self._pending[lane].weight, self._pending[lane].max_no
                    f, fid, hs, gc = self.ws.surrogate_eval(self._thetas, w, mx, mode, br, front)
                    self._basis_dirty[BUF_X2] = True   # the call wrote every lane's lhs state into X2
                    self.native_calls += 1
                    for lane in groups[sig]:
                        self._results[lane] = (f[lane:lane + 1].copy(), None if fid is None else fid[lane:lane + 1].copy(),
                                               hs[lane:lane + 1].copy(), gc[lane:lane + 1].copy(), float(w[lane]), int(mx[lane]))
                    continue
                vdag, gather, grad, x_buf, br, front = sig
                if grad and self._basis_dirty.get(x_buf, False):
                    idx, cf = self._combo_idx[x_buf], self._combo_coef[x_buf]
                    if (idx[:, 1] < 0).all() and (cf[:, 0] == 1.0).all():
                        self.ws.set_basis(x_buf, idx[:, 0].copy())
                    else:
                        self.ws.set_combo(x_buf, idx, cf)
                    self._basis_dirty[x_buf] = False
                hs, g = self.ws.eval(self._thetas, vdag=vdag, gather=gather, grad=grad, x_buf=x_buf,
                                     block_range=br, front_layer=front)
                self.native_calls += 1
                for lane in groups[sig]:
                    self._results[lane] = (hs[lane:lane + 1].copy() if gather else None,
                                           g[lane:lane + 1].copy() if grad else None)
            self._pending.clear()
            self.rounds += 1
        except BaseException as ex:  # wake everybody up with the failure
            self._error = ex
        finally:
            self._cv.notify_all()

    def _retire(self) -> None:
        with self._cv:
            self._active -= 1
            if self._pending and len(self._pending) >= self._active and self._error is None:
                self._serve()

    def run(self, jobs: Sequence[Callable[[LaneView], Any]]) -> List[Any]:
        """Runs ``jobs[i](lane_view_i)`` concurrently, one thread per job; returns their results in
        order.  A job that raises has its exception object returned in its slot."""
        if not 0 < len(jobs) <= self.nlanes:
            raise ValueError("need between 1 and nlanes jobs")
        out: List[Any] = [None] * len(jobs)
        with self._cv:
            self._active = len(jobs)
            self._pending.clear()
            self._results.clear()
            self._error = None

        def worker(i: int) -> None:
            try:
                out[i] = jobs[i](self.lane(i))
            except BaseException as ex:  # noqa: BLE001 -- handed back to the caller
                out[i] = ex
            finally:
                self._retire()

        threads = [threading.Thread(target=worker, args=(i,), name=f"aqc-lane-{i}") for i in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        return out


def run_jobs_lockstep(
    circ,
    configs: List[Dict],
    seed: int,
    job_function: Callable[[int, Dict, LaneView], Dict],
    *,
    nlanes: int = 64,
    device: Optional[int] = None,
) -> List[Dict]:
    """``run_jobs`` for jobs that share the ansatz ``circ`` (e.g. the seeds of one time horizon):
    chunks of ``nlanes`` jobs advance in lockstep on one GPU.  ``job_function(job_index, config,
    workspace)`` must hand ``workspace`` to its objective (``user_parameters["workspace"]``) and draw
    random numbers from ``config["rng"]`` (a Generator seeded ``seed + 7*(job_index+1)``, the
    reference's per-job seed, job_executor.py:64) -- threads share the global NumPy RNG, so it is not
    reseeded here.  Records carry ``status`` / ``job_index`` / ``seed`` like ``run_jobs``."""
    import traceback
    from time import perf_counter

    if not (isinstance(configs, list) and configs and isinstance(configs[0], dict)):
        raise ValueError("configs must be a non-empty list of dictionaries")
    results: List[Dict] = []
    batch = LockstepBatch(circ, min(nlanes, len(configs)), device=device)
    try:
        for first in range(0, len(configs), batch.nlanes):
            chunk = list(range(first, min(first + batch.nlanes, len(configs))))

            def make(j: int):
                def job(view: LaneView) -> Dict:
                    cfg = dict(configs[j])
                    job_seed = seed + 7 * (j + 1)
                    cfg["rng"] = np.random.default_rng(job_seed)
                    tic = perf_counter()
                    try:
                        rec = job_function(j, cfg, view)
                        rec.update({"time": perf_counter() - tic, "status": "ok", "job_index": j, "seed": job_seed})
                    except Exception:
                        rec = {"time": float(-1), "status": traceback.format_exc(), "job_index": j, "seed": job_seed}
                    return rec
                return job

            for rec in batch.run([make(j) for j in chunk]):
                if isinstance(rec, BaseException):
                    raise rec
                results.append(rec)
    finally:
        batch.close()
    if not any(r["status"].startswith("ok") for r in results):
        raise RuntimeError("there is no valid simulation results")
    return results
