"""
Full / sketched unitary-AQC objective ``1 - Re<X|V^H|Y>/k`` on the GPU: drop-in for
SketchingObjectiveEx + FullRangeSketchingVectors (sk_core.py:34-326).
"""
from time import perf_counter
from typing import Tuple

import numpy as np

from ..engine import BUF_X, BUF_Y, BUF_Z, HipContext, Workspace, zgemm
from ..parametric_circuit import ParametricCircuit


class SketchingVectorsBase:
    """Generator of X (2^n x k) and Y = U X (sk_core.py:34-91)."""

    def __init__(self, num_skvecs: int, target_mat: np.ndarray):
        if not (isinstance(target_mat, np.ndarray) and target_mat.ndim == 2 and target_mat.shape[0] == target_mat.shape[1]
                and target_mat.dtype == np.complex128):
            raise ValueError("target_mat must be a square complex128 matrix")
        num_skvecs = min(max(int(num_skvecs), 1), target_mat.shape[0])
        if num_skvecs & (num_skvecs - 1):
            raise ValueError("'num_skvecs' must be a power of 2 number")
        self._num_skvecs, self._target_mat = num_skvecs, target_mat

    num_skvecs = property(lambda self: self._num_skvecs)
    target_matrix = property(lambda self: self._target_mat)

    def generate(self, circ=None, thetas=None) -> Tuple[np.ndarray, np.ndarray]:
        raise NotImplementedError("abstract method")


class FullRangeSketchingVectors(SketchingVectorsBase):
    """X = I, Y = U (sk_core.py:300-326).  ``device_resident`` tells the objective that X
    and Y never change, so they are placed in HBM once instead of per call."""

    device_resident = True

    def __init__(self, target_mat: np.ndarray):
        super().__init__(target_mat.shape[0], target_mat)

    def generate(self, _=None, __=None) -> Tuple[np.ndarray, np.ndarray]:
        return np.eye(self._target_mat.shape[0], dtype=np.complex128), np.array(self._target_mat, dtype=np.complex128)


class RandomSketchingVectors(SketchingVectorsBase):
    """Fresh random orthonormal X on every request, Y = U X (sk_core.py:329-356).  The random draw and the
    QR are input generation on the host (same np.random call order as the reference); U X runs on the GPU."""

    def __init__(self, num_skvecs: int, target_mat: np.ndarray):
        super().__init__(num_skvecs, target_mat)
        if target_mat.shape[0] % self.num_skvecs:
            raise ValueError("the dimension must be divisible by num_skvecs")

    def generate(self, _=None, __=None) -> Tuple[np.ndarray, np.ndarray]:
        dim, k = self.target_matrix.shape[0], self.num_skvecs
        x_vecs, _r = np.linalg.qr(np.random.rand(dim, k) + 1j * np.random.rand(dim, k))
        x_vecs = np.ascontiguousarray(x_vecs)
        return x_vecs, zgemm(self.target_matrix, x_vecs)


class AlternatingSketchingVectors(SketchingVectorsBase):
    """Random subset of unit vectors / target columns per request (sk_core.py:359-401)."""

    def __init__(self, num_skvecs: int, target_mat: np.ndarray):
        super().__init__(num_skvecs, target_mat)
        dim = target_mat.shape[0]
        if dim % self.num_skvecs:
            raise ValueError("the dimension must be divisible by num_skvecs")
        self._offset = 0
        self._indices = np.random.permutation(dim)

    def generate(self, _=None, __=None) -> Tuple[np.ndarray, np.ndarray]:
        target, dim, k = self.target_matrix, self.target_matrix.shape[0], self.num_skvecs
        if self._offset >= dim:
            self._offset = 0
            self._indices = np.random.permutation(dim)
        idx = self._indices[self._offset : self._offset + k]
        x_vecs = np.zeros((dim, k), dtype=np.complex128)
        x_vecs[idx, np.arange(idx.size)] = 1
        y_vecs = np.ascontiguousarray(target[:, idx])
        self._offset += k
        return x_vecs, y_vecs


class EigenSketchingVectors(SketchingVectorsBase):
    """Randomised range finder of (V^H - U^H) (sk_core.py:404-464): X = qr((V^H - U^H) Omega), Y = U X.
    V^H Omega and both GEMMs run on the GPU."""

    def generate(self, circ=None, thetas=None) -> Tuple[np.ndarray, np.ndarray]:
        from ..core_op_matrix import v_dagger_mul_mat

        if circ is None or thetas is None or np.size(thetas) != circ.num_thetas:
            raise ValueError("EigenSketchingVectors needs the circuit and its thetas")
        dim, k, target = circ.dimension, self.num_skvecs, self.target_matrix
        omega = np.random.randn(dim, k).astype(np.complex128)
        omega = omega * 1j
        omega += np.random.randn(dim, k)
        uh_omega = zgemm(target, omega, conj_trans_a=True)                 # U^H Omega
        vh_omega = v_dagger_mul_mat(circ, np.asarray(thetas, dtype=np.float64), np.ascontiguousarray(omega.copy()), None)
        x_vecs, _r = np.linalg.qr(vh_omega - uh_omega)
        x_vecs = np.ascontiguousarray(x_vecs)
        return x_vecs, zgemm(target, x_vecs)


def skvecs_generator(skvecs_type: str, num_skvecs: int, target_mat: np.ndarray) -> SketchingVectorsBase:
    """Factory of sk_core.py:467-494."""
    if skvecs_type == "full" or num_skvecs == target_mat.shape[0]:
        return FullRangeSketchingVectors(target_mat)
    if skvecs_type == "rand":
        return RandomSketchingVectors(num_skvecs, target_mat)
    if skvecs_type == "alt":
        return AlternatingSketchingVectors(num_skvecs, target_mat)
    if skvecs_type == "eigen":
        return EigenSketchingVectors(num_skvecs, target_mat)
    raise ValueError(f"unknown type of sketching vectors generator, expects one of: ['full', 'rand', 'alt', 'eigen'], got {skvecs_type}")


class SketchingObjectiveEx:
    """fobj = 1 - Re Tr<V Q, U Q>/k and its gradient (sk_core.py:94-297); stoppers and the
    gradient amplifier are duck-typed host objects."""

    def __init__(self, circ: ParametricCircuit, skvecs: SketchingVectorsBase, *, enable_stats: bool = False,
                 grad_scaler=None, stop_timeout=None, stop_stagnant=None, stop_small_fobj=None, logger=None, device=None,
                 column_shard: bool = False):
        if not isinstance(skvecs, SketchingVectorsBase):
            raise TypeError("skvecs must derive from SketchingVectorsBase")
        if skvecs.target_matrix.shape[0] != circ.dimension:
            raise ValueError("target matrix does not match the circuit dimension")
        self._circ, self._skvecs, self._target = circ, skvecs, skvecs.target_matrix
        self._enable_stats, self._grad_scaler, self._logger = enable_stats, grad_scaler, logger
        self._stop_timeout, self._stop_stagnant, self._stop_small_fobj = stop_timeout, stop_stagnant, stop_small_fobj
        self._fobj_best = float(np.inf)
        self._thetas_best = np.zeros(circ.num_thetas)
        self._nit = 0
        self._fobj_profile = []
        self._fobj_latest = float(1e30)
        self._grad_latest = np.empty(0)
        self._thetas_latest = np.empty(0)
        self._elapsed_time = perf_counter()
        self._period = int(round(10 + 60.0 / (1 + 2.0 ** (6 - circ.num_qubits))))
        self._structure = None
        self._device = device
        self._ws = None
        # column_shard: with one process per GPU every rank runs the gate sequence on its own slab of the k sketching
        # columns -- columns are independent until the final trace -- and the per-rank (trace, complex gradient)
        # records are summed with ONE all-reduce of 2(T+1) doubles per evaluation (aqc_comm: RCCL over xGMI bound
        # directly; latency-bound at this size; the gloo double on CPU).
        self._shard = None
        if column_shard:
            from ..comm import from_environment

            comm = from_environment()
            if comm.size > 1:
                k, world, rank = skvecs.num_skvecs, comm.size, comm.rank
                self._shard = (comm, (k * rank) // world, (k * (rank + 1)) // world)
                if self._shard[2] <= self._shard[1]:
                    raise ValueError("more ranks than sketching columns")

    def _workspace(self):
        """Workspace for the circuit's current structure (blocks may be edited between calls)."""
        ctx = HipContext.of(self._circ)
        if self._ws is None or self._structure != ctx.key:
            # A PRIVATE workspace: X = I and the target stay resident in it between evaluations, so it must not be
            # the cached one the function-level drop-ins (core_op_matrix) share and overwrite.
            if self._ws is not None:
                self._ws.close()
            self._ws = Workspace(ctx, batch=1, ncols=self._skvecs.num_skvecs, device=self._device)
            self._structure = ctx.key
            if getattr(self._skvecs, "device_resident", False):
                self._ws.set_identity(BUF_X)
                self._ws.upload(BUF_Y, np.ascontiguousarray(self._target, dtype=np.complex128))
        return self._ws

    def objective_and_gradient(self, thetas: np.ndarray) -> Tuple[float, np.ndarray]:
        now = perf_counter()
        if self._elapsed_time + self._period < now:
            print(".", end="", flush=True)
            self._elapsed_time = now
        k = self._skvecs.num_skvecs
        if self._shard is None:
            ws = self._workspace()
            if not getattr(self._skvecs, "device_resident", False):
                x, y = self._skvecs.generate(self._circ, thetas)
                ws.upload(BUF_X, x)
                ws.upload(BUF_Y, y)
            ws.set_thetas(thetas)
            ws.apply(True, BUF_Y, BUF_Z)                      # V^H Y            (sk_core.py:191)
            trace = ws.vdot(BUF_X, BUF_Z)[0]                  # <X|V^H Y>        (:192)
            ws.grad(None, True)                               # sweep            (:193)
            cgrad = ws.get_grads()[0]
        else:
            trace, cgrad = self._sharded_eval(thetas)
        fobj = float(1 - np.real(trace) / k)
        grad = -np.real(cgrad) / k
        if self._grad_scaler:
            grad *= self._grad_scaler.estimate(fobj)
        if fobj < self._fobj_best:
            self._fobj_best = fobj
            np.copyto(self._thetas_best, thetas)
        self._nit += 1
        if self._enable_stats:
            self._fobj_profile.append(float(fobj))
        if self._logger is not None:
            print(f"\riter: {self._nit:4d}, fobj: {fobj:0.4f}, |grad|: {np.linalg.norm(grad):0.5f}")
        if self._stop_timeout:
            self._stop_timeout.check()
        if self._stop_stagnant:
            self._stop_stagnant.check(fobj=fobj, iter_no=self._nit)
        if self._stop_small_fobj:
            self._stop_small_fobj.check(fobj=fobj)
        return fobj, grad

    def _sharded_eval(self, thetas: np.ndarray):
        """This rank's column slab, then the all-reduce of (trace, gradient)."""
        comm, c0, c1 = self._shard
        ctx = HipContext.of(self._circ)
        resident = getattr(self._skvecs, "device_resident", False)
        fresh = self._ws is None or self._structure != ctx.key
        if fresh:
            if self._ws is not None:
                self._ws.close()
            self._ws = Workspace(ctx, batch=1, ncols=c1 - c0, device=self._device)   # private, see _workspace
            self._structure = ctx.key
        ws = self._ws
        if fresh or not resident:
            x, y = (np.eye(self._circ.dimension, dtype=np.complex128), self._target) if resident else self._skvecs.generate(self._circ, thetas)
            ws.upload(BUF_X, np.ascontiguousarray(x[:, c0:c1], dtype=np.complex128))
            ws.upload(BUF_Y, np.ascontiguousarray(y[:, c0:c1], dtype=np.complex128))
        ws.set_thetas(thetas)
        ws.apply(True, BUF_Y, BUF_Z)
        rec = np.empty(1 + self._circ.num_thetas, dtype=np.complex128)
        rec[0] = ws.vdot(BUF_X, BUF_Z)[0]
        ws.grad(None, True)
        rec[1:] = ws.get_grads()[0]
        comm.allreduce(rec.view(np.float64), "sum")
        return rec[0], rec[1:]

    def objective(self, thetas: np.ndarray) -> float:
        self._thetas_latest = np.array(thetas, dtype=np.float64)
        self._fobj_latest, self._grad_latest = self.objective_and_gradient(thetas)
        return self._fobj_latest

    def gradient(self, thetas: np.ndarray) -> np.ndarray:
        tol = float(10.0 * np.finfo(np.float64).eps)
        last = self._thetas_latest
        if last.size == 0 or not np.allclose(thetas, last, atol=tol, rtol=tol):
            self.objective(thetas)
        return self._grad_latest

    @property
    def statistics(self) -> dict:
        return {"convergence_profile": np.asarray(self._fobj_profile, dtype=np.float32), "nit": self._nit}

    num_iterations = property(lambda self: int(self._nit))

    @property
    def optim_results(self) -> dict:
        return {
            "cost": float(self._fobj_best),
            "num_fun_ev": int(self._nit),
            "num_grad_ev": int(self._nit),
            "num_iters": int(self._nit),
            "thetas": self._thetas_best,
            "entangler": self._circ.entangler,
            "blocks": self._circ.blocks.copy(),
        }

    def set_status_trackers(self, timeout, stopper):
        """Compatibility with AqcOptimizer.optimize (optimizer.py:561-563)."""
