"""
MPS helpers with the reference signatures (mps_operations.py:33-371) on the GPU.

The reference delegates gate application on MPS to qiskit-aer's matrix_product_state
simulator (third party).  This path works at the level the reference's own tests pin it
-- exact (no truncation) dense semantics: an MPS is contracted to its 2^n amplitudes on the
device (chain of zgemm), the state-vector kernels do the work, and a result that must be an MPS
again comes back as a ``DenseBackedMPS``: a QiskitMPS tuple whose canonical (Vidal-form, Schmidt
values sorted, discarded weight <= trunc_thr) tensors are computed when somebody looks at them, and
which remembers where its dense state still sits on the device, so that ``v_dagger_mul_mps`` ->
``fast_dot_gradient`` hands the state over without a round trip through the host.  Registers beyond
dense reach and real truncation go to the native MPS engine (mps_engine.py).
"""
from typing import List, Optional, Tuple

import numpy as np

from .engine import BUF_Y, BUF_Z, HipContext  # noqa: F401

_NO_TRUNCATION_THR = 1e-16
QiskitMPS = Tuple[List[Tuple[np.ndarray, np.ndarray]], List[np.ndarray]]


def no_truncation_threshold() -> float:
    return _NO_TRUNCATION_THR


def check_mps(qiskit_mps) -> bool:
    """Structural check of a (gammas, lambdas) tuple (mps_operations.py:87-123)."""
    if not (isinstance(qiskit_mps, tuple) and len(qiskit_mps) == 2):
        return False
    if isinstance(qiskit_mps, DenseBackedMPS):   # canonical by construction; looking inside would run its SVD chain
        return True
    gam, lam = qiskit_mps
    n = len(gam)
    if len(lam) != n - 1:
        return False
    for q in range(n):
        g = gam[q]
        if len(g) != 2 or np.ndim(g[0]) != 2 or np.shape(g[0]) != np.shape(g[1]):
            return False
        if q < n - 1:
            l = np.asarray(lam[q])
            if not (l.ndim == 1 or (l.ndim == 2 and min(l.shape) == 1)):
                return False
            l = l.ravel()
            if not np.all(l[:-1] >= l[1:]):
                return False
    return True


def mps_num_qubits(qiskit_mps) -> int:
    """Number of sites, without looking inside a ``DenseBackedMPS`` (that would run its SVD chain)."""
    return qiskit_mps.num_qubits if isinstance(qiskit_mps, DenseBackedMPS) else len(qiskit_mps[0])


class _Circ:
    """Gate-free n-qubit descriptor: MPS helpers only need a device workspace of size 2^n."""

    entangler = "cx"

    def __init__(self, n):
        self.num_qubits, self.blocks = n, np.zeros((2, 0), dtype=np.int64)


def _workspace(n: int):
    return HipContext.of(_Circ(n)).workspace(1, 1)


_DOT_ON_WORKSPACE_MAX = 24   # registers up to this size contract MPS on the (cached) dense workspace of their size


def mps_to_vector(qiskit_mps) -> np.ndarray:
    """Dense state of an MPS; index bit q <-> site q (mps_operations.py:159-189)."""
    if not check_mps(qiskit_mps):
        raise ValueError("not a valid MPS in Qiskit format")
    if isinstance(qiskit_mps, DenseBackedMPS) and qiskit_mps.exact:
        return qiskit_mps.dense_state.copy()
    ws = _workspace(len(qiskit_mps[0]))
    ws.mps_upload(0, qiskit_mps)
    ws.mps_to_vec(0, BUF_Y, 0)
    return ws.download(BUF_Y, lane=0)


def mps_dot(qiskit_mps1, qiskit_mps2) -> np.complex128:
    """<mps1|mps2> by transfer matrices (mps_operations.py:192-213)."""
    if not (check_mps(qiskit_mps1) and check_mps(qiskit_mps2)):
        raise ValueError("not a valid MPS in Qiskit format")
    if len(qiskit_mps1[0]) != len(qiskit_mps2[0]):
        raise ValueError("MPS with different numbers of qubits")
    if len(qiskit_mps1[0]) > _DOT_ON_WORKSPACE_MAX:   # beyond dense reach: the transfer-matrix chain of the MPS engine, no 2^n workspace behind it
        from .mps_engine import DeviceMPS

        a, b = DeviceMPS.from_qiskit(qiskit_mps1), DeviceMPS.from_qiskit(qiskit_mps2)
        try:
            return np.complex128(a.dot(b))
        finally:
            a.close()
            b.close()
    ws = _workspace(len(qiskit_mps1[0]))
    ws.mps_upload(0, qiskit_mps1)
    ws.mps_upload(1, qiskit_mps2)
    return np.complex128(ws.mps_dot(0, 1))


def vector_to_exact_mps(vec: np.ndarray) -> QiskitMPS:
    """Exact MPS of a dense state without any SVD: index-routing (0/1) tensors on both
    halves, all amplitudes in the middle site, lambdas = 1.  Pure data movement."""
    vec = np.asarray(vec, dtype=np.complex128).ravel()
    n = int(round(np.log2(vec.size)))
    if vec.size != 1 << n or n < 2:
        raise ValueError("expects a vector of size 2^n, n >= 2")
    h = n // 2
    gam, lam = [], []
    for q in range(h):  # left half: new bond index j = i + 2^q b
        g = np.zeros((2, 1 << q, 2 << q), dtype=np.complex128)
        i = np.arange(1 << q)
        g[0, i, i] = 1
        g[1, i, i + (1 << q)] = 1
        gam.append((g[0], g[1]))
    rest = n - 1 - h  # bits above the middle site
    mid = vec.reshape(1 << rest, 2, 1 << h)  # [r, b, j]
    gam.append((np.ascontiguousarray(mid[:, 0, :].T), np.ascontiguousarray(mid[:, 1, :].T)))
    for q in range(h + 1, n):  # right half: left bond l = b + 2 r
        r = 1 << (n - 1 - q)
        g = np.zeros((2, 2 * r, r), dtype=np.complex128)
        j = np.arange(r)
        g[0, 2 * j, j] = 1
        g[1, 2 * j + 1, j] = 1
        gam.append((g[0], g[1]))
    for q in range(n - 1):
        lam.append(np.ones(gam[q][0].shape[1]))
    return gam, lam


def _svd(a: np.ndarray):
    """Thin SVD.  LAPACK's divide-and-conquer driver (numpy's only one) occasionally gives up on the nearly rank-deficient
    matrices met here ("SVD did not converge", seen at 16 qubits / bond 16); the QR-iteration driver does not."""
    try:
        return np.linalg.svd(a, full_matrices=False)
    except np.linalg.LinAlgError:
        import scipy.linalg

        return scipy.linalg.svd(a, full_matrices=False, lapack_driver="gesvd")


def vector_to_canonical_mps(vec: np.ndarray, trunc_thr: float = _NO_TRUNCATION_THR) -> QiskitMPS:
    """Canonical Vidal form (Gamma, lambda) of a dense state by successive SVDs -- the form Aer hands the reference
    (mps_operations.py:216-243): Schmidt values descending, at every bond the smallest ones dropped while the sum of their
    squares stays below ``trunc_thr`` (none at the reference's no-truncation threshold 1e-16) and those below 1e-14 of the
    largest, kept values renormalised."""
    vec = np.asarray(vec, dtype=np.complex128).ravel()
    n = int(round(np.log2(vec.size)))
    if vec.size != 1 << n or n < 2:
        raise ValueError("expects a vector of size 2^n, n >= 2")
    rest = vec.reshape([2] * n).transpose(list(range(n - 1, -1, -1))).reshape(1, -1)   # axes (b_0, ..., b_{n-1})
    gam, lam, prev = [], [], np.ones(1)
    for _ in range(n - 1):
        chi_l = rest.shape[0]
        u, sv, vh = _svd(rest.reshape(chi_l * 2, -1))
        keep = int((sv > 1e-14 * sv[0]).sum())
        total, dropped = float(np.sum(sv ** 2)), 0.0
        # the rule of the native engine (mps_engine.py); the reference's "no truncation" threshold (1e-16) really means none:
        # only numerically zero Schmidt values go, so that the tensors reproduce the state to ~1e-14
        while trunc_thr > _NO_TRUNCATION_THR and keep > 1 and dropped + sv[keep - 1] ** 2 <= trunc_thr * total:
            dropped += sv[keep - 1] ** 2
            keep -= 1
        u, sv, vh = u[:, :keep], sv[:keep] / np.linalg.norm(sv[:keep]), vh[:keep]
        a = u.reshape(chi_l, 2, keep)
        gam.append((np.ascontiguousarray(a[:, 0, :] / prev[:, None]), np.ascontiguousarray(a[:, 1, :] / prev[:, None])))
        lam.append(sv.copy())
        prev = sv
        rest = sv[:, None] * vh
    a = rest.reshape(rest.shape[0], 2, 1)
    gam.append((np.ascontiguousarray(a[:, 0, :] / prev[:, None]), np.ascontiguousarray(a[:, 1, :] / prev[:, None])))
    return gam, lam


class DenseBackedMPS(tuple):
    """A QiskitMPS ``(gammas, lambdas)`` tuple that was produced from a dense state on the device.

    * It IS a tuple of length 2 (``check_mps``, unpacking, indexing and iteration all work); its canonical tensors are
      computed from the host copy of the state the first time any of them is asked for.
    * ``dense_on(ws, buf)`` tells a later call on the same workspace whether the dense state is still in that buffer
      (nothing has overwritten it since): ``fast_dot_gradient`` then sweeps from it directly."""

    def __new__(cls, vec: np.ndarray, trunc_thr: float, ws=None, buf: int = BUF_Z):
        self = super().__new__(cls, (None, None))
        self._vec = np.asarray(vec, dtype=np.complex128).ravel()
        self._thr = float(trunc_thr)
        self._ws, self._buf = ws, buf
        self._gen = None if ws is None else ws.generation(buf)
        self._mps = None
        return self

    def _materialise(self) -> tuple:
        if self._mps is None:
            self._mps = vector_to_canonical_mps(self._vec, self._thr)
        return self._mps

    @property
    def exact(self) -> bool:
        """No truncation asked for: the tensors stand for the dense state itself (to rounding)."""
        return self._thr <= 1e-12

    def dense_on(self, ws, buf: int) -> bool:
        return ws is self._ws and buf == self._buf and ws.generation(buf) == self._gen

    @property
    def num_qubits(self) -> int:
        return int(self._vec.size).bit_length() - 1

    @property
    def dense_state(self) -> np.ndarray:
        """The 2^n amplitudes this MPS stands for (host copy)."""
        return self._vec

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return 2

    def __eq__(self, other):
        return self is other

    def __hash__(self):
        return id(self)

    def __reduce__(self):   # pickled (result files of the drivers) as the plain tuple it stands for
        return (tuple, (self._materialise(),))

    def __repr__(self):
        return f"DenseBackedMPS(n={int(np.log2(self._vec.size))}, materialised={self._mps is not None})"


def _apply_to_mps(circ, thetas, mps_vec, inverse: bool, trunc_thr) -> QiskitMPS:
    if not check_mps(mps_vec) or mps_num_qubits(mps_vec) != circ.num_qubits:
        raise ValueError("MPS does not match the circuit")
    thr = _NO_TRUNCATION_THR if trunc_thr is None else float(trunc_thr)
    from .mps_dot_objective import use_dense

    if use_dense(circ.num_qubits, thr):   # exact: fused state-vector kernels on the densified state
        ws = HipContext.of(circ).workspace(1, 1)
        ws.set_thetas(thetas)
        if isinstance(mps_vec, DenseBackedMPS) and mps_vec.dense_on(ws, BUF_Z):
            ws.copy_lane_from(ws, BUF_Z, 0, BUF_Y, 0)          # the operand is still on the device
        elif isinstance(mps_vec, DenseBackedMPS):
            ws.upload(BUF_Y, mps_vec.dense_state)
        else:
            ws.mps_to_vec_batch([mps_vec], BUF_Y)              # resident copy of the operand (slot cache)
        ws.apply(inverse, BUF_Y, BUF_Z)
        return DenseBackedMPS(ws.download(BUF_Z, lane=0), thr, ws, BUF_Z)
    from . import mps_engine                # large registers / real truncation: gate by gate on the MPS

    m = mps_engine.DeviceMPS.from_qiskit(mps_vec, trunc_thr=thr)
    try:   # (one lockstep lane while bonds stay <= 32 -- the gates of a circuit layer in one launch --, the single-lane engine otherwise)
        out = (mps_engine.v_dagger_mul_mps if inverse else mps_engine.v_mul_mps)(circ, thetas, m, trunc_thr=thr)
        try:
            return out.to_qiskit()
        finally:
            out.close()
    finally:
        m.close()


def v_mul_mps(circ, thetas: np.ndarray, mps_vec, *, trunc_thr: Optional[float] = _NO_TRUNCATION_THR) -> QiskitMPS:
    """V |mps> (mps_operations.py:326-346): exact below 25 qubits, truncated-SVD MPS arithmetic beyond."""
    return _apply_to_mps(circ, thetas, mps_vec, False, trunc_thr)


def v_dagger_mul_mps(circ, thetas: np.ndarray, mps_vec, *, trunc_thr: Optional[float] = _NO_TRUNCATION_THR) -> QiskitMPS:
    """V^H |mps> (mps_operations.py:349-371): exact below 25 qubits, truncated-SVD MPS arithmetic beyond."""
    return _apply_to_mps(circ, thetas, mps_vec, True, trunc_thr)
