"""
Device-resident matrix-product states with truncated 2-qubit gates (C ABI ``aqc_mps_*``): the part of the
reference that runs inside qiskit-aer's ``matrix_product_state`` simulator (``mps_operations.py:216-298``,
``mps_dot_objective.py:41-468``), for registers too large for a dense state.

``DeviceMPS`` wraps one state; ``fast_dot_gradient_mps`` is the gate-by-gate gradient sweep of
``mps_dot_objective.fast_dot_gradient`` (:41-242) on two such states, ``v_mul_mps`` / ``v_dagger_mul_mps`` the
circuit applications of ``mps_operations.py:326-371``.  With ``trunc_thr -> 0`` everything is exact (tested
against the dense oracle); Aer's truncation arithmetic itself is third party and parity unpinned.
"""
import ctypes
from ctypes import POINTER, byref, c_int, c_int32, c_void_p
from typing import Optional, Tuple

import numpy as np

from . import _lib, gates
from ._lib import check, dptr

_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
_P1 = np.diag([0, 1]).astype(np.complex128)


def _pack(qiskit_mps):
    gam, lam = qiskit_mps
    n = len(gam)
    dims = np.ones(n + 1, dtype=np.int32)
    for q in range(n):
        g0 = np.asarray(gam[q][0])
        if g0.ndim != 2 or np.asarray(gam[q][1]).shape != g0.shape or g0.shape[0] != dims[q]:
            raise ValueError("inconsistent MPS tensor shapes")
        dims[q + 1] = g0.shape[1]
    if dims[n] != 1 or len(lam) != n - 1:
        raise ValueError("not a valid MPS in Qiskit format")
    g = np.concatenate([np.stack([np.asarray(a, dtype=np.complex128), np.asarray(b, dtype=np.complex128)]).ravel() for a, b in gam])
    lm = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in lam]) if n > 1 else np.zeros(1)
    for q in range(n - 1):
        if np.asarray(lam[q]).size != dims[q + 1]:
            raise ValueError("Schmidt vector size differs from the bond dimension")
    return n, dims, np.ascontiguousarray(g), np.ascontiguousarray(lm)


def is_canonical(qiskit_mps, tol: float = 1e-8) -> bool:
    """Vidal canonical form, checked on the host: with A_q[b] = Gamma_q[b] diag(lambda_q) every site is right-normalised,
    sum_b A_q[b] A_q[b]^H = 1, and with B_q[b] = diag(lambda_{q-1}) Gamma_q[b] left-normalised, sum_b B_q[b]^H B_q[b] = 1
    (O(n chi^3) on small matrices; Aer's states pass, hand-made tensors such as the oracle's random_mps do not)."""
    gam, lam = qiskit_mps
    n = len(gam)
    for q in range(n):
        g0, g1 = np.asarray(gam[q][0], dtype=np.complex128), np.asarray(gam[q][1], dtype=np.complex128)
        lr = np.asarray(lam[q], dtype=np.float64).ravel() if q < n - 1 else np.ones(1)
        ll = np.asarray(lam[q - 1], dtype=np.float64).ravel() if q > 0 else np.ones(1)
        right = sum((g * lr[None, :]) @ (g * lr[None, :]).conj().T for g in (g0, g1))
        left = sum((ll[:, None] * g).conj().T @ (ll[:, None] * g) for g in (g0, g1))
        if np.abs(right - np.eye(right.shape[0])).max() > tol or np.abs(left - np.eye(left.shape[0])).max() > tol:
            return False
    return True


def _default_device() -> int:
    from .engine import default_device

    return default_device()


_CANONICAL_CACHE: dict = {}   # id(tuple) -> (weak liveness token, verdict); a few entries (targets of running optimisations)


def _canonical_verdict(qiskit_mps) -> bool:
    """``is_canonical`` once per tuple object: the same target tuple arrives on every evaluation of an optimisation
    (mps_dot_objective.py:41).  Keyed by identity plus the shapes and a corner of every tensor (an in-place edit that keeps
    those is the caller's to announce by passing a new tuple)."""
    key = id(qiskit_mps)
    gam, lam = qiskit_mps
    mark = tuple((np.shape(g0), complex(np.asarray(g0).flat[0]), complex(np.asarray(g1).flat[-1])) for g0, g1 in gam)
    hit = _CANONICAL_CACHE.get(key)
    if hit is not None and hit[0] == mark:
        return hit[1]
    verdict = is_canonical(qiskit_mps)
    if len(_CANONICAL_CACHE) > 64:
        _CANONICAL_CACHE.clear()
    _CANONICAL_CACHE[key] = (mark, verdict)
    return verdict


_SERIALS = __import__("itertools").count(1)


class DeviceMPS:
    """One MPS resident on the GPU."""

    def __init__(self, handle: c_void_p):
        self.handle = handle
        self._L = _lib.lib()
        self.version = 0   # counts the in-place changes of the state (copies held elsewhere, e.g. by LockstepLanes, go stale)
        self.serial = next(_SERIALS)   # never reused (unlike id()): what holders of copies remember instead of a reference to the state

    @classmethod
    def from_qiskit(cls, qiskit_mps, device: Optional[int] = None, trunc_thr: float = 0.0, assume_canonical: bool = False) -> "DeviceMPS":
        """Uploads a QiskitMPS tuple.  The engine's truncation rule is stated on Schmidt values, i.e. for states in
        canonical (Vidal) form -- what Aer hands the reference (mps_operations.py:216-243).  When a real truncation is asked
        for (``trunc_thr`` > 1e-12) an input that is NOT canonical is brought into that form first (two sweeps of exact
        SVDs, ``canonicalize``); with exact arithmetic the gauge does not matter and the tensors are taken as they are.
        ``device`` None: this rank's GPU (``engine.default_device()``: LOCAL_RANK under a one-process-per-GPU launcher).
        The O(n chi^3) host check is made once per tuple (the verdict is remembered by the tuple's identity, the target of an
        optimisation arrives on every evaluation); ``assume_canonical`` skips it (states that come from Aer or this engine)."""
        n, dims, g, lm = _pack(qiskit_mps)
        h = c_void_p()
        dev = _default_device() if device is None else int(device)
        check(_lib.lib().aqc_mps_create(dev, n, dims.ctypes.data_as(POINTER(c_int32)), dptr(g), dptr(lm), byref(h)))
        m = cls(h)
        if trunc_thr > 1e-12 and not assume_canonical and not _canonical_verdict(qiskit_mps):
            m.canonicalize()
        return m

    def canonicalize(self) -> "DeviceMPS":
        """Canonical form by exact identity "gates" on every bond: right to left (all sites right-orthonormal), then left to
        right (the singular values of each bond are now the state's Schmidt values)."""
        eye4 = np.eye(4, dtype=np.complex128)
        n = self.num_qubits
        for q in range(n - 2, -1, -1):
            self.gate2(eye4, q, q + 1, 0.0)
        for q in range(n - 1):
            self.gate2(eye4, q, q + 1, 0.0)
        return self

    @classmethod
    def basis_state(cls, num_qubits: int, index: int = 0, device: Optional[int] = None) -> "DeviceMPS":
        """Product state |index> (bit q of ``index`` = qubit q)."""
        index = int(index)   # Python integer: registers beyond 63 qubits
        gam = [((np.array([[1.0 - ((index >> q) & 1)]], dtype=np.complex128)), np.array([[float((index >> q) & 1)]], dtype=np.complex128))
               for q in range(num_qubits)]
        return cls.from_qiskit((gam, [np.ones(1) for _ in range(num_qubits - 1)]), device)

    def clone(self) -> "DeviceMPS":
        h = c_void_p()
        check(self._L.aqc_mps_clone(self.handle, byref(h)))
        return DeviceMPS(h)

    @property
    def num_qubits(self) -> int:
        return int(self._L.aqc_mps_num_qubits(self.handle))

    @property
    def bond_dims(self) -> np.ndarray:
        d = np.zeros(self.num_qubits + 1, dtype=np.int32)
        check(self._L.aqc_mps_dims(self.handle, d.ctypes.data_as(POINTER(c_int32))))
        return d

    @property
    def discarded_weight(self) -> float:
        """Sum of the squared singular values dropped by all truncations so far."""
        return float(self._L.aqc_mps_discarded_weight(self.handle))

    def to_qiskit(self):
        d = self.bond_dims
        n = self.num_qubits
        sizes = [2 * int(d[q]) * int(d[q + 1]) for q in range(n)]
        g = np.empty(sum(sizes), dtype=np.complex128)
        lm = np.empty(max(int(d[1:n].sum()), 1), dtype=np.float64)
        check(self._L.aqc_mps_export(self.handle, dptr(g), dptr(lm)))
        gam, lam, off, loff = [], [], 0, 0
        for q in range(n):
            t = g[off:off + sizes[q]].reshape(2, int(d[q]), int(d[q + 1]))
            gam.append((t[0].copy(), t[1].copy()))
            off += sizes[q]
            if q < n - 1:
                lam.append(lm[loff:loff + int(d[q + 1])].copy())
                loff += int(d[q + 1])
        return gam, lam

    def gate1(self, gate, qubit: int) -> "DeviceMPS":
        g = np.ascontiguousarray(gate, dtype=np.complex128)
        if g.shape != (2, 2):
            raise ValueError("expects a 2x2 gate")
        check(self._L.aqc_mps_gate1(self.handle, int(qubit), dptr(g)))
        self.version += 1
        return self

    def gate2(self, gate4, ctrl: int, targ: int, trunc_thr: float = 0.0, max_bond: int = 0) -> "DeviceMPS":
        g = np.ascontiguousarray(gate4, dtype=np.complex128)
        if g.shape != (4, 4):
            raise ValueError("expects a 4x4 gate")
        check(self._L.aqc_mps_gate2(self.handle, int(ctrl), int(targ), dptr(g), float(trunc_thr), int(max_bond)))
        self.version += 1
        return self

    def dot(self, other: "DeviceMPS") -> np.complex128:
        """<self|other>."""
        out = np.empty(1, dtype=np.complex128)
        check(self._L.aqc_mps_dot(self.handle, other.handle, dptr(out)))
        return np.complex128(out[0])

    def dot_ops(self, other: "DeviceMPS", ops) -> np.complex128:
        """<(prod G_i on qubit_i) self|other> for ``ops = [(qubit, 2x2 gate), ...]`` on distinct qubits, without
        building the transformed state."""
        qs = np.ascontiguousarray([q for q, _ in ops], dtype=np.int32)
        gs = np.ascontiguousarray(np.stack([np.asarray(g, dtype=np.complex128) for _, g in ops]))
        out = np.empty(1, dtype=np.complex128)
        check(self._L.aqc_mps_dot_ops(self.handle, other.handle, len(ops), qs.ctypes.data_as(POINTER(c_int32)), dptr(gs), dptr(out)))
        return np.complex128(out[0])

    def close(self) -> None:
        if self.handle:
            self._L.aqc_mps_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def svd(a: np.ndarray, device: Optional[int] = None):
    """(U, S, Vh, sweeps) of a complex matrix by the engine's one-sided Jacobi kernel."""
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if a.ndim != 2:
        raise ValueError("expects a matrix")
    m, n = a.shape
    k = min(m, n)
    u, s, vh = np.empty((m, k), dtype=np.complex128), np.empty(k), np.empty((k, n), dtype=np.complex128)
    sweeps = c_int(0)
    check(_lib.lib().aqc_svd(_default_device() if device is None else int(device), m, n, dptr(a), dptr(u), dptr(s), dptr(vh), byref(sweeps)))
    return u, s, vh, sweeps.value


# ---- circuits on MPS ------------------------------------------------------------------------------------

def _ent_matrix(entangler: str, angle: float) -> np.ndarray:
    if entangler == "cx":
        return gates.controlled(_X)
    if entangler == "cz":
        return gates.controlled(_Z)
    return gates.controlled(np.diag([1.0, np.exp(1j * angle)]))


def _blocks_of(circ):
    """(running index i, parameter block j, ctrl, targ) incl. the virtual trailing half-layer of a 2nd-order
    Trotter ansatz (parametric_circuit.py:328-333)."""
    L = circ.num_blocks
    trotter = hasattr(circ, "is_second_order")
    tail = 3 * (circ.num_qubits // 2) if (trotter and circ.is_second_order) else 0
    return trotter, [(i, i % L, int(circ.blocks[0, i % L]), int(circ.blocks[1, i % L])) for i in range(L + tail)]


class _CircuitDesc(ctypes.Structure):
    """``aqc_circuit`` of include/aqc_hip.h."""

    _fields_ = [("num_qubits", c_int32), ("entangler", c_int32), ("num_blocks", c_int32), ("trotter", c_int32),
                ("second_order", c_int32), ("blocks", POINTER(c_int32))]


def _describe(circ):
    """(aqc_circuit, keep-alive array) of a ParametricCircuit / TrotterAnsatz."""
    blocks = np.ascontiguousarray(circ.blocks, dtype=np.int32)
    trotter = hasattr(circ, "is_second_order")
    desc = _CircuitDesc(circ.num_qubits, _lib.ENTANGLERS[circ.entangler], circ.num_blocks, int(trotter),
                        int(trotter and circ.is_second_order), blocks.ctypes.data_as(POINTER(c_int32)))
    return desc, blocks


def _thetas(circ, thetas) -> np.ndarray:
    th = np.ascontiguousarray(thetas, dtype=np.float64)
    if th.shape != (circ.num_thetas,):
        raise ValueError(f"expected {circ.num_thetas} thetas, got an array of shape {th.shape}")
    return th


def _apply_circuit(circ, thetas, mps: DeviceMPS, inverse: bool, trunc_thr: float, max_bond: int) -> DeviceMPS:
    """The whole ansatz in ONE ABI call (``aqc_mps_apply_circuit``)."""
    desc, keep = _describe(circ)
    th = _thetas(circ, thetas)
    check(_lib.lib().aqc_mps_apply_circuit(mps.handle, byref(desc), dptr(th), int(inverse), float(trunc_thr), int(max_bond)))
    mps.version += 1
    del keep
    return mps


def _apply_circuit_gatewise(circ, thetas, mps: DeviceMPS, inverse: bool, trunc_thr: float, max_bond: int) -> DeviceMPS:
    """The same walk with one ABI call per gate -- the cross-check of the C-side loop used by the tests."""
    n = circ.num_qubits
    th = np.asarray(thetas, dtype=np.float64)
    t1 = th[: 3 * n].reshape(n, 3)
    tpb = 5 if circ.entangler == "cp" else 4
    t2 = th[3 * n:].reshape(-1, tpb)
    rs = gates.rx_matrix if circ.entangler == "cx" else gates.rz_matrix
    trotter, blocks = _blocks_of(circ)
    if not inverse:   # core_operations.py:671-708
        for q in range(n):
            mps.gate1(gates.rz_matrix(t1[q, 0]) @ gates.ry_matrix(t1[q, 1]) @ gates.rz_matrix(t1[q, 2]), q)
        for i, j, c, t in blocks:
            b = t2[j]
            if trotter and i % 3 == 0:
                mps.gate1(gates.rz_matrix(-np.pi / 2), c)
            mps.gate2(_ent_matrix(circ.entangler, b[4] if tpb == 5 else 0.0), c, t, trunc_thr, max_bond)
            mps.gate1(gates.rz_matrix(b[1]) @ gates.ry_matrix(b[0]), c)
            mps.gate1(rs(b[3]) @ gates.ry_matrix(b[2]), t)
            if trotter and i % 3 == 2:
                mps.gate1(gates.rz_matrix(np.pi / 2), t)
    else:             # core_operations.py:787-818
        for i, j, c, t in reversed(blocks):
            b = t2[j]
            if trotter and i % 3 == 2:
                mps.gate1(gates.rz_matrix(-np.pi / 2), t)
            mps.gate1(gates.ry_matrix(-b[2]) @ rs(-b[3]), t)
            mps.gate1(gates.ry_matrix(-b[0]) @ gates.rz_matrix(-b[1]), c)
            mps.gate2(_ent_matrix(circ.entangler, -b[4] if tpb == 5 else 0.0), c, t, trunc_thr, max_bond)
            if trotter and i % 3 == 0:
                mps.gate1(gates.rz_matrix(np.pi / 2), c)
        for q in range(n):
            mps.gate1(gates.rz_matrix(-t1[q, 2]) @ gates.ry_matrix(-t1[q, 1]) @ gates.rz_matrix(-t1[q, 0]), q)
    return mps


def _apply_on_a_lane(circ, thetas, mps: DeviceMPS, inverse: bool, trunc_thr: float, max_bond: int) -> Optional[DeviceMPS]:
    """The circuit on a copy of ``mps`` through ONE lockstep lane (``LockstepLanes.apply_circuit``: the gates of a circuit layer in one
    launch, ranks decided on the device -- a 32-qubit Trotter circuit in a few milliseconds where the single-lane engine's launch chain
    takes tens); None when the state, the bond cap or a bond met on the way does not fit the lanes (bonds <= 32)."""
    n = circ.num_qubits
    if n < 2 or max_bond > LOCKSTEP_MAX_BOND or mps.num_qubits != n or int(mps.bond_dims.max()) > LOCKSTEP_MAX_BOND:
        return None
    import threading

    dev = int(_lib.lib().aqc_mps_device(mps.handle))
    key = (dev, n, 1, "apply", threading.get_ident())   # (a lane of its own per host thread: the thread lanes of evaluate_lanes call in parallel)
    ls = _lockstep_cached(key, 6, lambda: LockstepLanes(n, 1, dev))
    try:
        ls.set_targets(mps)
        ls.apply_circuit(circ, np.asarray(thetas, dtype=np.float64)[None, :], inverse=inverse, trunc_thr=trunc_thr, max_bond=max_bond)
        return ls.export(0)
    except RuntimeError as err:
        if "lockstep lanes" not in str(err):
            raise
        return None


def _apply_method(method: Optional[str]) -> str:
    import os

    method = os.environ.get("AQC_MPS_APPLY", "auto") if method is None else method
    if method not in ("auto", "single", "lockstep"):
        raise ValueError("method (or AQC_MPS_APPLY): 'auto', 'single' or 'lockstep'")
    return method


def v_mul_mps(circ, thetas, mps: DeviceMPS, trunc_thr: float = 0.0, max_bond: int = 0, method: Optional[str] = None) -> DeviceMPS:
    """V(thetas)|mps> on a copy (mps_operations.py:326-346).  ``method``: "single" -- the single-lane engine (one whole-circuit ABI
    call, any bond); "lockstep" / "auto" -- through one lockstep lane while bonds stay <= 32 ("auto" falls back to "single"); None:
    the environment's AQC_MPS_APPLY, else "auto"."""
    method = _apply_method(method)
    if method != "single":
        out = _apply_on_a_lane(circ, thetas, mps, False, trunc_thr, max_bond)
        if out is not None:
            return out
        if method == "lockstep":
            raise RuntimeError("aqc_hip: the state or a bond on the way exceeds the lockstep lanes (bonds <= 32)")
    return _apply_circuit(circ, thetas, mps.clone(), False, trunc_thr, max_bond)


def v_dagger_mul_mps(circ, thetas, mps: DeviceMPS, trunc_thr: float = 0.0, max_bond: int = 0, method: Optional[str] = None) -> DeviceMPS:
    """V(thetas)^H|mps> on a copy (mps_operations.py:349-371); ``method`` as in ``v_mul_mps``."""
    method = _apply_method(method)
    if method != "single":
        out = _apply_on_a_lane(circ, thetas, mps, True, trunc_thr, max_bond)
        if out is not None:
            return out
        if method == "lockstep":
            raise RuntimeError("aqc_hip: the state or a bond on the way exceeds the lockstep lanes (bonds <= 32)")
    return _apply_circuit(circ, thetas, mps.clone(), True, trunc_thr, max_bond)


def fast_dot_gradient_mps(circ, thetas, lvec: DeviceMPS, vh_phi: DeviceMPS, *, trunc_thr: float = 0.0, max_bond: int = 0,
                          block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True, method: Optional[str] = None) -> np.ndarray:
    """Complex gradient of <V lvec|phi> given vh_phi = V^H|phi>, gate by gate on two MPS
    (mps_dot_objective.py:41-242): w <- lvec, z <- vh_phi; every gate is applied to both and each parametrised
    rotation records 0.5j <P w|z>; the CPhase derivative is -1j <P11 w|z> taken before the gate.
    ONE ABI call: the walk, the cached environments behind the inner products and the single download of all of them live on the
    C side -- on one lockstep lane (``aqc_mpsb_gradient_of``) while bonds stay <= 32, on the single-lane engine
    (``aqc_mps_fast_dot_gradient``) otherwise; ``method`` as in ``v_mul_mps``."""
    desc, keep = _describe(circ)
    th = _thetas(circ, thetas)
    lo, hi = (-1, -1) if block_range is None else (int(block_range[0]), int(block_range[1]))
    if block_range is not None and not 0 <= lo <= hi <= circ.num_blocks:
        raise ValueError("invalid block range")
    grad = np.zeros(circ.num_thetas, dtype=np.complex128)
    method = _apply_method(method)
    n = circ.num_qubits
    if (method != "single" and n >= 2 and max_bond <= LOCKSTEP_MAX_BOND and lvec.num_qubits == n == vh_phi.num_qubits
            and max(int(lvec.bond_dims.max()), int(vh_phi.bond_dims.max())) <= LOCKSTEP_MAX_BOND):
        import threading

        dev = int(_lib.lib().aqc_mps_device(vh_phi.handle))   # one lockstep lane: the walk is a chain of fused steps instead of ~10 launches per gate
        key = (dev, n, 1, "gradient", threading.get_ident())
        ls = _lockstep_cached(key, 6, lambda: LockstepLanes(n, 1, dev))
        try:
            ls.set_targets(vh_phi).set_lhs(lvec)
            check(_lib.lib().aqc_mpsb_gradient_of(ls.handle, byref(desc), dptr(th), float(trunc_thr), int(max_bond), lo, hi, int(bool(front_layer)),
                                                  dptr(grad)))
            del keep
            return grad
        except RuntimeError as err:
            if method == "lockstep" or "lockstep lanes" not in str(err):
                raise
    elif method == "lockstep":
        raise RuntimeError("aqc_hip: the operands exceed the lockstep lanes (bonds <= 32)")
    check(_lib.lib().aqc_mps_fast_dot_gradient(byref(desc), lvec.handle, vh_phi.handle, dptr(th), float(trunc_thr), int(max_bond),
                                               lo, hi, int(bool(front_layer)), dptr(grad)))
    del keep
    return grad


def fast_dot_gradient_mps_gatewise(circ, thetas, lvec: DeviceMPS, vh_phi: DeviceMPS, *, trunc_thr: float = 0.0, max_bond: int = 0,
                                   block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True) -> np.ndarray:
    """The same sweep with one ABI call per gate and one full transfer-matrix chain per inner product -- the
    cross-check of the C-side walk used by the tests."""
    n = circ.num_qubits
    th = np.asarray(thetas, dtype=np.float64)
    tpb = 5 if circ.entangler == "cp" else 4
    L = circ.num_blocks
    lo, hi = (0, L) if block_range is None else (int(block_range[0]), int(block_range[1]))
    grad = np.zeros(circ.num_thetas, dtype=np.complex128)
    g1 = grad[: 3 * n].reshape(n, 3)
    g2 = grad[3 * n:].reshape(-1, tpb)
    t1 = th[: 3 * n].reshape(n, 3)
    t2 = th[3 * n:].reshape(-1, tpb)
    w, z = lvec.clone(), vh_phi.clone()
    pauli = {"x": _X, "y": _Y, "z": _Z}

    def both(gate, q):
        w.gate1(gate, q)
        z.gate1(gate, q)

    def dot(p: str, q: int) -> np.complex128:   # 0.5j <P w|z>, P folded into the transfer-matrix chain
        return 0.5j * w.dot_ops(z, [(q, pauli[p])])

    for q in range(n):   # front layer: Rz(t2), Ry(t1), Rz(t0), rightmost first (core_operations.py:921-935)
        for slot, mat, p in ((2, gates.rz_matrix, "z"), (1, gates.ry_matrix, "y"), (0, gates.rz_matrix, "z")):
            both(mat(t1[q, slot]), q)
            if front_layer:
                g1[q, slot] = dot(p, q)
    rs, ps = (gates.rx_matrix, "x") if circ.entangler == "cx" else (gates.rz_matrix, "z")
    trotter, blocks = _blocks_of(circ)
    for i, j, c, t in blocks:
        b = t2[j]
        live = lo <= j < hi
        if trotter and i % 3 == 0:
            both(gates.rz_matrix(-np.pi / 2), c)
        if live and tpb == 5:    # -1j <P11 w|z> before the gate (core_op_matrix.py:430-477)
            g2[j, 4] += -1j * w.dot_ops(z, [(c, _P1), (t, _P1)])
        ent = _ent_matrix(circ.entangler, b[4] if tpb == 5 else 0.0)
        z.gate2(ent, c, t, trunc_thr, max_bond)
        w.gate2(ent, c, t, trunc_thr, max_bond)
        for slot, mat, p, q in ((0, gates.ry_matrix, "y", c), (1, gates.rz_matrix, "z", c), (2, gates.ry_matrix, "y", t), (3, rs, ps, t)):
            both(mat(b[slot]), q)
            if live:
                g2[j, slot] += dot(p, q)
        if trotter and i % 3 == 2:
            both(gates.rz_matrix(np.pi / 2), t)
    w.close()
    z.close()
    return grad


# ---- lanes beyond dense reach ---------------------------------------------------------------------------------------
# Every DeviceMPS owns its stream and every circuit / gradient walk is ONE native call that releases the GIL, so independent
# problems (the seeds / restarts of a horizon, mps_dot_objective.py:41 called once per job in the reference, job_executor.py:141)
# run as lanes on host threads: the walks are chains of small dependent launches with one host decision (the truncation rank) per
# 2-qubit gate, i.e. latency-bound -- concurrent lanes fill the device while one lane waits.

_POOL = None


def _pool(workers: int):
    global _POOL
    from concurrent.futures import ThreadPoolExecutor

    if _POOL is None or _POOL._max_workers < workers:
        if _POOL is not None:
            _POOL.shutdown(wait=True)
        _POOL = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="aqc-mps-lane")
    return _POOL


LOCKSTEP_MAX_BOND = 32   # kCap of csrc/aqc_mps_batch.cpp


class LockstepLanes:
    """``lanes`` problems on one ansatz evaluated together and device-resident (C ABI ``aqc_mpsb_*``): every step of the gate walk of
    ``mps_dot_objective.fast_dot_gradient`` (:41-242) is one launch for all lanes; a truncated 2-qubit gate is one workgroup per lane
    (two-site tensor, SVD, rank decision, new tensors), the lane's bond dimensions never leave the device and the host enqueues a
    whole evaluation without waiting.  Bonds up to ``LOCKSTEP_MAX_BOND`` per lane; a lane that would grow beyond makes ``evaluate``
    raise (nothing is truncated silently) and ``evaluate_lanes`` repeats the batch lane by lane on the single-lane engine."""

    def __init__(self, num_qubits: int, lanes: int, device: Optional[int] = None):
        h = c_void_p()
        check(_lib.lib().aqc_mpsb_create(_default_device() if device is None else int(device), int(num_qubits), int(lanes), byref(h)))
        self.handle, self.num_qubits, self.lanes = h, int(num_qubits), int(lanes)
        self._targets = self._lhs = None

    def gate2_stats(self, enable: Optional[bool] = None, reset: bool = False) -> dict:
        """Work and time of the truncated 2-qubit gates (``aqc_mpsb_gate2_stats``): fp64 flops of the Jacobi rotations that ran, SVDs,
        sweeps, rotations; ``lanes_gate2`` launches timed and their total duration while ``enable`` is on."""
        out = np.zeros(6)
        check(_lib.lib().aqc_mpsb_gate2_stats(self.handle, -1 if enable is None else int(bool(enable)), dptr(out), int(bool(reset))))
        return {"jacobi_flops": float(out[0]), "svds": int(out[1]), "sweeps": int(out[2]), "rotations": int(out[3]),
                "launches_timed": int(out[4]), "launch_ms": float(out[5])}

    def _handles(self, states):
        lst = list(states) if isinstance(states, (list, tuple)) else [states]
        if len(lst) not in (1, self.lanes):
            raise ValueError("one state per lane, or one for all lanes")
        arr = (c_void_p * len(lst))(*[m.handle for m in lst])
        return lst, arr, int(len(lst) == 1)

    def set_targets(self, targets) -> "LockstepLanes":
        """Copies |phi_l> of every lane into the lanes (one state per lane, or one for all).  A call with the very states of the
        previous one, unchanged since (``DeviceMPS.version``), is free: the batches of an optimisation come back every iteration."""
        lst, arr, shared = self._handles(targets)
        stamp = [(m.serial, m.version) for m in lst]   # (no reference to the states: a closed one is not kept alive by its copy)
        if self._targets != stamp:
            check(_lib.lib().aqc_mpsb_set_targets(self.handle, arr, shared))
            self._targets = stamp
        return self

    def set_lhs(self, lhs) -> "LockstepLanes":
        """The same for the left-hand states <lhs_l|."""
        lst, arr, shared = self._handles(lhs)
        stamp = [(m.serial, m.version) for m in lst]
        if self._lhs != stamp:
            check(_lib.lib().aqc_mpsb_set_lhs(self.handle, arr, shared))
            self._lhs = stamp
        return self

    def evaluate(self, circ, thetas, *, trunc_thr: float = 0.0, max_bond: int = 0, block_range: Optional[Tuple[int, int]] = None,
                 front_layer: bool = True, details: bool = False):
        """(h[lanes], grads[lanes][T]) -- per lane the values of ``v_dagger_mul_mps`` + ``dot`` + ``fast_dot_gradient_mps``;
        with ``details`` also (discarded weight, largest bond) of every lane's V^H|target>."""
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        if th.shape != (self.lanes, circ.num_thetas):
            raise ValueError("thetas: expects shape (lanes, circ.num_thetas)")
        if circ.num_qubits != self.num_qubits:
            raise ValueError("circuit and lanes differ in the number of qubits")
        desc, keep = _describe(circ)
        lo, hi = (-1, -1) if block_range is None else (int(block_range[0]), int(block_range[1]))
        if block_range is not None and not 0 <= lo <= hi <= circ.num_blocks:
            raise ValueError("invalid block range")
        h = np.zeros(self.lanes, dtype=np.complex128)
        grads = np.zeros((self.lanes, circ.num_thetas), dtype=np.complex128)
        disc = np.zeros(self.lanes, dtype=np.float64)
        bonds = np.zeros(self.lanes, dtype=np.int32)
        check(_lib.lib().aqc_mpsb_eval(self.handle, byref(desc), dptr(th), float(trunc_thr), int(max_bond), lo, hi, int(bool(front_layer)),
                                       dptr(h), dptr(grads), dptr(disc), bonds.ctypes.data_as(POINTER(c_int32))))
        del keep
        return (h, grads, disc, bonds) if details else (h, grads)

    def set_lhs_basis(self, bits) -> "LockstepLanes":
        """lhs state of every lane = the computational-basis state ``bits[lane][qubit]`` (0 / 1), built on the device."""
        arr = np.ascontiguousarray(bits, dtype=np.uint8)
        if arr.shape != (self.lanes, self.num_qubits):
            raise ValueError("bits: expects shape (lanes, num_qubits)")
        check(_lib.lib().aqc_mpsb_set_lhs_basis(self.handle, arr.ctypes.data_as(POINTER(ctypes.c_uint8))))
        self._lhs = None
        return self

    def apply_vh(self, circ, thetas, *, trunc_thr: float = 0.0, max_bond: int = 0, flips: bool = False, half: bool = False, details: bool = False):
        """Phase 1 of ``evaluate``: vh_l = V(thetas[l])^H|target_l> stays in the lanes; returns amps[lanes][1 (+ n)] with
        amps[l][0] = <lhs_l|vh_l> and, with ``flips``, amps[l][1 + q] = <X_q lhs_l|vh_l>.  ``half``: lanes [lanes/2, lanes) repeat targets
        and thetas of the first half -- V^H runs once for both.  ``gradient`` continues from here."""
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        if th.shape != (self.lanes, circ.num_thetas) or circ.num_qubits != self.num_qubits:
            raise ValueError("thetas: expects shape (lanes, circ.num_thetas) on a circuit of the lanes' size")
        desc, keep = _describe(circ)
        na = 1 + self.num_qubits if flips else 1
        amps = np.zeros((self.lanes, na), dtype=np.complex128)
        disc = np.zeros(self.lanes, dtype=np.float64)
        bonds = np.zeros(self.lanes, dtype=np.int32)
        check(_lib.lib().aqc_mpsb_vh(self.handle, byref(desc), dptr(th), float(trunc_thr), int(max_bond), int(bool(half)), na, dptr(amps), dptr(disc),
                                     bonds.ctypes.data_as(POINTER(c_int32))))
        del keep
        return (amps, disc, bonds) if details else amps

    def apply_circuit(self, circ, thetas, *, inverse: bool = False, trunc_thr: float = 0.0, max_bond: int = 0, details: bool = False):
        """The lanes' working state <- V(thetas[l])|target_l> (or V^H with ``inverse``): ``v_mul_mps`` / ``v_dagger_mul_mps`` for every lane,
        the gates of a circuit layer in one launch.  ``export(lane)`` hands a result out; no lhs states needed."""
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        if th.shape != (self.lanes, circ.num_thetas) or circ.num_qubits != self.num_qubits:
            raise ValueError("thetas: expects shape (lanes, circ.num_thetas) on a circuit of the lanes' size")
        desc, keep = _describe(circ)
        disc = np.zeros(self.lanes, dtype=np.float64)
        bonds = np.zeros(self.lanes, dtype=np.int32)
        check(_lib.lib().aqc_mpsb_apply_circuit(self.handle, byref(desc), dptr(th), int(bool(inverse)), float(trunc_thr), int(max_bond), dptr(disc),
                                                bonds.ctypes.data_as(POINTER(c_int32))))
        del keep
        return (disc, bonds) if details else None

    def export(self, lane: int) -> DeviceMPS:
        """Lane ``lane`` of the working state (after ``apply_circuit`` / ``apply_vh``) as a ``DeviceMPS`` of its own."""
        h = c_void_p()
        check(_lib.lib().aqc_mpsb_export(self.handle, int(lane), byref(h)))
        return DeviceMPS(h)

    def gradient(self, circ, *, block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True) -> np.ndarray:
        """Phase 2: grads[lanes][T] (complex) of <V lhs_l|target_l> from the CURRENT lhs states and the vh of ``apply_vh``."""
        desc, keep = _describe(circ)
        lo, hi = (-1, -1) if block_range is None else (int(block_range[0]), int(block_range[1]))
        if block_range is not None and not 0 <= lo <= hi <= circ.num_blocks:
            raise ValueError("invalid block range")
        grads = np.zeros((self.lanes, circ.num_thetas), dtype=np.complex128)
        check(_lib.lib().aqc_mpsb_grad(self.handle, byref(desc), lo, hi, int(bool(front_layer)), dptr(grads)))
        del keep
        return grads

    def close(self) -> None:
        if getattr(self, "handle", None):
            _lib.lib().aqc_mpsb_destroy(self.handle)
            self.handle = None
            self._targets = self._lhs = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_LOCKSTEP_CACHE: dict = {}   # (device, qubits, lanes) -> LockstepLanes; the batches of a running optimisation come back every iteration
_LOCKSTEP_LOCK = __import__("threading").Lock()


def _lockstep_cached(key, cap: int, make):
    """Look up / create an entry of the shared cache under its lock (the host-thread lanes of ``evaluate_lanes`` arrive here
    together); the evicted object is freed when its last user lets go of it."""
    with _LOCKSTEP_LOCK:
        ls = _LOCKSTEP_CACHE.get(key)
        if ls is None:
            while len(_LOCKSTEP_CACHE) >= cap:
                _LOCKSTEP_CACHE.pop(next(iter(_LOCKSTEP_CACHE)), None)
            ls = _LOCKSTEP_CACHE[key] = make()
        return ls


def _lockstep_for(num_qubits: int, lanes: int, device: int, targets, lhs) -> LockstepLanes:
    key = (device, num_qubits, lanes)
    ls = _lockstep_cached(key, 4, lambda: LockstepLanes(num_qubits, lanes, device))
    # the operands are copied into the lanes; the copies are refreshed when a state was replaced or edited in place (DeviceMPS.version)
    ls.set_targets(targets)
    ls.set_lhs(lhs)
    return ls


def evaluate_lanes(circ, thetas: np.ndarray, targets, lhs, *, trunc_thr: float = 0.0, max_bond: int = 0,
                   block_range: Optional[Tuple[int, int]] = None, front_layer: bool = True, workers: int = 0, method: str = "auto"):
    """One objective+gradient evaluation per lane on the native engine: lane b computes vh_b = V(thetas[b])^H |targets[b]>
    (v_dagger_mul_mps), h_b = <lhs[b]|vh_b> and the complex gradient of <V lhs[b]|targets[b]> (fast_dot_gradient_mps).
    ``targets`` / ``lhs``: one DeviceMPS per lane, or a single one shared by all lanes (operands are only read).
    ``method``: "lockstep" -- all lanes walk the circuit together, one launch per step (``LockstepLanes``; bonds <= 32);
    "threads" -- every lane on the single-lane engine, lanes concurrently on ``workers`` host threads (default: one per lane, at most
    16); "auto" -- lockstep when the operands' bonds allow it (also for one lane), and the thread lanes if a lane outgrows the lockstep bond.
    Returns (h[B] complex, grads[B][T] complex)."""
    th = np.ascontiguousarray(thetas, dtype=np.float64)
    if th.ndim != 2 or th.shape[1] != circ.num_thetas:
        raise ValueError("thetas: expects shape (lanes, circ.num_thetas)")
    if method not in ("auto", "lockstep", "threads"):
        raise ValueError("method: 'auto', 'lockstep' or 'threads'")
    lanes = th.shape[0]
    tg = list(targets) if isinstance(targets, (list, tuple)) else [targets] * lanes
    lh = list(lhs) if isinstance(lhs, (list, tuple)) else [lhs] * lanes
    if len(tg) != lanes or len(lh) != lanes:
        raise ValueError("one target and one lhs state per lane (or one for all)")

    if method != "threads":   # (a single lane as well: its walk is 5x shorter on the lockstep kernels than on the single-lane engine's launch chain)
        distinct = list({id(m): m for m in tg + lh}.values())
        fits = circ.num_qubits >= 2 and max_bond <= LOCKSTEP_MAX_BOND and all(int(m.bond_dims.max()) <= LOCKSTEP_MAX_BOND for m in distinct)
        if fits or method == "lockstep":
            try:
                shared_t = targets if not isinstance(targets, (list, tuple)) else tg
                shared_l = lhs if not isinstance(lhs, (list, tuple)) else lh
                devs = {int(_lib.lib().aqc_mps_device(m.handle)) for m in distinct}   # the lanes live where the operands live
                if len(devs) != 1:
                    raise ValueError(f"the operands of evaluate_lanes live on different devices: {sorted(devs)}")
                ls = _lockstep_for(circ.num_qubits, lanes, devs.pop(), shared_t, shared_l)
                return ls.evaluate(circ, th, trunc_thr=trunc_thr, max_bond=max_bond, block_range=block_range, front_layer=front_layer)
            except RuntimeError as err:
                if method == "lockstep" or "lockstep lanes" not in str(err):
                    raise

    def one(b: int):
        vh = v_dagger_mul_mps(circ, th[b], tg[b], trunc_thr=trunc_thr, max_bond=max_bond, method="single")
        try:
            h = np.conj(vh.dot(lh[b]))   # <lhs|vh> on vh's own scratch and stream: an lhs state shared by the lanes is only read
            g = fast_dot_gradient_mps(circ, th[b], lh[b], vh, trunc_thr=trunc_thr, max_bond=max_bond, block_range=block_range,
                                      front_layer=front_layer, method="single")
        finally:
            vh.close()
        return h, g

    nw = min(lanes, workers if workers > 0 else 16)
    res = [one(0)] if lanes == 1 else list(_pool(nw).map(one, range(lanes)))
    return np.array([r[0] for r in res], dtype=np.complex128), np.stack([r[1] for r in res])
