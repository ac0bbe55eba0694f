"""
MPS twin of the gradient sweep with the reference signature (mps_dot_objective.py:41-242):
both MPS operands are contracted to dense states on the device and the fused state-vector
sweep runs on them -- the no-truncation semantics the reference's tests pin
(test_mps_fast_dot_gradient.py:126-153).
"""
from typing import Optional, Tuple

import numpy as np

from .engine import BUF_X, BUF_Z, HipContext
from .mps_operations import check_mps, mps_num_qubits, no_truncation_threshold


_DENSE_MAX_QUBITS = 24


def use_dense(num_qubits: int, trunc_thr: float) -> bool:
    """Dense (exact) evaluation on the fused state-vector kernels whenever the register fits (n <= 24), WHATEVER the
    truncation threshold: the exact state is within the discarded weight of any truncated one, so the reference's own
    default ``trunc_thr = 1e-6`` (user_options.py:55, handed to the objective at time_evol_best_init.py:80) takes the same
    fast path as 1e-16 -- the threshold only matters where tensors are exported (``DenseBackedMPS`` truncates lazily).
    Larger registers run on the native MPS engine.  AQC_MPS_METHOD = dense | mps overrides (``mps`` is the opt-in to the
    truncated-SVD arithmetic below 25 qubits)."""
    import os

    forced = os.environ.get("AQC_MPS_METHOD", "auto")
    if forced in ("dense", "mps"):
        return forced == "dense"
    del trunc_thr
    return num_qubits <= _DENSE_MAX_QUBITS


def fast_dot_gradient(
    circ,
    thetas: np.ndarray,
    lvec,
    vh_phi,
    *,
    trunc_thr: Optional[float] = no_truncation_threshold(),
    block_range: Optional[Tuple[int, int]] = None,
    front_layer: Optional[bool] = True,
) -> np.ndarray:
    """Complex gradient of <lvec|V^H|phi> given vh_phi = V^H|phi>, both in Qiskit MPS format."""
    if not (check_mps(lvec) and check_mps(vh_phi)):
        raise ValueError("lvec / vh_phi must be MPS in Qiskit format")
    if mps_num_qubits(lvec) != circ.num_qubits or mps_num_qubits(vh_phi) != circ.num_qubits:
        raise ValueError("lvec / vh_phi do not match the circuit")
    th = np.asarray(thetas, dtype=np.float64)
    if th.ndim != 1 or th.size != circ.num_thetas:
        raise ValueError("thetas: expects a float vector of size circ.num_thetas")
    if not (isinstance(trunc_thr, float) and trunc_thr >= 0):
        raise ValueError("trunc_thr must be a non-negative float")
    if block_range is not None and not (isinstance(block_range, tuple) and len(block_range) == 2
                                        and 0 <= block_range[0] < block_range[1] <= circ.num_blocks):
        raise ValueError("block_range must be a tuple (from, to) with 0 <= from < to <= num_blocks")
    if use_dense(circ.num_qubits, trunc_thr):
        from .mps_operations import DenseBackedMPS

        ws = HipContext.of(circ).workspace(1, 1)
        ws.set_thetas(th)
        # vh_phi straight from v_dagger_mul_mps on this workspace: its dense state is still in Z -- no host round trip;
        # operands that arrive as plain tuples keep a resident device copy (slot cache) and are contracted on the device
        if not (isinstance(vh_phi, DenseBackedMPS) and vh_phi.dense_on(ws, BUF_Z)):
            if isinstance(vh_phi, DenseBackedMPS):
                ws.upload(BUF_Z, vh_phi.dense_state)
            else:
                ws.mps_to_vec_batch([vh_phi], BUF_Z)
        if isinstance(lvec, DenseBackedMPS):
            ws.upload(BUF_X, lvec.dense_state)
        else:
            ws.mps_to_vec_batch([lvec], BUF_X)
        ws.grad(block_range, bool(front_layer))
        return ws.get_grads()[0]
    from .mps_engine import DeviceMPS, fast_dot_gradient_mps   # registers beyond dense reach, or real truncation

    w, z = DeviceMPS.from_qiskit(lvec, trunc_thr=float(trunc_thr)), DeviceMPS.from_qiskit(vh_phi, trunc_thr=float(trunc_thr))
    try:
        return fast_dot_gradient_mps(circ, th, w, z, trunc_thr=float(trunc_thr), block_range=block_range, front_layer=bool(front_layer))
    finally:
        w.close()
        z.close()


# ---- single gates on an MPS (mps_dot_objective.py:245-516) ------------------------------------------------
# 1-qubit gates are exact at the MPS level (bond dimensions do not change): the (2, chi_l * chi_r) tensor of
# the site is a one-qubit "state" with chi_l * chi_r columns for aqc_gate_1q.  2-qubit gates run on the native MPS
# engine (contraction + truncated Jacobi SVD on the device, any register size).

from . import gates as _gates  # noqa: E402
from .mps_operations import mps_dot as _mps_dot  # noqa: E402

_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)


def _gate1q_mul_mps(gate: np.ndarray, qubit: int, mps_vec):
    if not check_mps(mps_vec):
        raise ValueError("not a valid MPS in Qiskit format")
    gam, lam = mps_vec
    if not (isinstance(qubit, (int, np.integer)) and 0 <= qubit < len(gam)):
        raise ValueError("qubit out of range")
    g0, g1 = gam[qubit]
    site = np.ascontiguousarray(np.stack([g0, g1]).reshape(2, -1), dtype=np.complex128)
    _gates.apply_1q(gate, 0, site, site)
    new = list(gam)
    new[qubit] = (site[0].reshape(g0.shape).copy(), site[1].reshape(g1.shape).copy())
    return new, [np.array(v, dtype=np.float64) for v in lam]


def x_mul_mps(qubit: int, mps_vec):
    return _gate1q_mul_mps(_X, qubit, mps_vec)


def y_mul_mps(qubit: int, mps_vec):
    return _gate1q_mul_mps(_Y, qubit, mps_vec)


def z_mul_mps(qubit: int, mps_vec):
    return _gate1q_mul_mps(_Z, qubit, mps_vec)


def rx_mul_mps(angle: float, qubit: int, mps_vec):
    return _gate1q_mul_mps(_gates.rx_matrix(float(angle)), qubit, mps_vec)


def ry_mul_mps(angle: float, qubit: int, mps_vec):
    return _gate1q_mul_mps(_gates.ry_matrix(float(angle)), qubit, mps_vec)


def rz_mul_mps(angle: float, qubit: int, mps_vec):
    return _gate1q_mul_mps(_gates.rz_matrix(float(angle)), qubit, mps_vec)


def _gate2q_mul_mps(g2x2, ctrl: int, targ: int, mps_vec, trunc_thr: float):
    """Controlled gate on the MPS itself: contraction of the two sites, truncated SVD on the device
    (mps_engine.DeviceMPS.gate2); the singular values whose squares sum to less than trunc_thr are dropped."""
    from .mps_engine import DeviceMPS

    if not check_mps(mps_vec):
        raise ValueError("not a valid MPS in Qiskit format")
    if not (isinstance(trunc_thr, float) and trunc_thr >= 0):
        raise ValueError("trunc_thr must be a non-negative float")
    m = DeviceMPS.from_qiskit(mps_vec, trunc_thr=float(trunc_thr))
    try:
        return m.gate2(_gates.controlled(g2x2), int(ctrl), int(targ), float(trunc_thr)).to_qiskit()
    finally:
        m.close()


def cx_mul_mps(_: float, ctrl: int, targ: int, mps_vec, *, trunc_thr: float = no_truncation_threshold()):
    return _gate2q_mul_mps(_X, ctrl, targ, mps_vec, trunc_thr)


def cz_mul_mps(_: float, ctrl: int, targ: int, mps_vec, *, trunc_thr: float = no_truncation_threshold()):
    return _gate2q_mul_mps(_Z, ctrl, targ, mps_vec, trunc_thr)


def cp_mul_mps(angle: float, ctrl: int, targ: int, mps_vec, *, trunc_thr: float = no_truncation_threshold()):
    return _gate2q_mul_mps(np.diag([1.0, np.exp(1j * float(angle))]), ctrl, targ, mps_vec, trunc_thr)


def dot_x(qubit: int, w_vec, z_vec) -> np.complex128:
    """0.5j <X w|z> (mps_dot_objective.py:471-484)."""
    return np.complex128(0.5j * _mps_dot(x_mul_mps(qubit, w_vec), z_vec))


def dot_y(qubit: int, w_vec, z_vec) -> np.complex128:
    """0.5j <Y w|z> (mps_dot_objective.py:487-500)."""
    return np.complex128(0.5j * _mps_dot(y_mul_mps(qubit, w_vec), z_vec))


def dot_z(qubit: int, w_vec, z_vec) -> np.complex128:
    """0.5j <Z w|z> (mps_dot_objective.py:503-516)."""
    return np.complex128(0.5j * _mps_dot(z_mul_mps(qubit, w_vec), z_vec))
