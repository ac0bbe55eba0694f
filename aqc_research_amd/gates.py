"""
Gate-level building blocks on the GPU (aqc_gate_1q / aqc_gate_2q / aqc_gate_dot): the shared back end of the
single-gate functions of ``core_operations`` and ``core_op_matrix``.  Arrays are (2^n x k) row-major complex128
in host memory (k = 1: a state vector); ``qubit`` is the bit index of the row (Qiskit order).  Only the tiny
gate matrices are built on the host; every pass over the data runs on the device.
"""
from typing import Optional

import numpy as np

from . import _lib
from ._lib import check, dptr

_P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)
_P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)


def shape_of(arr: np.ndarray):
    """(n, ncols) of a contiguous complex128 vector (2^n,) or matrix (2^n, k)."""
    if not (isinstance(arr, np.ndarray) and arr.dtype == np.complex128 and arr.flags.c_contiguous and arr.ndim in (1, 2)):
        raise TypeError("expects a C-contiguous complex128 vector or matrix")
    rows = arr.shape[0]
    n = int(rows).bit_length() - 1
    if rows < 2 or (1 << n) != rows:
        raise ValueError("the leading dimension must be a power of two >= 2")
    return n, (1 if arr.ndim == 1 else int(arr.shape[1]))


def rx_matrix(angle: float) -> np.ndarray:
    c, s = np.cos(0.5 * angle), np.sin(0.5 * angle)
    return np.array([[c, -1j * s], [-1j * s, c]], dtype=np.complex128)


def ry_matrix(angle: float) -> np.ndarray:
    c, s = np.cos(0.5 * angle), np.sin(0.5 * angle)
    return np.array([[c, -s], [s, c]], dtype=np.complex128)


def _dev(device) -> int:
    """None: this rank's GPU (LOCAL_RANK under a one-process-per-GPU launcher, engine.default_device)."""
    if device is not None:
        return int(device)
    from .engine import default_device

    return default_device()


def rz_matrix(angle: float) -> np.ndarray:
    return np.array([[np.exp(-0.5j * angle), 0], [0, np.exp(0.5j * angle)]], dtype=np.complex128)


def apply_1q(gate, qubit: int, src: np.ndarray, dst: np.ndarray, device: Optional[int] = None) -> np.ndarray:
    """dst <- (I x gate x I) src; dst may be src."""
    n, k = shape_of(src)
    if dst.shape != src.shape or shape_of(dst) != (n, k):
        raise ValueError("source and destination differ in shape")
    if not 0 <= qubit < n:
        raise ValueError("qubit out of range")
    g = np.ascontiguousarray(gate, dtype=np.complex128)
    if g.shape != (2, 2):
        raise ValueError("expects a 2x2 gate")
    check(_lib.lib().aqc_gate_1q(_dev(device), n, k, int(qubit), dptr(g), dptr(src), dptr(dst)))
    return dst


def apply_2q(gate4, ctrl: int, targ: int, src: np.ndarray, dst: np.ndarray, device: Optional[int] = None) -> np.ndarray:
    """dst <- (4x4 gate on (ctrl, targ), basis index 2*bit_ctrl + bit_targ) src; dst may be src."""
    n, k = shape_of(src)
    if dst.shape != src.shape or shape_of(dst) != (n, k):
        raise ValueError("source and destination differ in shape")
    if not (0 <= ctrl < n and 0 <= targ < n and ctrl != targ):
        raise ValueError("invalid qubit pair")
    g = np.ascontiguousarray(gate4, dtype=np.complex128)
    if g.shape != (4, 4):
        raise ValueError("expects a 4x4 gate")
    check(_lib.lib().aqc_gate_2q(_dev(device), n, k, int(ctrl), int(targ), dptr(g), dptr(src), dptr(dst)))
    return dst


def controlled(g2x2) -> np.ndarray:
    """|0><0| x I + |1><1| x g  in the (ctrl, targ) basis."""
    m = np.eye(4, dtype=np.complex128)
    m[2:, 2:] = g2x2
    return m


def cp_derivative(angle: float) -> np.ndarray:
    """i e^{i angle} |11><11| (derv_cphase_mul_vec, core_operations.py:561-603)."""
    m = np.zeros((4, 4), dtype=np.complex128)
    m[3, 3] = 1j * np.exp(1j * angle)
    return m


def block_matrix(c_mat, t_mat, g_mat, dagger: bool) -> np.ndarray:
    """c.|0><0| x t + c.|1><1| x t.g, or its horizontally flipped form (block_mul_vec, core_operations.py:393-404)."""
    c_mat, t_mat, g_mat = (np.asarray(m, dtype=np.complex128) for m in (c_mat, t_mat, g_mat))
    if dagger:
        return np.kron(_P0 @ c_mat, t_mat) + np.kron(_P1 @ c_mat, g_mat @ t_mat)
    return np.kron(c_mat @ _P0, t_mat) + np.kron(c_mat @ _P1, t_mat @ g_mat)


def dot(kind: int, q0: int, q1: int, w: np.ndarray, z: np.ndarray, device: Optional[int] = None) -> np.complex128:
    """kind 0/1/2: 0.5j <P w|z> with P = X/Y/Z on qubit q0; kind 3: -1j <P11(q0, q1) w|z>."""
    n, k = shape_of(w)
    if z.shape != w.shape or shape_of(z) != (n, k):
        raise ValueError("w and z differ in shape")
    if not 0 <= q0 < n or (kind == 3 and not (0 <= q1 < n and q1 != q0)):
        raise ValueError("qubit out of range")
    out = np.empty(1, dtype=np.complex128)
    check(_lib.lib().aqc_gate_dot(_dev(device), n, k, int(kind), int(q0), int(q1), dptr(w), dptr(z), dptr(out)))
    return np.complex128(out[0])
