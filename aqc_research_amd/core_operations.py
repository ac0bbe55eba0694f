"""
State-vector entry points with the reference signatures (core_operations.py:606,713,823),
executed by the HIP kernels.  ``workspace`` arguments are accepted for drop-in
compatibility and validated for overlap, but the device keeps its own scratch.
"""
from typing import Optional, Tuple

import numpy as np

from . import gates
from .engine import BUF_X, BUF_Y, BUF_Z, HipContext


def bit2bit_transform(n: int, i: int) -> int:
    """core_operations.py:34-43 (the HIP path itself works in Qiskit order)."""
    return n - 1 - i


# ---- gate-level building blocks (core_operations.py:46-603).  ``pos`` / ``c`` / ``t`` are the reference's
# big-endian positions (h = 2^(n-1-pos), :77); ``temp`` / workspace arguments are accepted and left untouched.

def _vec_ok(n: int, *arrs) -> None:
    for a in arrs:
        if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and a.shape == (2**n,) and a.flags.c_contiguous):
            raise ValueError("expects contiguous complex128 vectors of size 2^n")


def _pos_ok(n: int, *pos) -> None:
    if not all(isinstance(p, (int, np.integer)) and 0 <= p < n for p in pos) or len(set(pos)) != len(pos):
        raise ValueError("qubit position out of range")


def gate2x2_mul_vec(num_qubits: int, pos: int, gate2x2: np.ndarray, vec: np.ndarray, out: np.ndarray, inplace: bool) -> np.ndarray:
    """(I x g x I) vec, formed in ``vec`` (inplace) or in ``out`` (core_operations.py:46-119)."""
    _vec_ok(num_qubits, vec, out)
    _pos_ok(num_qubits, pos)
    if np.may_share_memory(vec, out):
        raise ValueError("vec and out must not overlap")
    if not isinstance(inplace, (bool, np.bool_)):
        raise TypeError("inplace must be bool")
    return gates.apply_1q(gate2x2, num_qubits - 1 - pos, vec, vec if inplace else out)


def proj00_mul_vec(num_qubits: int, pos: int, vec: np.ndarray) -> np.ndarray:
    """|0><0| on one qubit, in place (core_operations.py:122-140)."""
    _vec_ok(num_qubits, vec)
    _pos_ok(num_qubits, pos)
    return gates.apply_1q(gates._P0, num_qubits - 1 - pos, vec, vec)


def proj11_mul_vec(num_qubits: int, pos: int, vec: np.ndarray) -> np.ndarray:
    """|1><1| on one qubit, in place (core_operations.py:143-161)."""
    _vec_ok(num_qubits, vec)
    _pos_ok(num_qubits, pos)
    return gates.apply_1q(gates._P1, num_qubits - 1 - pos, vec, vec)


def _rot(matrix_of, n, pos, angle, vec):
    _vec_ok(n, vec)
    _pos_ok(n, pos)
    return gates.apply_1q(matrix_of(float(angle)), n - 1 - pos, vec, vec)


def rx_mul_vec(n: int, pos: int, angle: float, vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.ndarray:
    """Rx(angle) on one qubit, in place (core_operations.py:164-197)."""
    return _rot(gates.rx_matrix, n, pos, angle, vec)


def ry_mul_vec(n: int, pos: int, angle: float, vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.ndarray:
    """Ry(angle) on one qubit, in place (core_operations.py:200-233)."""
    return _rot(gates.ry_matrix, n, pos, angle, vec)


def rz_mul_vec(n: int, pos: int, angle: float, vec: np.ndarray, _: Optional[np.ndarray] = None) -> np.ndarray:
    """Rz(angle) on one qubit, in place (core_operations.py:236-264)."""
    return _rot(gates.rz_matrix, n, pos, angle, vec)


def _dot(kind, n, pos, w_vec, z_vec):
    _vec_ok(n, w_vec, z_vec)
    _pos_ok(n, pos)
    return gates.dot(kind, n - 1 - pos, -1, w_vec, z_vec)


def dot_x(n: int, pos: int, w_vec: np.ndarray, z_vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <X w|z> (core_operations.py:267-293)."""
    return _dot(0, n, pos, w_vec, z_vec)


def dot_y(n: int, pos: int, w_vec: np.ndarray, z_vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <Y w|z> (core_operations.py:296-322)."""
    return _dot(1, n, pos, w_vec, z_vec)


def dot_z(n: int, pos: int, w_vec: np.ndarray, z_vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <Z w|z> (core_operations.py:325-351)."""
    return _dot(2, n, pos, w_vec, z_vec)


def block_mul_vec(n: int, c: int, t: int, c_mat: np.ndarray, t_mat: np.ndarray, g_mat: np.ndarray, vec: np.ndarray,
                  workspace: Optional[np.ndarray], dagger: bool) -> np.ndarray:
    """vec <- (unit-block) vec in place; ``dagger`` only flips the block structure (core_operations.py:354-419)."""
    _vec_ok(n, vec)
    _pos_ok(n, c, t)
    for m in (c_mat, t_mat, g_mat):
        if np.shape(m) != (2, 2):
            raise ValueError("expects 2x2 matrices")
    _no_overlap(vec, workspace, "vec")
    return gates.apply_2q(gates.block_matrix(c_mat, t_mat, g_mat, bool(dagger)), n - 1 - c, n - 1 - t, vec, vec)


def _ent(g2x2, n, c, t, vec):
    _vec_ok(n, vec)
    _pos_ok(n, c, t)
    return gates.apply_2q(gates.controlled(g2x2), n - 1 - c, n - 1 - t, vec, vec)


def cx_mul_vec(n: int, c: int, t: int, _: float, vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.ndarray:
    """CNOT in place (core_operations.py:422-465)."""
    return _ent([[0, 1], [1, 0]], n, c, t, vec)


def cz_mul_vec(n: int, c: int, t: int, _: float, vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.ndarray:
    """CZ in place (core_operations.py:468-511)."""
    return _ent([[1, 0], [0, -1]], n, c, t, vec)


def cp_mul_vec(n: int, c: int, t: int, angle: float, vec: np.ndarray, temp: Optional[np.ndarray] = None) -> np.ndarray:
    """CPhase(angle) in place (core_operations.py:514-558)."""
    return _ent([[1, 0], [0, np.exp(1j * float(angle))]], n, c, t, vec)


def derv_cphase_mul_vec(n: int, c: int, t: int, angle: float, vec: np.ndarray, out: np.ndarray) -> np.ndarray:
    """out <- d/d(angle) CPhase(angle) vec = i e^{i angle} |11><11| vec (core_operations.py:561-603)."""
    _vec_ok(n, vec, out)
    _pos_ok(n, c, t)
    if np.may_share_memory(vec, out):
        raise ValueError("vec and out must not overlap")
    return gates.apply_2q(gates.cp_derivative(float(angle)), n - 1 - c, n - 1 - t, vec, out)


def _check_vec(circ, a: np.ndarray, name: str) -> None:
    if not (isinstance(a, np.ndarray) and a.dtype == np.complex128 and a.shape == (circ.dimension,) and a.flags.c_contiguous):
        raise ValueError(f"{name}: expects a contiguous complex128 vector of size 2^n")


def _check_thetas(circ, thetas) -> np.ndarray:
    th = np.asarray(thetas)
    if th.ndim != 1 or th.size != circ.num_thetas or not np.issubdtype(th.dtype, np.floating):
        raise ValueError("thetas: expects a float vector of size circ.num_thetas")
    return np.ascontiguousarray(th, dtype=np.float64)


def _no_overlap(a, b, what: str) -> None:
    if b is not None and isinstance(b, np.ndarray) and np.may_share_memory(a, b):
        raise ValueError(f"{what} must not overlap the workspace")


def _apply(circ, thetas, vec, out, workspace, inverse: bool) -> np.ndarray:
    th = _check_thetas(circ, thetas)
    _check_vec(circ, vec, "vec")
    _check_vec(circ, out, "out")
    _no_overlap(vec, workspace, "vec")
    _no_overlap(out, workspace, "out")
    ws = HipContext.of(circ).workspace(1, 1)
    ws.set_thetas(th)
    ws.upload(BUF_Y, vec)
    ws.apply(inverse, BUF_Y, BUF_Z)
    ws.download(BUF_Z, lane=0, out=out)  # out may alias vec (test_core_operations.py:270)
    return out


def v_mul_vec(circ, thetas: np.ndarray, vec: np.ndarray, out: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """out = V(thetas) @ vec (core_operations.py:606); returns ``out``."""
    return _apply(circ, thetas, vec, out, workspace, False)


def v_dagger_mul_vec(circ, thetas: np.ndarray, vec: np.ndarray, out: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """out = V(thetas)^H @ vec (core_operations.py:713); returns ``out``."""
    return _apply(circ, thetas, vec, out, workspace, True)


def grad_of_dot_product(
    circ,
    thetas: np.ndarray,
    x_vec: np.ndarray,
    vh_y_vec: np.ndarray,
    workspace: Optional[np.ndarray] = None,
    block_range: Optional[Tuple[int, int]] = None,
    front_layer: bool = True,
) -> np.ndarray:
    """Complex gradient of <V x|y> given vh_y = V^H y (core_operations.py:823); the
    inputs are left intact."""
    th = _check_thetas(circ, thetas)
    _check_vec(circ, x_vec, "x_vec")
    _check_vec(circ, vh_y_vec, "vh_y_vec")
    _no_overlap(x_vec, workspace, "x_vec")
    _no_overlap(vh_y_vec, workspace, "vh_y_vec")
    if not isinstance(front_layer, (bool, np.bool_)):
        raise TypeError("front_layer must be bool")
    if block_range is not None:
        if not (isinstance(block_range, tuple) and len(block_range) == 2
                and 0 <= block_range[0] < block_range[1] <= circ.num_blocks):
            raise ValueError("block_range must be a tuple (from, to) with 0 <= from < to <= num_blocks")
    ws = HipContext.of(circ).workspace(1, 1)
    ws.set_thetas(th)
    ws.upload(BUF_X, x_vec)
    ws.upload(BUF_Z, vh_y_vec)
    ws.grad(block_range, bool(front_layer))
    return ws.get_grads()[0]
