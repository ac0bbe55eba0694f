"""
Matrix entry points with the reference signatures (core_op_matrix.py:480,562,645,765),
executed by the HIP kernels.  A row-major (2^n, k) matrix is a flat array in which qubit q
has stride k*2^q, so the same fused stage kernels run with the column index as extra
low-order ("batch") address bits; k is padded to a power of two on the device.
"""
from typing import Optional

import numpy as np

from . import gates
from .engine import BUF_X, BUF_Y, BUF_Z, HipContext
from .parametric_circuit import TrotterAnsatz


# ---- gate-level building blocks (core_op_matrix.py:32-477): qubit numbers are plain bit indices of the row,
# matrices are (2^n, k) with k <= 2^n, modified in place; workspace arguments are accepted and left untouched.

def _mat_ok(*mats) -> int:
    n = -1
    for m in mats:
        if not (isinstance(m, np.ndarray) and m.ndim == 2 and m.dtype == np.complex128 and m.flags.c_contiguous and m.shape[1] <= m.shape[0]):
            raise ValueError("expects C-contiguous complex128 matrices of shape (2^n, k), k <= 2^n")
        n = gates.shape_of(m)[0]
    if any(m.shape != mats[0].shape for m in mats):
        raise ValueError("matrices differ in shape")
    return n


def gate2x2_mul_mat(qubit_no: int, gate2x2: np.ndarray, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """mat <- (I x g x I) mat (core_op_matrix.py:392-427)."""
    _mat_ok(mat)
    return gates.apply_1q(gate2x2, int(qubit_no), mat, mat)


def rx_mul_mat(angle: float, qubit_no: int, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:32-63."""
    _mat_ok(mat)
    return gates.apply_1q(gates.rx_matrix(float(angle)), int(qubit_no), mat, mat)


def ry_mul_mat(angle: float, qubit_no: int, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:66-97."""
    _mat_ok(mat)
    return gates.apply_1q(gates.ry_matrix(float(angle)), int(qubit_no), mat, mat)


def rz_mul_mat(angle: float, qubit_no: int, mat: np.ndarray, ___: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:100-127."""
    _mat_ok(mat)
    return gates.apply_1q(gates.rz_matrix(float(angle)), int(qubit_no), mat, mat)


def cx_mul_mat(ctrl: int, targ: int, ___: float, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:130-178."""
    _mat_ok(mat)
    return gates.apply_2q(gates.controlled([[0, 1], [1, 0]]), int(ctrl), int(targ), mat, mat)


def cz_mul_mat(ctrl: int, targ: int, ___: float, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:181-229."""
    _mat_ok(mat)
    return gates.apply_2q(gates.controlled([[1, 0], [0, -1]]), int(ctrl), int(targ), mat, mat)


def cp_mul_mat(ctrl: int, targ: int, angle: float, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """core_op_matrix.py:232-281."""
    _mat_ok(mat)
    return gates.apply_2q(gates.controlled([[1, 0], [0, np.exp(1j * float(angle))]]), int(ctrl), int(targ), mat, mat)


def x_dot_mat(qubit_no: int, w_mat: np.ndarray, z_mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <X w|z>_F (core_op_matrix.py:284-317)."""
    _mat_ok(w_mat, z_mat)
    return gates.dot(0, int(qubit_no), -1, w_mat, z_mat)


def y_dot_mat(qubit_no: int, w_mat: np.ndarray, z_mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <Y w|z>_F (core_op_matrix.py:320-353)."""
    _mat_ok(w_mat, z_mat)
    return gates.dot(1, int(qubit_no), -1, w_mat, z_mat)


def z_dot_mat(qubit_no: int, w_mat: np.ndarray, z_mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.complex128:
    """0.5j <Z w|z>_F (core_op_matrix.py:356-389)."""
    _mat_ok(w_mat, z_mat)
    return gates.dot(2, int(qubit_no), -1, w_mat, z_mat)


def derv_cphase(ctrl: int, targ: int, w_mat: np.ndarray, z_mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.complex128:
    """Derivative of <w|z> by the CPhase angle, taken before the gate: -1j <P11 w|z>_F (core_op_matrix.py:430-477)."""
    _mat_ok(w_mat, z_mat)
    return gates.dot(3, int(ctrl), int(targ), w_mat, z_mat)


def _check(circ, thetas, mat: np.ndarray, name: str) -> np.ndarray:
    if isinstance(circ, TrotterAnsatz):
        raise ValueError("the matrix path takes a plain ParametricCircuit (core_op_matrix.py:480-559)")
    th = np.asarray(thetas)
    if th.ndim != 1 or th.size != circ.num_thetas or not np.issubdtype(th.dtype, np.floating):
        raise ValueError("thetas: expects a float vector of size circ.num_thetas")
    if not (isinstance(mat, np.ndarray) and mat.dtype == np.complex128 and mat.ndim == 2 and mat.flags.c_contiguous):
        raise ValueError(f"{name}: expects a C-contiguous complex128 matrix")
    if mat.shape[0] != circ.dimension or mat.shape[1] > mat.shape[0] or mat.shape[1] < 1:
        raise ValueError(f"{name}: expects shape (2^n, k) with 1 <= k <= 2^n")
    return np.ascontiguousarray(th, dtype=np.float64)


def _no_overlap(a, workspace) -> None:
    if isinstance(workspace, np.ndarray) and np.may_share_memory(a, workspace):
        raise ValueError("matrix must not overlap the workspace")


def _apply(circ, thetas, mat, workspace, inverse: bool) -> np.ndarray:
    th = _check(circ, thetas, mat, "mat")
    _no_overlap(mat, workspace)
    ws = HipContext.of(circ).workspace(1, mat.shape[1])
    ws.set_thetas(th)
    ws.upload(BUF_Y, mat)
    ws.apply(inverse, BUF_Y, BUF_Z)
    ws.download(BUF_Z, lane=0, out=mat)
    return mat


def v_mul_mat(circ, thetas: np.ndarray, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """mat <- V(thetas) @ mat in place; returns ``mat`` (core_op_matrix.py:480)."""
    return _apply(circ, thetas, mat, workspace, False)


def v_dagger_mul_mat(circ, thetas: np.ndarray, mat: np.ndarray, workspace: Optional[np.ndarray] = None) -> np.ndarray:
    """mat <- V(thetas)^H @ mat in place; returns ``mat`` (core_op_matrix.py:562)."""
    return _apply(circ, thetas, mat, workspace, True)


def grad_of_matrix_dot_product(
    circ, thetas: np.ndarray, x_mat: np.ndarray, vh_y_mat: np.ndarray, workspace: Optional[np.ndarray] = None
) -> np.ndarray:
    """Complex gradient of <V X|Y>_F given vh_y = V^H Y (core_op_matrix.py:645).  The
    reference clobbers both inputs; here they are left intact (their final content is
    never consumed by any reference caller: sk_core.py:190-193 regenerates them)."""
    th = _check(circ, thetas, x_mat, "x_mat")
    _check(circ, thetas, vh_y_mat, "vh_y_mat")
    if x_mat.shape != vh_y_mat.shape:
        raise ValueError("x_mat and vh_y_mat must have the same shape")
    if np.may_share_memory(x_mat, vh_y_mat):
        raise ValueError("x_mat and vh_y_mat must not overlap")
    _no_overlap(x_mat, workspace)
    _no_overlap(vh_y_mat, workspace)
    ws = HipContext.of(circ).workspace(1, x_mat.shape[1])
    ws.set_thetas(th)
    ws.upload(BUF_X, x_mat)
    ws.upload(BUF_Z, vh_y_mat)
    ws.grad(None, True)
    return ws.get_grads()[0]


def coord_descent_single_sweep(circ, thetas: np.ndarray, target: np.ndarray, workspace: Optional[np.ndarray] = None) -> float:
    """One Gauss-Seidel sweep over all parameters of ``1 - |<V,U>|^2 / d^2``
    (core_op_matrix.py:765-917).  ``thetas`` is updated in place; returns the objective at the
    end of the sweep.  The whole sweep runs on the device: ONE persistent launch while the two d x d operands fit a
    workgroup's LDS (up to 6 qubits; ``aqc_ws_cd_sweeps``), a chain of 2 launches per parameter beyond."""
    from . import _lib
    from ._lib import check, dptr

    if circ.entangler == "cp":
        raise NotImplementedError("CPhase entangler is not supported yet")
    if not (isinstance(thetas, np.ndarray) and thetas.dtype == np.float64 and thetas.ndim == 1
            and thetas.size == circ.num_thetas and thetas.flags.c_contiguous):
        raise ValueError("thetas: expects a contiguous float64 vector of size circ.num_thetas (updated in place)")
    _check(circ, thetas, target, "target")
    if target.shape[0] != target.shape[1]:
        raise ValueError("target must be square")
    _no_overlap(target, workspace)
    ws = HipContext.of(circ).workspace(1, target.shape[1])
    ws.upload(BUF_Y, target)
    fobj = np.zeros(1)
    ws._touch(_lib.BUF_X, _lib.BUF_Z, _lib.BUF_W, _lib.BUF_ZW)   # rewritten by the sweep
    check(_lib.lib().aqc_ws_cd_sweep(ws.handle, dptr(thetas), dptr(fobj)))
    return float(fobj[0])


def coord_descent_sweeps(circ, thetas: np.ndarray, targets: np.ndarray, num_sweeps: int = 1, *, device: Optional[int] = None,
                         max_steps: int = -1) -> np.ndarray:
    """``num_sweeps`` consecutive ``coord_descent_single_sweep``s for every lane of a batch in ONE launch
    (core_op_matrix.py:765-917 called in a loop, docs/aqc.ipynb: 1000 sweeps): ``thetas`` (lanes, T) -- random restarts of one
    ansatz, updated in place -- and ``targets`` (lanes, d, d) or one (d, d) target shared by all lanes.  Returns the objective
    at the end of every sweep, shape (lanes, num_sweeps).  Up to 6 qubits (the operands live in LDS for the whole walk)."""
    from . import _lib
    from ._lib import check, dptr
    from .engine import Workspace

    if circ.entangler == "cp":
        raise NotImplementedError("CPhase entangler is not supported yet")
    th = thetas.reshape(1, -1) if thetas.ndim == 1 else thetas
    if not (isinstance(thetas, np.ndarray) and thetas.dtype == np.float64 and thetas.flags.c_contiguous and th.ndim == 2
            and th.shape[1] == circ.num_thetas):
        raise ValueError("thetas: expects a contiguous float64 array (lanes, circ.num_thetas), updated in place")
    d = circ.dimension
    tg = np.asarray(targets)
    if tg.dtype != np.complex128 or tg.shape[-2:] != (d, d) or tg.ndim not in (2, 3):
        raise ValueError("targets: expects complex128 (d, d) or (lanes, d, d)")
    lanes = th.shape[0]
    if tg.ndim == 3 and tg.shape[0] != lanes:
        raise ValueError("one target per lane (or one for all)")
    if int(num_sweeps) < 1:
        raise ValueError("num_sweeps must be positive")
    ws = Workspace(HipContext.of(circ), batch=lanes, ncols=d, device=device)
    try:
        ws.upload(BUF_Y, np.ascontiguousarray(np.broadcast_to(tg, (lanes, d, d))))
        fobj = np.zeros((lanes, int(num_sweeps)))
        check(_lib.lib().aqc_ws_cd_sweeps(ws.handle, dptr(th), dptr(fobj), int(num_sweeps), int(max_steps)))
    finally:
        ws.close()
    return fobj
