"""
State-vector entry points with the reference signatures (core_operations.py:606,713,823),
executed by the HIP kernels.  ``workspace`` arguments are accepted for drop-in
compatibility and validated for overlap, but the device keeps its own scratch.
"""
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .engine import BUF_X, BUF_Y, BUF_Z, HipContext


def bit2bit_transform(n: int, i: int) -> int:
    """core_operations.py:34-43 (the HIP path itself works in Qiskit order)."""
    return n - 1 - i


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
