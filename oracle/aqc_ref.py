"""
ctypes front end of ``oracle/aqc_ref.c`` -- the compiled CPU restatement of the reference algorithm.

TEST INFRASTRUCTURE ONLY (see the header of ``aqc_ref.c``): checker for the HIP path at sizes where
the NumPy oracle is slow, and the CPU baseline of ``bench.py``.  Function names and argument order
follow ``oracle/aqc_oracle.py``.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

from .aqc_oracle import as_ansatz

_HERE = os.path.dirname(os.path.abspath(__file__))
_ENT = {"cx": 0, "cz": 1, "cp": 2}
_lib = None


def _cpu_tag() -> str:
    """Identifies the host CPU's instruction set: the library is built with -march=native, and a copy built in
    one container may be carried to a box with a different CPU."""
    import hashlib

    try:
        with open("/proc/cpuinfo") as f:
            flags = next((line for line in f if line.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()[:16]


def ensure_built() -> str:
    """Builds ``libaqc_ref.so`` (gcc is part of the image) unless an up-to-date copy made on this CPU exists."""
    path = os.path.join(_HERE, "libaqc_ref.so")
    src = os.path.join(_HERE, "aqc_ref.c")
    stamp = os.path.join(_HERE, "libaqc_ref.stamp")
    tag = _cpu_tag()
    try:
        with open(stamp) as f:
            same_cpu = f.read().strip() == tag
    except OSError:
        same_cpu = False
    if not (same_cpu and os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(src)):
        subprocess.run(["make", "-B", "-C", _HERE], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(tag + "\n")
    return path


def lib() -> ctypes.CDLL:
    """Loads ``libaqc_ref.so``, building it first if needed."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(ensure_built())
        i32, i64, ptr = ctypes.c_int, ctypes.c_long, ctypes.c_void_p
        _lib.aqc_ref_apply.argtypes = [i32, i32, ptr, i32, i32, i32, ptr, i64, i32, ptr]
        _lib.aqc_ref_grad.argtypes = [i32, i32, ptr, i32, i32, i32, ptr, i64, ptr, ptr, i32, i32, i32, ptr]
        _lib.aqc_ref_eval_batch.argtypes = [i32, i32, ptr, i32, i32, i32, i32, ptr, ptr, i64, i32, ptr, ptr]
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _head(circ, matrix: bool = False):
    a = as_ansatz(circ)
    blocks = np.ascontiguousarray(a.blocks, dtype=np.int32)
    trotter = a.trotter and not matrix  # the matrix code has no Trotter decorations (core_op_matrix.py)
    return a, blocks, (a.n, _ENT[a.entangler], _p(blocks), a.num_blocks, int(trotter), int(trotter and a.second_order))


def _thetas(a, thetas) -> np.ndarray:
    t = np.ascontiguousarray(thetas, dtype=np.float64).ravel()
    if t.size != a.num_thetas:
        raise ValueError("wrong number of thetas")
    return t


def _apply(circ, thetas, arr, inverse: bool, matrix: bool) -> np.ndarray:
    a, blocks, head = _head(circ, matrix)
    t = _thetas(a, thetas)
    out = np.array(arr, dtype=np.complex128, order="C", copy=True)
    ncols = out.shape[1] if matrix else 1
    if out.size != a.dim * ncols:
        raise ValueError("wrong array size")
    if lib().aqc_ref_apply(*head, _p(t), ncols, int(inverse), _p(out)):
        raise ValueError("aqc_ref_apply rejected its arguments")
    return out


def v_mul_vec(circ, thetas, vec) -> np.ndarray:
    return _apply(circ, thetas, np.asarray(vec).ravel(), False, False)


def v_dagger_mul_vec(circ, thetas, vec) -> np.ndarray:
    return _apply(circ, thetas, np.asarray(vec).ravel(), True, False)


def v_mul_mat(circ, thetas, mat) -> np.ndarray:
    return _apply(circ, thetas, np.atleast_2d(mat), False, True)


def v_dagger_mul_mat(circ, thetas, mat) -> np.ndarray:
    return _apply(circ, thetas, np.atleast_2d(mat), True, True)


def _grad(circ, thetas, x, z, block_range, front_layer, matrix) -> np.ndarray:
    a, blocks, head = _head(circ, matrix)
    t = _thetas(a, thetas)
    w = np.array(x, dtype=np.complex128, order="C", copy=True)
    zz = np.array(z, dtype=np.complex128, order="C", copy=True)
    ncols = w.shape[1] if matrix else 1
    if w.shape != zz.shape or w.size != a.dim * ncols:
        raise ValueError("wrong array size")
    br = (0, a.num_blocks) if block_range is None else (int(block_range[0]), int(block_range[1]))
    grad = np.zeros(a.num_thetas, dtype=np.complex128)
    if lib().aqc_ref_grad(*head, _p(t), ncols, _p(w), _p(zz), br[0], br[1], int(bool(front_layer)), _p(grad)):
        raise ValueError("aqc_ref_grad rejected its arguments")
    return grad


def grad_of_dot_product(circ, thetas, x_vec, vh_y_vec, block_range: Optional[Tuple[int, int]] = None,
                        front_layer: bool = True) -> np.ndarray:
    return _grad(circ, thetas, np.asarray(x_vec).ravel(), np.asarray(vh_y_vec).ravel(), block_range, front_layer, False)


def grad_of_matrix_dot_product(circ, thetas, x_mat, vh_y_mat) -> np.ndarray:
    return _grad(circ, thetas, np.atleast_2d(x_mat), np.atleast_2d(vh_y_mat), None, True, True)


def eval_batch(circ, thetas, target, x_index: int = 0, threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """B objective+gradient evaluations (theta rows) on ``threads`` cores: returns (hs[B], grads[B][T])
    with hs[b] = <x|V(theta_b)^H|target> and grads[b] the complex gradient of <V x|target>."""
    a, blocks, head = _head(circ)
    t = np.ascontiguousarray(thetas, dtype=np.float64)
    t = t.reshape(-1, a.num_thetas)
    y = np.ascontiguousarray(target, dtype=np.complex128).ravel()
    if y.size != a.dim:
        raise ValueError("wrong target size")
    hs = np.zeros(t.shape[0], dtype=np.complex128)
    grads = np.zeros((t.shape[0], a.num_thetas), dtype=np.complex128)
    if lib().aqc_ref_eval_batch(*head, t.shape[0], _p(t), _p(y), int(x_index), int(threads), _p(hs), _p(grads)):
        raise ValueError("aqc_ref_eval_batch failed")
    return hs, grads


def coord_descent_sweeps(circ, thetas, targets, nsweeps: int = 1, threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """``nsweeps`` consecutive coordinate-descent sweeps (core_op_matrix.py:765-917) per lane: thetas (B, T) or (T,), targets
    (B, d, d) or one shared (d, d).  Returns (updated thetas copy, fobj[B][nsweeps]); the inputs are not modified."""
    a, blocks, head = _head(circ, True)
    if a.entangler == "cp":
        raise NotImplementedError("CPhase entangler is not supported yet")
    t = np.array(thetas, dtype=np.float64, order="C", copy=True).reshape(-1, a.num_thetas)
    u = np.ascontiguousarray(targets, dtype=np.complex128)
    shared = u.ndim == 2
    if u.shape[-2:] != (a.dim, a.dim) or (not shared and u.shape[0] != t.shape[0]):
        raise ValueError("wrong target shape")
    fobj = np.zeros((t.shape[0], int(nsweeps)))
    lb = lib()
    i32, ptr = ctypes.c_int, ctypes.c_void_p
    lb.aqc_ref_cd_sweeps.argtypes = [i32, i32, ptr, i32, i32, ptr, ptr, i32, i32, i32, ptr]
    if lb.aqc_ref_cd_sweeps(a.n, _ENT[a.entangler], _p(blocks), a.num_blocks, t.shape[0], _p(t), _p(u), int(shared), int(nsweeps),
                            int(threads), _p(fobj)):
        raise ValueError("aqc_ref_cd_sweeps failed")
    return t, fobj
