"""
MPS twin of the gradient sweep with the reference signature (mps_dot_objective.py:41-242):
both MPS operands are contracted to dense states on the device and the fused state-vector
sweep runs on them -- the no-truncation semantics the reference's tests pin
(test_mps_fast_dot_gradient.py:126-153).
"""
from typing import Optional, Tuple

import numpy as np

from .engine import BUF_X, BUF_Z, HipContext
from .mps_operations import check_mps, no_truncation_threshold


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
    th = np.asarray(thetas, dtype=np.float64)
    if th.ndim != 1 or th.size != circ.num_thetas:
        raise ValueError("thetas: expects a float vector of size circ.num_thetas")
    if not (isinstance(trunc_thr, float) and trunc_thr >= 0):
        raise ValueError("trunc_thr must be a non-negative float")
    if block_range is not None and not (isinstance(block_range, tuple) and len(block_range) == 2
                                        and 0 <= block_range[0] < block_range[1] <= circ.num_blocks):
        raise ValueError("block_range must be a tuple (from, to) with 0 <= from < to <= num_blocks")
    ws = HipContext.of(circ).workspace(1, 1)
    ws.set_thetas(th)
    ws.mps_upload(0, lvec)
    ws.mps_upload(1, vh_phi)
    ws.mps_to_vec(0, BUF_X, 0)
    ws.mps_to_vec(1, BUF_Z, 0)
    ws.grad(block_range, bool(front_layer))
    return ws.get_grads()[0]
