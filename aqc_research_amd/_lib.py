"""
ctypes binding of libaqc_hip.so (C ABI: include/aqc_hip.h).

The library is the only compute path of this package.  If it is missing or no
AMD GPU is usable, calls raise -- there is no CPU fallback.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AQC_HIP_LIB", os.path.join(_HERE, "libaqc_hip.so"))

BUF_Y, BUF_Z, BUF_X, BUF_W, BUF_ZW, BUF_X2 = range(6)
K_APPLY, K_SWEEP, K_COEF, K_FINALIZE, K_MISC, K_SWEEP_LIST, K_APPLY_LIST, K_PROJECT, K_SWEEP_VIRTUAL, K_APPLY_VIRTUAL = range(10)
ENTANGLERS = {"cx": 0, "cz": 1, "cp": 2}

_P = c_void_p
_D = POINTER(c_double)

# name -> (restype, argtypes); mirrors include/aqc_hip.h line by line
SIGNATURES = {
    "aqc_version": (c_char_p, []),
    "aqc_last_error": (c_char_p, []),
    "aqc_device_count": (c_int, []),
    "aqc_create": (c_int, [c_int, c_int, POINTER(c_int32), c_int, c_int, c_int, POINTER(_P)]),
    "aqc_destroy": (c_int, [_P]),
    "aqc_num_thetas": (c_int, [_P]),
    "aqc_num_gate_groups": (c_int, [_P]),
    "aqc_v_mul_vec": (c_int, [_P, _D, _D, _D]),
    "aqc_vdag_mul_vec": (c_int, [_P, _D, _D, _D]),
    "aqc_grad_dot_vec": (c_int, [_P, _D, _D, _D, c_int, c_int, c_int, _D]),
    "aqc_v_mul_mat": (c_int, [_P, _D, _D, c_int]),
    "aqc_vdag_mul_mat": (c_int, [_P, _D, _D, c_int]),
    "aqc_grad_dot_mat": (c_int, [_P, _D, _D, _D, c_int, _D]),
    "aqc_ws_create": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, POINTER(_P)]),
    "aqc_ws_destroy": (c_int, [_P]),
    "aqc_ws_set_thetas": (c_int, [_P, _D]),
    "aqc_ws_upload": (c_int, [_P, c_int, _D]),
    "aqc_ws_upload_lane": (c_int, [_P, c_int, c_int, _D]),
    "aqc_ws_broadcast": (c_int, [_P, c_int, _D]),
    "aqc_ws_download": (c_int, [_P, c_int, _D]),
    "aqc_ws_download_lane": (c_int, [_P, c_int, c_int, _D]),
    "aqc_ws_copy_lane": (c_int, [_P, c_int, c_int, _P, c_int, c_int]),
    "aqc_ws_set_basis": (c_int, [_P, c_int, POINTER(c_int64)]),
    "aqc_ws_set_combo": (c_int, [_P, c_int, POINTER(c_int64), _D]),
    "aqc_ws_set_identity": (c_int, [_P, c_int]),
    "aqc_ws_apply": (c_int, [_P, c_int, c_int, c_int]),
    "aqc_ws_grad": (c_int, [_P, c_int, c_int, c_int]),
    "aqc_ws_get_grads": (c_int, [_P, _D]),
    "aqc_ws_gather": (c_int, [_P, c_int, POINTER(c_int64), c_int, _D]),
    "aqc_ws_vdot": (c_int, [_P, c_int, c_int, _D]),
    "aqc_ws_sync": (c_int, [_P]),
    # array arguments as void*: the single-evaluation path passes raw addresses (ndarray.ctypes.data: 0.9 us instead of the
    # 1.9 us of data_as per argument, three arguments per call)
    "aqc_ws_eval": (c_int, [_P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "aqc_ws_grad_from": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "aqc_ws_cd_sweep": (c_int, [_P, _D, _D]),
    "aqc_ws_cd_sweeps": (c_int, [_P, _D, _D, c_int, c_int]),
    "aqc_ws_cd_fits_one_launch": (c_int, [_P]),
    "aqc_zgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, _D, c_int, _D, c_int, _D, c_int]),
    "aqc_gate_1q": (c_int, [c_int, c_int, c_int64, c_int, _D, _D, _D]),
    "aqc_gate_2q": (c_int, [c_int, c_int, c_int64, c_int, c_int, _D, _D, _D]),
    "aqc_mps_create": (c_int, [c_int, c_int, POINTER(c_int32), _D, _D, POINTER(_P)]),
    "aqc_mps_destroy": (c_int, [_P]),
    "aqc_mps_clone": (c_int, [_P, POINTER(_P)]),
    "aqc_mps_num_qubits": (c_int, [_P]),
    "aqc_mps_dims": (c_int, [_P, POINTER(c_int32)]),
    "aqc_mps_discarded_weight": (c_double, [_P]),
    "aqc_mps_export": (c_int, [_P, _D, _D]),
    "aqc_mps_gate1": (c_int, [_P, c_int, _D]),
    "aqc_mps_gate2": (c_int, [_P, c_int, c_int, _D, c_double, c_int]),
    "aqc_mps_dot": (c_int, [_P, _P, _D]),
    "aqc_mps_dot_ops": (c_int, [_P, _P, c_int, POINTER(c_int32), _D, _D]),
    "aqc_svd": (c_int, [c_int, c_int, c_int, _D, _D, _D, _D, POINTER(c_int)]),
    "aqc_mps_apply_circuit": (c_int, [_P, _P, _D, c_int, c_double, c_int]),
    "aqc_mps_fast_dot_gradient": (c_int, [_P, _P, _P, _D, c_double, c_int, c_int, c_int, c_int, _D]),
    "aqc_mpsb_create": (c_int, [c_int, c_int, c_int, POINTER(_P)]),
    "aqc_mpsb_destroy": (c_int, [_P]),
    "aqc_mpsb_gate2_stats": (c_int, [_P, c_int, _D, c_int]),
    "aqc_mpsb_set_targets": (c_int, [_P, POINTER(_P), c_int]),
    "aqc_mpsb_set_lhs": (c_int, [_P, POINTER(_P), c_int]),
    "aqc_mpsb_eval": (c_int, [_P, _P, _D, c_double, c_int, c_int, c_int, c_int, _D, _D, _D, POINTER(c_int32)]),
    "aqc_mpsb_set_lhs_basis": (c_int, [_P, POINTER(c_uint8)]),
    "aqc_mpsb_vh": (c_int, [_P, _P, _D, c_double, c_int, c_int, c_int, _D, _D, POINTER(c_int32)]),
    "aqc_mpsb_grad": (c_int, [_P, _P, c_int, c_int, c_int, _D]),
    "aqc_mpsb_gradient_of": (c_int, [_P, _P, _D, c_double, c_int, c_int, c_int, c_int, _D]),
    "aqc_mpsb_apply_circuit": (c_int, [_P, _P, _D, c_int, c_double, c_int, _D, POINTER(c_int32)]),
    "aqc_mpsb_export": (c_int, [_P, c_int, POINTER(_P)]),
    "aqc_mps_device": (c_int, [_P]),
    "aqc_gate_dot": (c_int, [c_int, c_int, c_int64, c_int, c_int, c_int, _D, _D, _D]),
    "aqc_ws_mps_upload": (c_int, [_P, c_int, POINTER(c_int32), _D, _D]),
    "aqc_ws_mps_to_vec": (c_int, [_P, c_int, c_int, c_int]),
    "aqc_ws_mps_to_vec_batch": (c_int, [_P, c_int, POINTER(c_int32), c_int, POINTER(c_int32)]),
    "aqc_ws_mps_dot": (c_int, [_P, c_int, c_int, _D]),
    "aqc_ws_theta_bank": (c_int, [_P, _D, c_int]),
    "aqc_ws_use_theta_set": (c_int, [_P, c_int]),
    "aqc_ws_gather_setup": (c_int, [_P, POINTER(c_int64), c_int]),
    "aqc_ws_gather_launch": (c_int, [_P, c_int]),
    "aqc_ws_gather_fetch": (c_int, [_P, _D]),
    "aqc_ws_vdot_launch": (c_int, [_P, c_int, c_int]),
    "aqc_ws_vdot_fetch": (c_int, [_P, _D]),
    "aqc_ws_results_async": (c_int, [_P]),
    "aqc_ws_objective_launch": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "aqc_ws_results_fetch": (c_int, [_P, _D, _D]),
    "aqc_ws_timer_start": (c_int, [_P]),
    "aqc_ws_timer_stop": (c_int, [_P, POINTER(c_float)]),
    "aqc_ws_profile_enable": (c_int, [_P, c_int]),
    "aqc_ws_profile_get": (c_int, [_P, c_int, POINTER(c_int64), POINTER(c_double)]),
    "aqc_ws_profile_reset": (c_int, [_P]),
    "aqc_ws_profile_log": (c_int, [_P, POINTER(c_int32), _D, c_int, POINTER(c_int)]),
    "aqc_ws_plan_stage": (c_int, [_P, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "aqc_ws_sparse_counts": (c_int, [_P, POINTER(c_int64)]),
    "aqc_ws_projected_info": (c_int, [_P, POINTER(c_int32)]),
    "aqc_plan_projected": (c_int, [_P, c_int, c_int, POINTER(c_int32)]),
    "aqc_ws_plan_skips": (c_int, [_P, c_int, c_int, POINTER(c_int), c_int]),
    "aqc_ws_sweep_r_only_sub": (c_int, [_P]),
    "aqc_ws_plan_info": (c_int, [_P, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "aqc_ws_kernel_family": (c_int, [_P, c_int]),
    "aqc_ws_plan_substages": (c_int, [_P, c_int]),
    "aqc_ws_lbfgs": (c_int, [_P, _D, c_int, c_int, c_double, c_double, c_double, c_int, c_int, c_int, c_int, _D, _D, _D, POINTER(c_int64),
                     POINTER(c_int64), _D, POINTER(c_int64)]),
    "aqc_ws_surrogate_eval": (c_int, [_P, _D, c_int, _D, POINTER(c_int64), c_int, c_int, c_int, _D, _D, _D, _D, _D]),
    "aqc_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "aqc_comm_create": (c_int, [ctypes.c_char_p, c_int, c_int, c_int, POINTER(_P)]),
    "aqc_comm_destroy": (c_int, [_P]),
    "aqc_comm_rank": (c_int, [_P]),
    "aqc_comm_size": (c_int, [_P]),
    "aqc_comm_allgather": (c_int, [_P, _D, _D, ctypes.c_size_t]),
    "aqc_comm_allreduce": (c_int, [_P, _D, ctypes.c_size_t, c_int]),
    "aqc_comm_barrier": (c_int, [_P]),
    "aqc_plan_substages": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int]),
    "aqc_plan_query": (
        c_int,
        [_P, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    ),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Loads the shared library once (CDLL => the GIL is released during calls)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C aqc_research_amd/csrc` "
                "(or __graft_entry__.build()); this package has no CPU fallback"
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError => ABI mismatch, fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(status: int) -> None:
    if status != 0:
        raise RuntimeError("aqc_hip: " + lib().aqc_last_error().decode("utf-8", "replace"))


def dptr(arr: np.ndarray):
    """Pointer to the (re, im)-interleaved doubles of a float64/complex128 array."""
    return arr.ctypes.data_as(_D)


def as_c128(a, shape=None, name="array") -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {a.shape}")
    return a


def as_f64(a, size=None, name="array") -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if size is not None and a.size != size:
        raise ValueError(f"{name}: expected {size} values, got {a.size}")
    return a
