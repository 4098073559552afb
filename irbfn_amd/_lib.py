"""ctypes binding of libirbfn_hip.so (include/irbfn_hip.h).  No fallback: a missing library raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IRBFN_LIB") or os.path.join(_HERE, "libirbfn_hip.so")   # IRBFN_LIB: A/B builds (tools/build_variant.sh)

IRBFN_OK = 0
STATUS_NAMES = {0: "IRBFN_OK", -1: "IRBFN_ERR_BAD_ARG", -2: "IRBFN_ERR_UNSUPPORTED", -3: "IRBFN_ERR_HIP",
                -4: "IRBFN_ERR_NO_PARAMS", -5: "IRBFN_ERR_NO_DEVICE"}

BASIS_ENUM = {"gaussian": 0, "gaussian_wide": 1, "gaussian_wider": 2, "inverse_quadratic": 3, "linear": 4,
              "quadratic": 5, "multiquadric": 6, "inverse_multiquadric": 7, "spline": 8, "poisson_one": 9,
              "poisson_two": 10, "matern32": 11, "matern52": 12}

ROLLOUT_ST_SELECT, ROLLOUT_ST_KS, ROLLOUT_FULLINT, ROLLOUT_FRENET_LS, ROLLOUT_SPIRAL = 0, 1, 2, 3, 4

# irbfn_option / irbfn_fwd_kernel / irbfn_vjp_kernel (include/irbfn_hip.h)
OPTIONS = {"fwd_kernel": 0, "fwd_small": 1, "fwd_f16_terms": 2, "fwd_f16_minb": 3, "fwd_q": 4, "fwd_nw": 5, "fwd_qj": 6,
           "fwd_f16_s": 7, "fwd_f16_qg": 8, "vjp_kernel": 9, "vjp_f16_ct": 10, "lds_pad": 11, "fwd_wide_pipe": 12, "tick_fused": 13, "gram_sticky": 14, "vjp_qsb": 15}
FWD_AUTO, FWD_K1, FWD_K1M, FWD_K1H, FWD_K1R, FWD_K1G = 0, 1, 2, 3, 4, 5
VJP_AUTO, VJP_K2, VJP_K2H, VJP_K2R, VJP_K2G = 0, 1, 2, 3, 4

# every symbol include/irbfn_hip.h declares: (name, restype, argtypes)
_vp, _fp, _ip, _i, _i64, _f = C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "irbfn_net_create": (_i, [C.POINTER(C.c_void_p), _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _ip, _i]),
    "irbfn_net_destroy": (_i, [_vp]),
    "irbfn_net_set_params": (_i, [_vp, _fp, _fp, _fp, _fp, _vp]),
    "irbfn_net_set_option": (_i, [_vp, _i, _i]),
    "irbfn_net_get_option": (_i, [_vp, _i, C.POINTER(_i)]),
    "irbfn_net_forward": (_i, [_vp, _fp, _fp, _i64, _vp]),
    "irbfn_net_gate": (_i, [_vp, _fp, _fp, _i64, _vp]),
    "irbfn_net_vjp_workspace_bytes": (_i64, [_vp, _i64]),
    "irbfn_net_vjp": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _vp, _i64, _vp]),
    "irbfn_rollout_state_dim": (_i, [_i]),
    "irbfn_rollout_input_dim": (_i, [_i, _i]),
    "irbfn_rollout_forward": (_i, [_i, _fp, _fp, _fp, _i64, _i, _vp]),
    "irbfn_rollout_vjp": (_i, [_i, _fp, _fp, _fp, _fp, _i64, _i, _f, _vp]),
    "irbfn_net_forward_rollout": (_i, [_vp, _i, _fp, _fp, _fp, _fp, _fp, _i64, _i, _vp]),
    "irbfn_train_loss_partials": (_i, []),
    "irbfn_train_seeds_oneint": (_i, [_fp, _fp, _fp, _fp, _f, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "irbfn_train_seeds_fullint": (_i, [_fp, _fp, _fp, _f, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "irbfn_train_seeds_frenet_fullint": (_i, [_fp, _fp, _fp, _fp, _f, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "irbfn_adam_clip_step": (_i, [_fp, _fp, _fp, _fp, _i64, _vp, _f, _f, _f, _f, _f, _fp, _vp]),
    "irbfn_plan_queries_cartesian": (_i, [_fp, _fp, _fp, _fp, _ip, _i64, _vp]),
    "irbfn_plan_queries_frenet": (_i, [_fp, _fp, _fp, _fp, _ip, _i64, _vp]),
    "irbfn_plan_tick": (_i, [_vp, _i, _fp, _ip, _fp, _fp, _fp, _fp, _i64, _i, _vp]),
    "irbfn_lut_grid_lookup": (_i, [_fp, _ip, _ip, _fp, _fp, _ip, _fp, _i64, _i, _i, _vp]),
    "irbfn_lut_nearest_workspace_bytes": (_i64, [_i64, _i64]),
    "irbfn_lut_nearest": (_i, [_fp, _fp, _fp, _ip, _fp, _fp, _i64, _i64, _i, _i, _vp, _i64, _vp]),
    "irbfn_nearest_point": (_i, [_fp, _fp, _fp, _fp, _fp, _ip, _i64, _i, _vp]),
    "irbfn_intersect_point": (_i, [_fp, _fp, _fp, _f, _i, _fp, _ip, _fp, _ip, _i64, _i, _vp]),
    "irbfn_cluster_gate": (_i, [_fp, _fp, _fp, _fp, _fp, _i64, _i, _i, _vp]),
    "irbfn_net_tick_needs_controls": (_i, [_vp, _i, _i64, _i]),
    "irbfn_net_forward_gamma": (_i, [_vp, _fp, _fp, _fp, _i64, _vp]),
    "irbfn_net_vjp_gamma": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _vp, _i64, _vp]),
    "irbfn_cluster_gate_vjp_workspace_bytes": (_i64, [_i, _i]),
    "irbfn_cluster_gate_vjp": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _i, _i, _vp, _i64, _vp]),
    "irbfn_softmax_xent": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i64, _i, _vp]),
    "irbfn_mlp_head_forward": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i64, _i, _i, _i, _vp]),
    "irbfn_mlp_head_vjp_workspace_bytes": (_i64, [_i, _i, _i]),
    "irbfn_mlp_head_vjp": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "irbfn_f64_workspace_bytes": (_i64, [_vp, _i64, _i]),
    "irbfn_f64_forward": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _vp, _i64, _vp]),
    "irbfn_f64_vjp": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _vp, _i64, _vp]),
    "irbfn_abi_version": (_i, []),
    "irbfn_device_count": (_i, []),
    "irbfn_last_hip_error": (_i, []),
    "irbfn_strerror": (C.c_char_p, [_i]),
    "irbfn_net_last_launch": (_i, [_vp, C.c_char_p, _i, C.POINTER(_i), C.POINTER(_i)]),
}

class F64Card(C.Structure):
    """irbfn_f64_card (include/irbfn_hip.h): sizes + device pointers of the float64 gate tables."""
    _fields_ = [("D", C.c_int), ("R", C.c_int), ("K", C.c_int), ("O", C.c_int), ("basis", C.c_int), ("nsplit", C.c_int),
                ("max_ranges", C.c_int), ("n_ranges", C.c_int), ("lo_dev", C.c_void_p), ("hi_dev", C.c_void_p),
                ("delta_dev", C.c_void_p), ("dim_ranges_dev", C.c_void_p)]


_lock = threading.Lock()
_lib = None


class IrbfnError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {STATUS_NAMES.get(status, status)} ({detail})")


def load() -> C.CDLL:
    """Loads the shared library (torch first, so that both use the same HIP runtime)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m irbfn_amd.build` (hipcc, gfx950). "
                "irbfn_amd has no CPU fallback.")
        import torch  # noqa: F401  (loads libamdhip64 that the extension must share)
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the header and the library diverge
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def check(status: int, where: str) -> None:
    if status != IRBFN_OK:
        lib = load()
        detail = lib.irbfn_strerror(status).decode()
        if status == -3:
            detail += f"; hipError={lib.irbfn_last_hip_error()}"
        if status in (-1, -2, -4):
            raise ValueError(f"{where}: {STATUS_NAMES.get(status, status)} ({detail})")
        raise IrbfnError(status, where, detail)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("irbfn_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
    return torch
