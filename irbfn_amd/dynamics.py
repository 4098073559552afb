"""Roll-outs with the reference's names and shapes (src/irbfn_mpc/dynamics.py), executed by the HIP
roll-out kernels.  ``params`` is the 13-vector ``[mu, m, I, lf, lr, C_Sf, C_Sr, h, dt, sv_max, a_max,
s_max, v_max]`` (dynamics.py:24-36); controls are ``u = [a_0..a_{T-1}, sv_0..sv_{T-1}]``
(dynamics.py:98).  The horizon T is inferred from the input width (the reference hard-codes T = 5).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import _ptr, _stream_ptr, like_input, to_device_f32


def _dyn(params):
    if params is None:
        return None, C.c_void_p(None)
    p = np.ascontiguousarray(np.asarray(params.detach().cpu() if hasattr(params, "detach") else params,
                                        dtype=np.float32).reshape(-1))
    if p.size != 13:
        raise ValueError("dynamics params must have 13 entries (dynamics.py:24-36)")
    return p, p.ctypes.data_as(C.c_void_p)


def _infer_T(mode: int, width: int) -> int:
    s0 = {_lib.ROLLOUT_ST_SELECT: 7, _lib.ROLLOUT_ST_KS: 7, _lib.ROLLOUT_FULLINT: 1, _lib.ROLLOUT_FRENET_LS: 8}[mode]
    nu = width - s0
    if nu < 0 or nu % 2:
        raise ValueError(f"input width {width} is not {s0} + 2*T")
    return nu // 2


def rollout_forward(mode: int, x0u, params, T: int):
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd = to_device_f32(x0u, torch)
    if xd.dim() != 2:
        raise ValueError("roll-out input must be 2-D [B, L]")
    B = xd.shape[0]
    L = lib.irbfn_rollout_input_dim(mode, T)
    if xd.shape[1] != L:
        raise ValueError(f"roll-out input must have {L} columns for T={T}, got {xd.shape[1]}")
    S = lib.irbfn_rollout_state_dim(mode)
    keep, pp = _dyn(params)
    out = torch.empty((B, T, S), dtype=torch.float32, device=xd.device)
    st = lib.irbfn_rollout_forward(mode, _ptr(xd), pp, _ptr(out), B, T, _stream_ptr(torch))
    _lib.check(st, "irbfn_rollout_forward")
    return like_input(out, x0u, torch)


def rollout_vjp(mode: int, x0u, params, gstates, T: int, clip_tie: float = 0.5):
    """Cotangent of all_states [B,T,S] -> cotangent of the input rows [B,L]."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd, gd = to_device_f32(x0u, torch), to_device_f32(gstates, torch)
    B = xd.shape[0]
    S = lib.irbfn_rollout_state_dim(mode)
    if tuple(gd.shape) != (B, T, S):
        raise ValueError(f"gstates must have shape ({B}, {T}, {S}), got {tuple(gd.shape)}")
    keep, pp = _dyn(params)
    g = torch.empty_like(xd)
    st = lib.irbfn_rollout_vjp(mode, _ptr(xd), pp, _ptr(gd), _ptr(g), B, T, float(clip_tie), _stream_ptr(torch))
    _lib.check(st, "irbfn_rollout_vjp")
    return like_input(g, x0u, torch)


def integrate_st_mult(x_and_pred_u, params):
    """src/irbfn_mpc/dynamics.py:94-100: [B, 7+2T] -> all_states [B, T, 7]; select(V > 3, dynamic, kinematic)."""
    return rollout_forward(_lib.ROLLOUT_ST_SELECT, x_and_pred_u, params,
                           _infer_T(_lib.ROLLOUT_ST_SELECT, x_and_pred_u.shape[1]))


def integrate_st_ks_mult(x_and_pred_u, params):
    """T-step scan of the kinematic one-step map (dynamic_st_onestep_aux applied T times)."""
    return rollout_forward(_lib.ROLLOUT_ST_KS, x_and_pred_u, params,
                           _infer_T(_lib.ROLLOUT_ST_KS, x_and_pred_u.shape[1]))


def dynamic_st_onestep_aux(x_u, params):
    """src/irbfn_mpc/dynamics.py:103-187: x_u [B, 9] -> [B, 7] (kinematic RHS, one Euler step)."""
    if x_u.shape[1] != 9:
        raise ValueError("dynamic_st_onestep_aux expects [B, 9] = 7 state + [a, sv]")
    out = rollout_forward(_lib.ROLLOUT_ST_KS, x_u, params, 1)
    return out[:, 0, :]


def integrate_frenet_mult(x_and_pred_u, params):
    """src/irbfn_mpc/dynamics.py:284-290: [B, 8+2T] -> [B, T, 8] (low-speed Frenet RHS)."""
    return rollout_forward(_lib.ROLLOUT_FRENET_LS, x_and_pred_u, params,
                           _infer_T(_lib.ROLLOUT_FRENET_LS, x_and_pred_u.shape[1]))


def rollout_fullint(v0, u):
    """Inline kinematic bicycle of train_step_fullint (scripts/train_nmpc.py:329-374):
    v0 [B], u [B, 2T] -> states [B, T, 5] = (x, y, delta, v, yaw) after every step."""
    torch = _lib.require_gpu()
    vt, ut = to_device_f32(v0, torch), to_device_f32(u, torch)
    x0u = torch.cat([vt.reshape(-1, 1), ut], dim=1)
    out = rollout_forward(_lib.ROLLOUT_FULLINT, x0u, None, _infer_T(_lib.ROLLOUT_FULLINT, x0u.shape[1]))
    return like_input(out, u, torch)
