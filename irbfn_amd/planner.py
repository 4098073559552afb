"""Batched planning tick: the data-parallel core of ``IRBFNPlanner.plan``
(src/irbfn_mpc/irbfn_planner.py:205-212): ``pred_step`` -> ``hstack((states, pred_u))`` ->
``integrate_st_mult`` / ``dynamic_st_onestep_aux``, for B (state, goal) pairs at once, in ONE fused
launch (the predicted controls stay in LDS between the network and the roll-out).  The scalar host
geometry of the reference planner (way-point lookup, frame rotation, mirror trick,
irbfn_planner.py:147-201) is out of scope."""
from __future__ import annotations

import ctypes as C

from . import _lib
from .dynamics import _dyn
from .model import WCRBFNet, _ptr, _stream_ptr, like_input, to_device_f32


def plan_batch(net: WCRBFNet, params: dict, x, state0, dyn_params, mode: int = _lib.ROLLOUT_ST_SELECT,
               return_controls: bool = True):
    """x [B, D] network queries, state0 [B, S] initial vehicle states ->
    (controls [B, 2T] or None, states [B, T, S]); T = out_features // 2."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    net.bind(params)
    xd, sd = to_device_f32(x, torch), to_device_f32(state0, torch)
    B = xd.shape[0]
    if net.out_features % 2:
        raise ValueError("fused roll-out needs out_features = 2*T")
    T = net.out_features // 2
    S = lib.irbfn_rollout_state_dim(mode)
    s0 = 1 if mode == _lib.ROLLOUT_FULLINT else S
    if tuple(xd.shape) != (B, net.in_features) or sd.reshape(B, -1).shape[1] != s0:
        raise ValueError(f"x must be [B, {net.in_features}] and state0 [B, {s0}]")
    keep, pp = _dyn(dyn_params)
    # wide nets (O > 16) run the matrix-core forward + split-row roll-out and need the controls buffer
    need_ctrl = return_controls or net.out_features > 16
    ctrl = torch.empty((B, net.out_features), dtype=torch.float32, device=xd.device) if need_ctrl else None
    states = torch.empty((B, T, S), dtype=torch.float32, device=xd.device)
    st = lib.irbfn_net_forward_rollout(net._handle(torch), mode, _ptr(xd), _ptr(sd), pp,
                                       _ptr(ctrl) if ctrl is not None else C.c_void_p(None), _ptr(states), B, T,
                                       _stream_ptr(torch))
    _lib.check(st, "irbfn_net_forward_rollout")
    return (like_input(ctrl, x, torch) if return_controls else None), like_input(states, x, torch)
