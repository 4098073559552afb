"""Batched planning tick: the data-parallel core of ``IRBFNPlanner.plan``
(src/irbfn_mpc/irbfn_planner.py:205-212): ``pred_step`` -> ``hstack((states, pred_u))`` ->
``integrate_st_mult`` / ``dynamic_st_onestep_aux``, for B (state, goal) pairs at once, in ONE fused
launch (the predicted controls stay in LDS between the network and the roll-out), plus the batched
query construction / mirror trick around it (irbfn_planner.py:181-208, :456-492).  The
way-point search of the reference planner (numba ``nearest_point`` / ``intersect_point``,
planner_utils.py:109-240: sequential, mixed float32/float64) stays host-side and out of scope."""
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
    # wide nets outside the one-launch tick (plan_tick_wide.hip) run forward -> split-row roll-out through a controls buffer
    need_ctrl = return_controls or lib.irbfn_net_tick_needs_controls(net._handle(torch), mode, B, T) != 0
    ctrl = torch.empty((B, net.out_features), dtype=torch.float32, device=xd.device) if need_ctrl else None
    states = torch.empty((B, T, S), dtype=torch.float32, device=xd.device)
    st = lib.irbfn_net_forward_rollout(net._handle(torch), mode, _ptr(xd), _ptr(sd), pp,
                                       _ptr(ctrl) if ctrl is not None else C.c_void_p(None), _ptr(states), B, T,
                                       _stream_ptr(torch))
    _lib.check(st, "irbfn_net_forward_rollout")
    return (like_input(ctrl, x, torch) if return_controls else None), like_input(states, x, torch)


def _dev_f64(a, torch):
    if isinstance(a, torch.Tensor):
        return a.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float64).contiguous()
    import numpy as np
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def build_queries(pose, goal):
    """Cartesian planner queries (irbfn_planner.py:181-201, :240).  pose [B,7] = [x, y, delta, v, theta,
    angv, beta], goal [B,4] = [x, y, theta, v] (float64) -> device tensors (x [B,7] f32, state0 [B,7] f32,
    mirror [B] int32)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    pd, gd = _dev_f64(pose, torch), _dev_f64(goal, torch)
    B = pd.shape[0]
    if tuple(pd.shape) != (B, 7) or tuple(gd.shape) != (B, 4):
        raise ValueError("pose must be [B,7] and goal [B,4]")
    x = torch.empty((B, 7), dtype=torch.float32, device=pd.device)
    s0 = torch.empty((B, 7), dtype=torch.float32, device=pd.device)
    mirror = torch.empty((B,), dtype=torch.int32, device=pd.device)
    _lib.check(lib.irbfn_plan_queries_cartesian(_ptr(pd), _ptr(gd), _ptr(x), _ptr(s0), _ptr(mirror), B,
                                                _stream_ptr(torch)), "irbfn_plan_queries_cartesian")
    return x, s0, mirror


def build_queries_frenet(frenet, vx_goal):
    """Frenet planner queries (irbfn_planner.py:456-502).  frenet [B,8] = [s, ey, delta, vx, vy, wz, epsi,
    curv], vx_goal [B] -> (x [B,8] f32, state0 [B,8] f32, mirror [B] int32)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    fd, vd = _dev_f64(frenet, torch), _dev_f64(vx_goal, torch).reshape(-1)
    B = fd.shape[0]
    if tuple(fd.shape) != (B, 8) or vd.shape[0] != B:
        raise ValueError("frenet must be [B,8] and vx_goal [B]")
    x = torch.empty((B, 8), dtype=torch.float32, device=fd.device)
    s0 = torch.empty((B, 8), dtype=torch.float32, device=fd.device)
    mirror = torch.empty((B,), dtype=torch.int32, device=fd.device)
    _lib.check(lib.irbfn_plan_queries_frenet(_ptr(fd), _ptr(vd), _ptr(x), _ptr(s0), _ptr(mirror), B,
                                             _stream_ptr(torch)), "irbfn_plan_queries_frenet")
    return x, s0, mirror


def plan_tick(net: WCRBFNet, params: dict, x, mirror, state0=None, dyn_params=None,
              mode: int = _lib.ROLLOUT_ST_SELECT, rollout: bool = True):
    """pred_step -> un-mirror the steer-velocity controls (irbfn_planner.py:203-204) -> roll-out (:205-212).
    -> (controls [B,2T], states [B,T,S] or None).  ``mirror`` None = no flip."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    net.bind(params)
    xd = to_device_f32(x, torch)
    B = xd.shape[0]
    if net.out_features % 2:
        raise ValueError("plan_tick needs out_features = 2*T")
    T = net.out_features // 2
    md = None
    if mirror is not None:
        md = (mirror if isinstance(mirror, torch.Tensor) else torch.as_tensor(mirror)).to(device=xd.device, dtype=torch.int32).contiguous()
        if md.shape != (B,):
            raise ValueError("mirror must be [B]")
    ctrl = torch.empty((B, net.out_features), dtype=torch.float32, device=xd.device)
    states = sd = None
    pp = None
    if rollout:
        S = lib.irbfn_rollout_state_dim(mode)
        s0 = 1 if mode == _lib.ROLLOUT_FULLINT else S
        sd = to_device_f32(state0, torch)
        if sd.reshape(B, -1).shape[1] != s0:
            raise ValueError(f"state0 must be [B, {s0}]")
        keep, pp = _dyn(dyn_params)
        states = torch.empty((B, T, S), dtype=torch.float32, device=xd.device)
    null = C.c_void_p(None)
    st = lib.irbfn_plan_tick(net._handle(torch), mode, _ptr(xd), _ptr(md) if md is not None else null,
                             _ptr(sd) if sd is not None else null, pp if pp is not None else null, _ptr(ctrl),
                             _ptr(states) if states is not None else null, B, T, _stream_ptr(torch))
    _lib.check(st, "irbfn_plan_tick")
    return ctrl, states
