"""``planner_utils`` of the reference (src/irbfn_mpc/planner_utils.py) on the GPU: the cubic-spiral path integrator
(:62-77) and the way-point geometry of the pure-pursuit front end (:109-233), batched."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .dynamics import rollout_forward
from .model import _ptr, _stream_ptr

N = 9   # planner_utils.py:8


def integrate_path_mult(params, n: int = N):
    """params [B, 5] = (k0, k1, k2, k3, s) -> all_states [B, n, 6] = [x, y, theta, kappa, dx, dy]."""
    if params.shape[1] != 5:
        raise ValueError("integrate_path_mult expects [B, 5] = (k0, k1, k2, k3, s)")
    return rollout_forward(_lib.ROLLOUT_SPIRAL, params, None, int(n))


def _f64(a, torch):
    if isinstance(a, torch.Tensor):
        return a.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def nearest_point(points, trajectory):
    """Batched ``nearest_point`` (planner_utils.py:109-146).  points [B,2] (or [2]), trajectory [N,2] ->
    (projections [B,2], dists [B], t [B], segment index [B]) as device tensors (float64 / int32)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    pd, td = _f64(points, torch).reshape(-1, 2), _f64(trajectory, torch)
    B, N = pd.shape[0], td.shape[0]
    proj = torch.empty((B, 2), dtype=torch.float64, device=pd.device)
    dist = torch.empty((B,), dtype=torch.float64, device=pd.device)
    t = torch.empty((B,), dtype=torch.float64, device=pd.device)
    seg = torch.empty((B,), dtype=torch.int32, device=pd.device)
    _lib.check(lib.irbfn_nearest_point(_ptr(pd), _ptr(td), _ptr(proj), _ptr(dist), _ptr(t), _ptr(seg), B, N,
                                       _stream_ptr(torch)), "irbfn_nearest_point")
    return proj, dist, t, seg


def intersect_point(points, radius, trajectory, t=None, wrap=False):
    """Batched ``intersect_point`` (planner_utils.py:149-233).  points [B,2], trajectory [N,2], t [B] = i + t of
    the search start (None = 0) -> (first_p [B,2] float32, first_i [B] int32, first_t [B] float32, found [B] int32);
    rows with found == 0 are the reference's ``(None, None, None)`` (NaN / undefined index)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    pd, td = _f64(points, torch).reshape(-1, 2), _f64(trajectory, torch)
    B, N = pd.shape[0], td.shape[0]
    ts = None if t is None else _f64(t, torch).reshape(-1)
    if ts is not None and ts.shape[0] != B:
        raise ValueError("t must have one entry per point")
    fp = torch.empty((B, 2), dtype=torch.float32, device=pd.device)
    fi = torch.empty((B,), dtype=torch.int32, device=pd.device)
    ft = torch.empty((B,), dtype=torch.float32, device=pd.device)
    found = torch.empty((B,), dtype=torch.int32, device=pd.device)
    _lib.check(lib.irbfn_intersect_point(_ptr(pd), _ptr(td), _ptr(ts) if ts is not None else C.c_void_p(None), float(radius),
                                         int(bool(wrap)), _ptr(fp), _ptr(fi), _ptr(ft), _ptr(found), B, N,
                                         _stream_ptr(torch)), "irbfn_intersect_point")
    return fp, fi, ft, found
