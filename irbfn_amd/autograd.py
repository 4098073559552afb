"""``torch.autograd`` surface -- the counterpart of ``jax.grad`` over the reference hot path.

``wcrbf_apply(net, centers, log_sigs, kernel, bias, x)`` is differentiable w.r.t. the four parameter
leaves (never w.r.t. x: the reference never takes that gradient); the roll-outs are
differentiable w.r.t. their input rows.  Backward passes call the hand-written VJP kernels.
"""
from __future__ import annotations

import torch

from . import _lib
from .dynamics import rollout_forward, rollout_vjp, _infer_T


class _WCRBFApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, centers, log_sigs, kernel, bias, x):
        params = {"rbf_list": {"centers": centers, "log_sigs": log_sigs}, "linear": {"kernel": kernel, "bias": bias}}
        ctx.net, ctx.params = net, params
        ctx.save_for_backward(x)
        return net.apply(params, x)

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        g = ctx.net.vjp(ctx.params, x, gout.contiguous())["params"]
        return (None, g["rbf_list"]["centers"], g["rbf_list"]["log_sigs"], g["linear"]["kernel"],
                g["linear"]["bias"], None)


def wcrbf_apply(net, params: dict, x: torch.Tensor) -> torch.Tensor:
    """Differentiable ``net.apply(params, x)``; ``params`` leaves are cuda float32 tensors."""
    p = params["params"] if "params" in params else params
    return _WCRBFApply.apply(net, p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], p["linear"]["kernel"],
                             p["linear"]["bias"], x)


class _Rollout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0u, mode, dyn_params, T, clip_tie):
        ctx.mode, ctx.dyn, ctx.T, ctx.tie = mode, dyn_params, T, clip_tie
        ctx.save_for_backward(x0u)
        return rollout_forward(mode, x0u, dyn_params, T)

    @staticmethod
    def backward(ctx, gstates):
        (x0u,) = ctx.saved_tensors
        g = rollout_vjp(ctx.mode, x0u, ctx.dyn, gstates.contiguous(), ctx.T, ctx.tie)
        return g, None, None, None, None


def integrate_st_ks_mult(x_and_pred_u, dyn_params, clip_tie: float = 0.5):
    T = _infer_T(_lib.ROLLOUT_ST_KS, x_and_pred_u.shape[1])
    return _Rollout.apply(x_and_pred_u, _lib.ROLLOUT_ST_KS, dyn_params, T, clip_tie)


def dynamic_st_onestep_aux(x_u, dyn_params, clip_tie: float = 0.5):
    """Differentiable dynamics.py:103-187 (used under grad at scripts/train_nmpc.py:275-276)."""
    return _Rollout.apply(x_u, _lib.ROLLOUT_ST_KS, dyn_params, 1, clip_tie)[:, 0, :]


def integrate_frenet_mult(x_and_pred_u, dyn_params, clip_tie: float = 0.5):
    """Differentiable dynamics.py:284-290 (under grad at scripts/train_nmpc_frenet.py:408-409)."""
    T = _infer_T(_lib.ROLLOUT_FRENET_LS, x_and_pred_u.shape[1])
    return _Rollout.apply(x_and_pred_u, _lib.ROLLOUT_FRENET_LS, dyn_params, T, clip_tie)


def rollout_fullint(v0, u, clip_tie: float = 0.5):
    """Differentiable inline bicycle (scripts/train_nmpc.py:356-374)."""
    x0u = torch.cat([v0.reshape(-1, 1), u], dim=1)
    T = _infer_T(_lib.ROLLOUT_FULLINT, x0u.shape[1])
    return _Rollout.apply(x0u, _lib.ROLLOUT_FULLINT, None, T, clip_tie)


def integrate_path_mult(params, n: int = 9, clip_tie: float = 0.5):
    """Differentiable planner_utils.py:62-77 (under grad at deprecated/train_newlut.py:194-199)."""
    return _Rollout.apply(params, _lib.ROLLOUT_SPIRAL, None, int(n), clip_tie)


def train_oneint_loss(net, params, x, y, dyn_params, clip_tie: float = 0.5):
    """Loss of train_step_oneint (scripts/train_nmpc.py:258-295) on device tensors."""
    B = x.shape[0]
    init = torch.zeros((B, 7), dtype=x.dtype, device=x.device)
    init[:, 3] = x[:, 0]     # :262
    init[:, 6] = x[:, 5]     # :264
    init[:, 5] = x[:, 6]     # :266
    y_pred = wcrbf_apply(net, params, x)
    x_pred_u = torch.hstack((init, y_pred))
    x_u = torch.hstack((init, y))
    actual = dynamic_st_onestep_aux(x_u, dyn_params, clip_tie)
    pred = dynamic_st_onestep_aux(x_pred_u, dyn_params, clip_tie)
    idx = [0, 1, 3, 4]
    l2 = lambda p, t: 0.5 * (p - t) ** 2      # optax.l2_loss
    return l2(y_pred, y).mean() + l2(pred[:, idx], actual[:, idx]).mean()
