"""Training step on device: the counterparts of ``train_step_oneint`` /
``train_step_fullint`` (scripts/train_nmpc.py:258-300, :303-421) with
``optax.chain(clip_by_global_norm(max_grad_norm), adam(lr))`` (scripts/train_nmpc.py:231-233).

One step = fused forward (K1) -> loss/seed kernel -> parameter VJP (K2) -> [one all-reduce of the flat
gradient buffer when torch.distributed is initialised] -> clip + Adam kernel.  Parameters, Adam
moments and gradients live in ONE flat float32 buffer each (the pytree leaves are views), the step
count and the loss stay on the device: no host synchronisation per step (the reference does a
``jax.device_get`` per step, scripts/train_nmpc.py:477-479).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib, distributed
from .dynamics import _dyn
from .model import ClusterWCRBFNet, WCRBFNet, _ptr, _stream_ptr, to_device_f32


class TrainState:
    """Stand-in of ``flax.training.train_state.TrainState`` for the hot path: ``params`` (pytree of views
    into a flat buffer), Adam moments, device-resident step count."""

    def __init__(self, net: WCRBFNet, flat, lr, max_grad_norm, b1, b2, eps):
        import torch
        self.net = net
        self.flat = flat
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        # gradients + two trailing slots (loss * B_local, B_local) so that ONE all-reduce yields the global-batch
        # mean gradient and loss whatever the shard sizes are (distributed.shard_range hands out sizes that
        # differ by one row)
        self.gbuf = torch.zeros(flat.numel() + 2, dtype=torch.float32, device=flat.device)
        self.g = self.gbuf[:flat.numel()]
        self.step = torch.zeros(1, dtype=torch.int32, device=flat.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.partials = torch.zeros(_lib.load().irbfn_train_loss_partials(), dtype=torch.float32, device=flat.device)
        self.lr, self.max_grad_norm, self.b1, self.b2, self.eps = float(lr), float(max_grad_norm), float(b1), float(b2), float(eps)
        self.params = distributed.unflatten_params(net, self.flat)
        self.grads = distributed.unflatten_params(net, self.g)
        self.apply_fn = net.apply

    @classmethod
    def create(cls, net: WCRBFNet, params: dict, lr: float = 1e-3, max_grad_norm: float = 1.0,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, opt_state=None) -> "TrainState":
        """opt_state: (mu pytree, nu pytree, count) to resume (``checkpoint.restore_opt_state``)."""
        _lib.require_gpu()
        flat = distributed.flatten_params(distributed.params_to_device(params)).clone()
        st = cls(net, flat, lr, max_grad_norm, b1, b2, eps)
        # a training loop re-binds the parameters every step: the matrix-core kernels' "do the parameters fit" verdict is read back
        # for the first bind only (the kernels test the device-side verdict themselves; include/irbfn_hip.h, IRBFN_OPT_GRAM_STICKY)
        net.set_options(gram_sticky=1)
        if opt_state is not None:
            st.m.copy_(distributed.flatten_params(distributed.params_to_device(opt_state[0])))
            st.v.copy_(distributed.flatten_params(distributed.params_to_device(opt_state[1])))
            st.step.fill_(int(opt_state[2]))
        return st

    def opt_state(self):
        """(mu pytree, nu pytree, count): what ``checkpoint.save_checkpoint(..., opt_state=)`` stores (one host sync)."""
        return (distributed.unflatten_params(self.net, self.m), distributed.unflatten_params(self.net, self.v),
                int(self.step.item()))


class ClusterTrainState(TrainState):
    """TrainState of a ``ClusterWCRBFNet`` (scripts/train_nmpc_frenet.py:424-453): the flat buffers hold the four
    leaves of the RBF stage followed by ``cluster.kernel`` [D,R] and ``cluster.bias`` [R]."""

    def __init__(self, net: ClusterWCRBFNet, flat, lr, max_grad_norm, b1, b2, eps):
        self._cluster = net
        super().__init__(net.stage, flat[:distributed.flat_param_count(net.stage)], lr, max_grad_norm, b1, b2, eps)
        import torch
        self.net = net
        self.flat = flat
        self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
        self.gbuf = torch.zeros(flat.numel() + 2, dtype=torch.float32, device=flat.device)
        self.g = self.gbuf[:flat.numel()]
        self.params, self.grads = self._views(self.flat), self._views(self.g)
        self.apply_fn = net.apply

    def _views(self, flat) -> dict:
        n = distributed.flat_param_count(self._cluster.stage)
        D, R = self._cluster.in_features, self._cluster.num_regions
        p = distributed.unflatten_params(self._cluster.stage, flat[:n])
        p["params"]["cluster"] = {"kernel": flat[n:n + D * R].view(D, R), "bias": flat[n + D * R:n + D * R + R].view(R)}
        return p

    @classmethod
    def create(cls, net: ClusterWCRBFNet, params: dict, lr: float = 1e-3, max_grad_norm: float = 1.0,
               b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, opt_state=None) -> "ClusterTrainState":
        """opt_state: (mu pytree, nu pytree, count) with the six leaves, to resume (``checkpoint.restore_opt_state``)."""
        torch = _lib.require_gpu()

        def flat6(tree):
            p = tree["params"] if "params" in tree else tree
            stage = distributed.flatten_params(distributed.params_to_device(p))
            return torch.cat([stage, to_device_f32(p["cluster"]["kernel"], torch).reshape(-1),
                              to_device_f32(p["cluster"]["bias"], torch).reshape(-1)]).clone()
        st = cls(net, flat6(params), lr, max_grad_norm, b1, b2, eps)
        if opt_state is not None:
            st.m.copy_(flat6(opt_state[0]))
            st.v.copy_(flat6(opt_state[1]))
            st.step.fill_(int(opt_state[2]))
        return st

    def opt_state(self):
        return self._views(self.m), self._views(self.v), int(self.step.item())


def _fresh_loss(state: TrainState, torch):
    """Every step writes its loss into a 1-element device buffer of its own (caching allocator: no launch), so the
    tensor a step returns stays valid without a device-to-device copy per step."""
    state.loss = torch.empty_like(state.loss)


def _backward_and_update(state: TrainState, x, gy, torch, lib, **vjp_kw):
    """VJP -> [all-reduce] -> clip + Adam.  Returns the loss of the (global) batch, a 1-element device tensor."""
    state.net.vjp(state.params, x, gy, out=state.grads, **vjp_kw)
    loss = state.loss            # a buffer of this step's own (see _fresh_loss): no copy, never overwritten later
    if distributed.is_dist() and torch.distributed.get_world_size() > 1:
        # every rank normalised its seeds and its loss by its LOCAL batch: weight both by B_local, sum over the
        # ranks in one all-reduce of the flat buffer, divide by the global batch (exact for unequal shards)
        n, b_local = state.flat.numel(), float(x.shape[0])
        state.g.mul_(b_local)
        state.gbuf[n:n + 1].copy_(state.loss * b_local)
        state.gbuf[n + 1:].fill_(b_local)
        torch.distributed.all_reduce(state.gbuf, op=torch.distributed.ReduceOp.SUM)
        state.g.div_(state.gbuf[n + 1])
        loss = (state.gbuf[n:n + 1] / state.gbuf[n + 1]).clone()
    st = lib.irbfn_adam_clip_step(_ptr(state.flat), _ptr(state.g), _ptr(state.m), _ptr(state.v), state.flat.numel(),
                                  _ptr(state.step), state.lr, state.b1, state.b2, state.eps, state.max_grad_norm,
                                  _ptr(state.partials), _stream_ptr(torch))
    _lib.check(st, "irbfn_adam_clip_step")
    # the parameter leaves were updated in place behind torch's back: re-bind on the next apply
    getattr(state.net, "stage", state.net)._bound_fp.pop(torch.cuda.current_device(), None)
    return loss


def train_step_oneint(state: TrainState, x, y, dyn_params, clip_tie: float = 0.5) -> Tuple[TrainState, "object"]:
    """scripts/train_nmpc.py:258-300.  x [B,7], y [B,O>=2] device tensors -> (state, loss[1] on device)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd, yd = to_device_f32(x, torch), to_device_f32(y, torch)
    B, O = yd.shape
    if xd.shape[1] < 7 or O != state.net.out_features or O < 2:
        raise ValueError("train_step_oneint needs x [B,7] and y [B,out_features >= 2]")
    y_pred = state.net.apply(state.params, xd)
    gy = torch.empty_like(y_pred)
    keep, pp = _dyn(dyn_params)
    _fresh_loss(state, torch)
    st = lib.irbfn_train_seeds_oneint(_ptr(xd), _ptr(y_pred), _ptr(yd), pp, float(clip_tie), _ptr(gy), _ptr(state.loss),
                                      _ptr(state.partials), B, xd.shape[1], O, _stream_ptr(torch))
    _lib.check(st, "irbfn_train_seeds_oneint")
    return state, _backward_and_update(state, xd, gy, torch, lib)


def train_step_fullint(state: TrainState, x, y, clip_tie: float = 0.5) -> Tuple[TrainState, "object"]:
    """scripts/train_nmpc.py:303-421.  x [B,D], y [B,2T] -> (state, loss[1] on device)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd, yd = to_device_f32(x, torch), to_device_f32(y, torch)
    B, O = yd.shape
    if O != state.net.out_features or O % 2:
        raise ValueError("train_step_fullint needs y [B, out_features = 2T]")
    y_pred = state.net.apply(state.params, xd)
    gy = torch.empty_like(y_pred)
    _fresh_loss(state, torch)
    st = lib.irbfn_train_seeds_fullint(_ptr(xd), _ptr(y_pred), _ptr(yd), float(clip_tie), _ptr(gy), _ptr(state.loss),
                                       _ptr(state.partials), B, xd.shape[1], O // 2, _stream_ptr(torch))
    _lib.check(st, "irbfn_train_seeds_fullint")
    return state, _backward_and_update(state, xd, gy, torch, lib)


def train_step_frenet_fullint(state: TrainState, x, y, dyn_params, clip_tie: float = 0.5) -> Tuple[TrainState, "object"]:
    """Frenet ``train_step_fullint`` (scripts/train_nmpc_frenet.py:394-421).  x [B,8], y [B,2T] (T <= 16; the
    reference's tables hold T = 5) -> (state, loss[1] on device).  The gradient runs through
    ``integrate_frenet_mult`` on device (the adjoint of the low-speed Frenet model)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd, yd = to_device_f32(x, torch), to_device_f32(y, torch)
    B, O = yd.shape
    if xd.shape[1] != 8 or O != state.net.out_features or O % 2 or O // 2 > 16:
        raise ValueError("train_step_frenet_fullint needs x [B,8] and y [B, out_features = 2T], T <= 16")
    y_pred = state.net.apply(state.params, xd)
    gy = torch.empty_like(y_pred)
    keep, pp = _dyn(dyn_params)
    _fresh_loss(state, torch)
    st = lib.irbfn_train_seeds_frenet_fullint(_ptr(xd), _ptr(y_pred), _ptr(yd), pp, float(clip_tie), _ptr(gy),
                                              _ptr(state.loss), _ptr(state.partials), B, 8, O // 2, _stream_ptr(torch))
    _lib.check(st, "irbfn_train_seeds_frenet_fullint")
    return state, _backward_and_update(state, xd, gy, torch, lib)


def train_step_fullint_withcluster(state: ClusterTrainState, x, y, cluster_ids, dyn_params, clip_tie: float = 0.5):
    """``train_step_fullint_withcluster`` (scripts/train_nmpc_frenet.py:424-453): the Frenet full-integration loss plus
    ``optax.softmax_cross_entropy(logits, cluster_ids).mean()`` on the gate of a ClusterWCRBFNet.  x [B,8], y [B,2T],
    cluster_ids [B,R] (one-hot or soft labels) -> (state, loss[1] on device)."""
    torch = _lib.require_gpu()
    lib = _lib.load()
    xd, yd, cd = to_device_f32(x, torch), to_device_f32(y, torch), to_device_f32(cluster_ids, torch)
    B, O = yd.shape
    net = state.net
    if xd.shape[1] != 8 or O != net.out_features or O % 2 or O // 2 > 16 or tuple(cd.shape) != (B, net.num_regions):
        raise ValueError("train_step_fullint_withcluster needs x [B,8], y [B, out_features = 2T] (T <= 16), cluster_ids [B,R]")
    y_pred, logits = net.apply(state.params, xd)
    gy, glogits = torch.empty_like(y_pred), torch.empty_like(logits)
    keep, pp = _dyn(dyn_params)
    stream = _stream_ptr(torch)
    _fresh_loss(state, torch)
    st = lib.irbfn_train_seeds_frenet_fullint(_ptr(xd), _ptr(y_pred), _ptr(yd), pp, float(clip_tie), _ptr(gy),
                                              _ptr(state.loss), _ptr(state.partials), B, 8, O // 2, stream)
    _lib.check(st, "irbfn_train_seeds_frenet_fullint")
    st = lib.irbfn_softmax_xent(_ptr(logits), _ptr(cd), _ptr(glogits), _ptr(state.loss), _ptr(state.partials), 1, B,
                                net.num_regions, stream)
    _lib.check(st, "irbfn_softmax_xent")
    return state, _backward_and_update(state, xd, gy, torch, lib, glogits=glogits)


def train_epoch(state: TrainState, table, batch_size: int, only_onestep: bool = False, dyn_params=None):
    """scripts/train_nmpc.py:455-486: one pass over a ``tables.DeviceTable``.  Returns (state, losses [steps]
    on the device) -- the reference fetches every batch loss to the host (``jax.device_get``, :477-481);
    here nothing synchronises until the caller reads ``losses``."""
    torch = _lib.require_gpu()
    losses = []
    for bx, by in table.epoch(batch_size):
        if only_onestep:
            state, loss = train_step_oneint(state, bx, by, dyn_params)
        else:
            state, loss = train_step_fullint(state, bx, by)
        losses.append(loss)
    return state, (torch.cat(losses) if losses else torch.zeros(0, device=state.flat.device))
