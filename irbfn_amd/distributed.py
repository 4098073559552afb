"""Multi-GPU driver: one process per GPU (``torch.distributed``; backend "nccl" == RCCL over xGMI).

The hot path shards embarrassingly over queries / trajectories:

* parameters are replicated: rank ``src`` packs the pytree into ONE flat float32 buffer
  ((N*(D+1) + K*O + O) * 4 B: 0.30 MB at cfg-2, 1.8 MB at O = 100) and a single broadcast puts it
  on every GPU -- at parameter-upload time, not per batch;
* each rank processes a contiguous ``B/G`` slice (``shard_range``); outputs stay sharded;
* forward / roll-out steady state has NO collective;
* fwd+VJP across GPUs adds exactly one all-reduce(sum) of the flat gradient buffer (same size).

The functions are device-agnostic (CPU tensors + gloo in the tests, CUDA tensors + RCCL on the box).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

_LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))


def _shapes(net):
    R, K, D, O = net.num_regions, net.num_kernels, net.in_features, net.out_features
    return ((R, K, D), (R, K), (K, O), (O,))


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def params_to_device(params: dict, device: Optional[torch.device] = None) -> dict:
    """numpy / torch pytree -> float32 tensors on ``device`` (same nesting as the checkpoint)."""
    device = device or default_device()
    p = params["params"] if "params" in params else params
    out = {"rbf_list": {}, "linear": {}}
    for grp, name in _LEAVES:
        a = p[grp][name]
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
        out[grp][name] = t.to(device=device, dtype=torch.float32).contiguous()
    return {"params": out}


def flatten_params(params: dict) -> torch.Tensor:
    p = params["params"] if "params" in params else params
    return torch.cat([p[g][n].reshape(-1) for g, n in _LEAVES])


def unflatten_params(net, flat: torch.Tensor) -> dict:
    if flat.numel() != flat_param_count(net):
        raise ValueError("flat parameter buffer has the wrong size for this model card")
    out = {"rbf_list": {}, "linear": {}}
    off = 0
    for (g, n), shp in zip(_LEAVES, _shapes(net)):
        cnt = int(np.prod(shp))
        out[g][n] = flat[off:off + cnt].view(*shp)
        off += cnt
    return {"params": out}


def flat_param_count(net) -> int:
    return int(sum(int(np.prod(s)) for s in _shapes(net)))


def is_dist() -> bool:
    """True when a process group is alive.  A group of ONE rank still runs its collectives (cheap, and it lets a
    one-GPU box execute the RCCL branch end to end); without a group nothing collective is called."""
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def broadcast_params(net, params: Optional[dict], src: int = 0, device: Optional[torch.device] = None) -> dict:
    """Rank ``src`` passes the pytree (others may pass None); every rank gets float32 tensors on its
    device, views into one flat buffer.  ONE collective (broadcast); none if not distributed."""
    device = device or default_device()
    if not is_dist():
        if params is None:
            raise ValueError("params required on a single rank")
        return unflatten_params(net, flatten_params(params_to_device(params, device)).clone())
    import torch.distributed as dist
    if dist.get_rank() == src:
        if params is None:
            raise ValueError(f"rank {src} must provide the parameters")
        flat = flatten_params(params_to_device(params, device)).clone()
        if flat.numel() != flat_param_count(net):
            raise ValueError("parameter pytree does not match the model card")
    else:
        flat = torch.empty(flat_param_count(net), dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    return unflatten_params(net, flat)


def shard_range(B: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous slice [start, stop) of a B-row batch owned by ``rank`` (sizes differ by at most 1)."""
    if rank is None or world is None:
        rank = torch.distributed.get_rank() if is_dist() else 0
        world = torch.distributed.get_world_size() if is_dist() else 1
    base, rem = divmod(B, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def allreduce_grads(net, grads: dict) -> dict:
    """Sum of the per-rank partial parameter gradients (each rank saw its own query shard):
    one all-reduce of the flat buffer.  Identity when not distributed."""
    if not is_dist():
        return grads
    import torch.distributed as dist
    flat = flatten_params(grads).clone()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return unflatten_params(net, flat)


def gather_outputs(local: torch.Tensor, B: int) -> torch.Tensor:
    """Optional: assemble the sharded outputs [B_local, ...] into [B, ...] on every rank."""
    if not is_dist():
        return local
    import torch.distributed as dist
    world = dist.get_world_size()
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
