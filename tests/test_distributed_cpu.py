"""world_size-2 gloo tests (CPU) of the multi-GPU driver logic: one broadcast of the flat parameter
buffer, contiguous query shards, one all-reduce of the gradient buffer, output gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = WCRBFNet.from_config(configs.model_card(1))
        params = configs.synth_params(1) if rank == 0 else None
        got = distributed.broadcast_params(net, params, src=0, device=torch.device("cpu"))
        ref = configs.synth_params(1)["params"]
        ok = all(np.array_equal(got["params"][g][n].numpy(), ref[g][n])
                 for g, n in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")))
        # shards tile the batch exactly
        B = 1001
        lo, hi = distributed.shard_range(B)
        x = torch.arange(B, dtype=torch.float32).reshape(B, 1)
        full = distributed.gather_outputs(x[lo:hi] * 2.0, B)
        ok = ok and torch.equal(full, x * 2.0)
        # gradient all-reduce = sum of per-rank partials
        grads = {"params": {g: {n: torch.full_like(v, float(rank + 1)) for n, v in d.items()}
                            for g, d in got["params"].items()}}
        red = distributed.allreduce_grads(net, grads)
        ok = ok and all(float(t.min()) == 3.0 == float(t.max()) for d in red["params"].values() for t in d.values())
        q.put((rank, ok, lo, hi))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_broadcast_shard_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 501, 501, 1001)


def test_shard_range_covers_batch():
    for B in (0, 1, 7, 64, 65536, 262144):
        for world in (1, 2, 4, 8):
            spans = [distributed.shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flatten_roundtrip_single_process():
    net = WCRBFNet.from_config(configs.model_card(2))
    p = distributed.params_to_device(configs.synth_params(2), torch.device("cpu"))
    flat = distributed.flatten_params(p)
    assert flat.numel() == distributed.flat_param_count(net) == 4096 * 7 + 4096 + 4096 * 10 + 10
    back = distributed.unflatten_params(net, flat)
    assert torch.equal(back["params"]["linear"]["kernel"], p["params"]["linear"]["kernel"])
    single = distributed.broadcast_params(net, configs.synth_params(2), device=torch.device("cpu"))
    assert torch.equal(single["params"]["rbf_list"]["centers"], p["params"]["rbf_list"]["centers"])
    with pytest.raises(ValueError):
        distributed.unflatten_params(net, flat[:-1])
