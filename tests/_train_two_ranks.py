"""Helper of tests/test_gpu_train.py::test_train_step_two_ranks_unequal_shards: launched by torch.distributed.run with two
ranks that share cuda:0 over gloo (rehearsal of the one-rank-per-GPU RCCL job on a one-GPU box).  Each rank takes its
shard_range of the batch (sizes differ by one row), runs two train_step_oneint steps and rank 0 stores parameters, loss
and gradient for the parent test to compare with the single-process step on the whole batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from irbfn_amd import configs, distributed, train  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402


def main():
    inp, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    d = np.load(inp)
    cfg = configs.model_card(3)
    net = WCRBFNet.from_config(cfg)
    params = {"params": {"rbf_list": {"centers": d["centers"], "log_sigs": d["log_sigs"]},
                         "linear": {"kernel": d["kernel"], "bias": d["bias"]}}}
    state = train.TrainState.create(net, params, lr=1e-3, max_grad_norm=1.0)
    lo, hi = distributed.shard_range(d["x"].shape[0], rank, world)
    x, y = torch.from_numpy(d["x"][lo:hi]).cuda(), torch.from_numpy(d["y"][lo:hi]).cuda()
    losses = []
    for _ in range(2):
        state, loss = train.train_step_oneint(state, x, y, np.array(configs.DYN_PARAMS))
        losses.append(float(loss))
    if rank == 0:
        np.savez(out, flat=state.flat.cpu().numpy(), g=state.g.cpu().numpy(), losses=np.array(losses), shard=np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
