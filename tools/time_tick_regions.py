"""Planning tick of the reference's multi-region checkpoints (K1 with the roll-out in its epilogue) vs their forward (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_ckpt_fixture
from irbfn_amd import _lib, configs
from irbfn_amd.model import WCRBFNet
from irbfn_amd.planner import plan_batch
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cfg, P, x, *_ = load_ckpt_fixture("dnmpc_128regions")
P = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
from irbfn_amd import distributed
P = distributed.params_to_device(P)          # device-resident leaves: bind() then only compares fingerprints
net = WCRBFNet.from_config(cfg); net.bind(P)
ns = len(cfg["activation_idx"])
lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
for B in (65536, 4096):
    rng = np.random.default_rng(1)
    xq = rng.uniform(lo, hi, size=(B, 7)).astype(np.float32)
    xt = torch.from_numpy(xq).cuda(); st = torch.from_numpy(configs.initial_state_from_query(xq)).cuda()
    for rep in range(2):
        f = timed(lambda: net(xt))
        a = timed(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)); k = net.last_launch()["kernel"]
        print(f"128 regions B={B}: forward {f:.1f} us | tick {a:.1f} us [{k}]", flush=True)
