"""K1 waves per workgroup on the reference's multi-region checkpoints (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_ckpt_fixture
from irbfn_amd.model import WCRBFNet
def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for run in ("dnmpc_128regions", "dnmpc_12regions_frenet_l1_bigdata"):
    cfg, P, x, *_ = load_ckpt_fixture(run)
    P = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    net = WCRBFNet.from_config(cfg); net.bind(P)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    for B in (65536, 4096):
        rng = np.random.default_rng(1)
        xq = np.hstack([rng.uniform(lo, hi, size=(B, ns)), rng.normal(size=(B, cfg["in_features"] - ns)) * 0.1]).astype(np.float32)
        xt = torch.from_numpy(xq).cuda()
        out = []
        for nw in (0, 16, 8, 4, 2, 1):
            net.set_options(fwd_nw=nw)
            out.append(f"nw={nw}: {timed(lambda: net(xt)):.1f}")
        net.set_options(fwd_nw=0)
        print(run, f"B={B}:", "  ".join(out), net.last_launch(), flush=True)
