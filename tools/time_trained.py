"""Forward (and tick) time of the reference's trained checkpoints (tests/golden fixtures) at planner-batch sizes: which
kernel runs and how long it takes, for the multi-region nets in particular (GPU box)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_ckpt_fixture, CKPT_RUNS
from irbfn_amd.model import WCRBFNet
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for run in CKPT_RUNS:
    cfg, P, x, *_ = load_ckpt_fixture(run)
    P = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    net = WCRBFNet.from_config(cfg)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    print(f"== {run}: R={cfg['num_regions']} K={cfg['num_kernels']} D={cfg['in_features']} O={cfg['out_features']} basis={cfg['basis_func']}")
    for B in (1024, 65536):
        rng = np.random.default_rng(1)
        xq = rng.uniform(lo, hi, size=(B, cfg["in_features"])).astype(np.float32) if ns == cfg["in_features"] else \
            np.hstack([rng.uniform(lo, hi, size=(B, ns)), rng.normal(size=(B, cfg["in_features"] - ns)) * 0.1]).astype(np.float32)
        xt = torch.from_numpy(xq).cuda()
        net.bind(P)
        t = timed(lambda: net(xt))
        N = cfg["num_regions"] * cfg["num_kernels"]
        print(f"   B={B}: {t:8.1f} us  [{net.last_launch()['kernel']}]  {B * N / t / 1e6:.2f} Tpairs/s nominal (N = {N})", flush=True)
