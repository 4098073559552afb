"""Training step (train_step_fullint: forward, loss seeds through the 5-step bicycle, parameter VJP, clip + Adam) of the
reference's trained checkpoints at the reference's batch size (80000; scripts/configs/*.yaml), and its kernels (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_ckpt_fixture
from irbfn_amd import configs, train
from irbfn_amd.model import WCRBFNet
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 80000
for run in ("dnmpc_1regions_newdata_oldintloss_nomirror_highk", "dnmpc_128regions"):
    cfg, P, x, *_ = load_ckpt_fixture(run)
    P = {"params": {g: {n: np.asarray(v, np.float32) for n, v in d.items()} for g, d in P["params"].items()}}
    net = WCRBFNet.from_config(cfg)
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    rng = np.random.default_rng(1)
    xq = torch.from_numpy(rng.uniform(lo, hi, size=(B, 7)).astype(np.float32)).cuda()
    y = torch.from_numpy(np.hstack([rng.normal(size=(B, 5)) * 2, rng.normal(size=(B, 5)) * 0.5]).astype(np.float32)).cuda()
    state = train.TrainState.create(net, P, lr=1e-3, max_grad_norm=1.0)
    def step():
        global state
        state, loss = train.train_step_fullint(state, xq, y)
    t = timed(step)
    f = timed(lambda: net.apply(state.params, xq))
    g = torch.randn(B, 10, device="cuda")
    v = timed(lambda: net.vjp(state.params, xq, g))
    print(f"{run}: R={cfg['num_regions']} N={cfg['num_regions'] * cfg['num_kernels']} B={B}: train step {t:.1f} us  (forward {f:.1f}, parameter VJP {v:.1f})", flush=True)
