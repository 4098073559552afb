"""DeeperWCRBFNet's RBF stage (K = 100 centres, 64-wide Dense) at batch 80000 on each forward kernel (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_deeper_fixture
from irbfn_amd import _lib
from irbfn_amd.model import DeeperWCRBFNet
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cfg, P, x, _ = load_deeper_fixture()
P32 = {"params": {k: {n: torch.from_numpy(np.asarray(v, np.float32)).cuda() for n, v in d.items()} for k, d in P["params"].items()}}
ns = len(cfg["activation_idx"])
lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
net = DeeperWCRBFNet.from_config(cfg)
for B in (80000, 4096):
    xq = torch.from_numpy(np.random.default_rng(0).uniform(lo, hi, size=(B, cfg["in_features"])).astype(np.float32)).cuda()
    for name, k in (("auto", _lib.FWD_AUTO), ("K1", _lib.FWD_K1), ("K1m", _lib.FWD_K1M), ("K1h", _lib.FWD_K1H)):
        net.stage.set_options(fwd_kernel=k)
        try:
            t = timed(lambda: net.apply(P32, xq))
            print(f"B={B} stage kernel {name}: Deeper forward {t:.1f} us [{net.stage.last_launch()['kernel']}]", flush=True)
        except Exception as e:
            print(f"B={B} {name}: {e}")
    net.stage.set_options(fwd_kernel=_lib.FWD_AUTO)
