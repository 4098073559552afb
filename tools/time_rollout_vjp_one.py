"""K4 at the config-4 size only (ST-kinematic, T = 50, B = 262144): time per call (GPU box; for A/B builds)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import _lib, configs, dynamics
rng = np.random.default_rng(0)
B, T, S = 262144, 50, 7
xu = torch.from_numpy(np.hstack([rng.uniform(0.5, 2.0, size=(B, 7)), rng.normal(0, 1.0, size=(B, 2 * T))]).astype(np.float32)).cuda()
gs = torch.from_numpy(rng.normal(size=(B, T, S)).astype(np.float32)).cuda()
f = lambda: dynamics.rollout_vjp(_lib.ROLLOUT_ST_KS, xu, configs.DYN_PARAMS, gs, T)
f(); torch.cuda.synchronize()
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"K4 st_ks B={B} T={T}: {e0.elapsed_time(e1)/10*1e3:.1f} us", flush=True)
