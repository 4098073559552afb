"""The side kernels at plausible sizes (looking for cliffs): spiral path roll-out and its VJP, way-point geometry.  GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import _lib, configs, dynamics, planner_utils
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rng = np.random.default_rng(0)
for B in (4096, 262144):
    p = np.hstack([rng.uniform(2, 10, size=(B, 1)), rng.normal(size=(B, 4)) * 0.2]).astype(np.float32)   # s, k0..k3
    pt = torch.from_numpy(p).cuda()
    for N in (100,):
        t = timed(lambda: planner_utils.integrate_path_mult(pt, N))
        print(f"spiral integrate_path_mult B={B} N={N}: {t:.1f} us  ({4 * B * (5 + N * 6) / t / 1e3:.0f} GB/s)", flush=True)
    g = torch.randn(B, 100, 6, device="cuda")
    t = timed(lambda: dynamics.rollout_vjp(_lib.ROLLOUT_SPIRAL, pt, None, g, 100))
    print(f"spiral VJP B={B} N=100: {t:.1f} us", flush=True)
# way-point geometry: batch of points against a 1000-point track
traj = np.stack([np.cos(np.linspace(0, 2 * np.pi, 1000)) * 20, np.sin(np.linspace(0, 2 * np.pi, 1000)) * 10], 1)
trj = torch.from_numpy(traj).cuda()
for B in (1, 4096, 65536):
    pts = torch.from_numpy(rng.normal(size=(B, 2)) * 8).cuda()
    t = timed(lambda: planner_utils.nearest_point(pts, trj))
    t2 = timed(lambda: planner_utils.intersect_point(pts, 2.0, trj))
    print(f"nearest_point B={B} (1000-point track): {t:.1f} us; intersect_point {t2:.1f} us", flush=True)
