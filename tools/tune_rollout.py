"""Times the stand-alone roll-out kernels over batch sizes; prints achieved HBM GB/s (algorithmic bytes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import _lib, configs, dynamics  # noqa: E402


def t_us(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(0)
    Bs = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (8192, 32768, 65536, 131072, 262144, 1048576)
    for B in Bs:
        st = rng.normal(size=(B, 7)).astype(np.float32) * np.array([1, 1, .2, 1, .5, .2, .1], np.float32)
        st[:, 3] = rng.uniform(0.5, 7, B)
        u = rng.normal(0, 2, size=(B, 2 * T)).astype(np.float32)
        xu = torch.from_numpy(np.hstack([st, u])).cuda()
        out = torch.empty((B, T, 7), device="cuda")
        for name, mode in (("st_ks", _lib.ROLLOUT_ST_KS), ("st_select", _lib.ROLLOUT_ST_SELECT)):
            us = t_us(lambda: dynamics.rollout_forward(mode, xu, configs.DYN_PARAMS, T))
            nbytes = 4 * B * (7 + 2 * T + 7 * T)
            print(f"{name:10s} B={B:8d} T={T}: {us:9.1f} us  {B / us:8.1f} Mtraj/s  {nbytes / us / 1e3:8.1f} GB/s "
                  f"({nbytes / us / 1e3 / 8000:.1%} of 8 TB/s)")
        # copy baseline: same bytes through torch (read input, write output)
        src = torch.empty((B, 7 * T), device="cuda")
        us = t_us(lambda: out.view(B, -1).copy_(src))
        print(f"{'memcpy':10s} B={B:8d}: {us:9.1f} us  {2 * 4 * B * 7 * T / us / 1e3:8.1f} GB/s (copy of the output-sized buffer)")


if __name__ == "__main__":
    main()
