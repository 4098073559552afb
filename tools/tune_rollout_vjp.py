"""Roll-out VJP kernels (K4): time per call and effective bandwidth.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import _lib, configs, dynamics  # noqa: E402

DP = configs.DYN_PARAMS


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    rng = np.random.default_rng(0)
    for mode, name, S, S0 in ((_lib.ROLLOUT_ST_KS, "st_ks", 7, 7), (_lib.ROLLOUT_FRENET_LS, "frenet", 8, 8),
                              (_lib.ROLLOUT_FULLINT, "fullint", 5, 1)):
        for B, T in ((65536, 5), (32768, 50), (262144, 50)):
            st = rng.uniform(0.5, 2.0, size=(B, S0)).astype(np.float32)
            u = rng.normal(0, 1.0, size=(B, 2 * T)).astype(np.float32)
            xu = torch.from_numpy(np.hstack([st, u])).cuda()
            gs = torch.from_numpy(rng.normal(size=(B, T, S)).astype(np.float32)).cuda()
            us_f = timeit(lambda: dynamics.rollout_forward(mode, xu, DP, T))
            us_b = timeit(lambda: dynamics.rollout_vjp(mode, xu, DP, gs, T))
            bytes_b = 4 * B * ((S0 + 2 * T) * 2 + T * S)
            print(f"{name:8s} B={B:7d} T={T:3d}: fwd {us_f:8.1f} us   vjp {us_b:8.1f} us  ({bytes_b / us_b / 1e3:7.1f} GB/s algorithmic)")


if __name__ == "__main__":
    main()
