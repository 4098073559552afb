"""Randomised differential test of the matrix-core kernels (K1h forward, wide forward, K2h VJP) against the
float32 VALU kernels (K1 / K1m / K2) over random shapes.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd.model import WCRBFNet  # noqa: E402


def setenv(**kv):
    for k, v in kv.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    worst_f, worst_v = 0.0, 0.0
    for it in range(60):
        D = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8]))
        K = int(rng.integers(1, 400))
        O = int(rng.choice([1, 2, 3, 5, 7, 10, 16, 17, 33, 64, 100, 128]))
        B = int(rng.integers(65, 5000))
        basis = str(rng.choice(["gaussian", "gaussian_wide", "inverse_quadratic", "inverse_multiquadric"]))
        lo, hi = -np.ones(D) * 2, np.ones(D) * 3
        nsplit = int(rng.integers(0, D + 1))
        cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": basis, "num_regions": 1,
               "lower_bounds": [[float(v)] for v in lo[:max(nsplit, 1)]], "upper_bounds": [[float(v)] for v in hi[:max(nsplit, 1)]],
               "dimension_ranges": [[0] * max(nsplit, 1)], "activation_idx": list(range(max(nsplit, 1))), "delta": [15.0] * max(nsplit, 1)}
        P = {"params": {"rbf_list": {"centers": rng.uniform(lo - 1, hi + 1, size=(1, K, D)).astype(np.float32),
                                     "log_sigs": rng.uniform(-0.5, 1.5, size=(1, K)).astype(np.float32)},
                        "linear": {"kernel": (rng.normal(size=(K, O)) * rng.choice([1e-3, 1.0, 100.0], size=(1, O))).astype(np.float32),
                                   "bias": rng.normal(size=(O,)).astype(np.float32)}}}
        net = WCRBFNet.from_config(cfg)
        x = torch.from_numpy(rng.uniform(lo - 0.1, hi + 0.1, size=(B, D)).astype(np.float32)).cuda()
        g = torch.from_numpy(rng.normal(size=(B, O)).astype(np.float32)).cuda()
        setenv(IRBFN_FWD_F16=1, IRBFN_FWD_F16_MINB=65, IRBFN_VJP_F16=1)
        a = net.apply(P, x); ka = net.last_launch()["kernel"]
        setenv(IRBFN_FWD_F16=0)
        b = net.apply(P, x); kb = net.last_launch()["kernel"]
        scale = float(b.abs().max()) + 1e-20
        ef = float((a - b).abs().max()) / scale
        worst_f = max(worst_f, ef)
        line = f"D={D} K={K} O={O} B={B} {basis:22s} fwd {ka.split('<')[0]:22s} vs {kb.split('<')[0]:14s} rel {ef:.1e}"
        if O <= 16 and B >= 2048:
            va = net.vjp(P, x, g)["params"]
            setenv(IRBFN_VJP_F16=0)
            vb = net.vjp(P, x, g)["params"]
            ev = 0.0
            for grp, nm in (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias")):
                sc = float(vb[grp][nm].abs().max()) + 1e-20
                ev = max(ev, float((va[grp][nm] - vb[grp][nm]).abs().max()) / sc)
            worst_v = max(worst_v, ev)
            line += f"  vjp rel {ev:.1e}"
        bad = (not np.isfinite(ef)) or ef > 2e-5
        print(("BAD " if bad else "ok  ") + line)
    print(f"worst forward {worst_f:.2e}  worst vjp {worst_v:.2e}")


if __name__ == "__main__":
    main()
