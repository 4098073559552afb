"""One live centre (all other weights zero), queries at distances giving phi = 2^-k: K1h output against exact."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import _lib
from irbfn_amd.model import WCRBFNet
D, K, O = 7, 64, 10
cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "gaussian", "num_regions": 1,
       "lower_bounds": [[-1e3]] * D, "upper_bounds": [[1e3]] * D, "dimension_ranges": [[0] * D],
       "activation_idx": list(range(D)), "delta": [10.0] * D}
for live_w in (1.0, 0.7):
    W = np.zeros((K, O), np.float32); W[5, :] = live_w
    W[6, 3] = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0          # optional second weight in column 3
    c = np.zeros((1, K, D), np.float32); c[0, 6] = 100.0
    P = {"params": {"rbf_list": {"centers": c, "log_sigs": np.zeros((1, K), np.float32)},
                    "linear": {"kernel": W, "bias": np.zeros(O, np.float32)}}}
    ks = np.arange(0, 44, 2)
    B = 128
    x = np.zeros((B, D), np.float32)
    d = np.sqrt(ks * np.log(2.0))
    x[:len(ks), 0] = d
    exact = np.exp(-(x[:, 0].astype(np.float64)) ** 2) * live_w
    net = WCRBFNet.from_config(cfg)
    for name, kk in (("K1h", _lib.FWD_K1H), ("K1", _lib.FWD_K1)):
        net.set_options(fwd_kernel=kk)
        got = net.apply(P, x)[:len(ks), 0]
        print(name, "w", live_w, " ".join(f"{k}:{abs(g - e) / e:.1e}" for k, g, e in zip(ks, got, exact[:len(ks)])))
