"""Relative error of the K1h / K1 forward against the float64 oracle for queries at growing distance from every
centre (bias = 0, non-negative weights: no cancellation), i.e. for ever smaller basis values: where the (hi, lo)
f16 pairs of the basis value stop carrying float32 precision.  Run with IRBFN_LIB=<variant> to compare builds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import _lib
from irbfn_amd.model import WCRBFNet
from oracle import irbfn_oracle as orc

rng = np.random.default_rng(0)
D, K, O, B = 7, 512, 10, 4096
cfg = {"in_features": D, "out_features": O, "num_kernels": K, "basis_func": "gaussian", "num_regions": 1,
       "lower_bounds": [[-1e3]] * D, "upper_bounds": [[1e3]] * D, "dimension_ranges": [[0] * D],
       "activation_idx": list(range(D)), "delta": [10.0] * D}
P = {"params": {"rbf_list": {"centers": rng.uniform(0, 1, size=(1, K, D)).astype(np.float32),
                             "log_sigs": np.zeros((1, K), np.float32)},
                "linear": {"kernel": (np.abs(rng.normal(size=(K, O))) + 0.05).astype(np.float32), "bias": np.zeros(O, np.float32)}}}
net = WCRBFNet.from_config(cfg)
for off in (0.0, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 4.5):
    x = (rng.uniform(0, 1, size=(B, D)) + off / np.sqrt(D) * np.sqrt(D)).astype(np.float32)   # shift every coordinate by off
    x[:, 1:] -= off                                                                        # ... only along axis 0
    ref = orc.wcrbfnet_apply(cfg, orc.cast_params(P, np.float64), x.astype(np.float64))
    row = [f"shift {off:3.1f}  max phi ~ {np.exp(-(max(off - 1, 0)) ** 2):8.1e}  |ref| med {np.median(ref):8.1e}"]
    for name, k in (("K1h", _lib.FWD_K1H), ("K1", _lib.FWD_K1)):
        net.set_options(fwd_kernel=k)
        got = net.apply(P, x)
        rel = np.abs(got - ref) / ref
        row.append(f"{name}: max rel {rel.max():8.1e} med {np.median(rel):8.1e}")
    print("   ".join(row))
