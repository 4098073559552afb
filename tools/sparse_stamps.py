"""Phase timings inside rbf_fwd_sparse (diagnosis build with -DIRBFN_SP_STAMPS, see rbf_sparse.hip):
   python tools/build_variant.py spstamps rbf_sparse.hip -DIRBFN_SP_STAMPS
   IRBFN_LIB=tools/_bin/libirbfn_spstamps.so python tools/sparse_stamps.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402

NAMES = ["start", "tables in LDS", "factors", "scan", "prefix", "owner fill + barrier", "pair setup", "K loop", "part + barrier",
         "owner add + barrier", "all rounds", "output"]
run = sys.argv[1] if len(sys.argv) > 1 else "dnmpc_128regions"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
gdir = os.path.join(ROOT, "tests", "golden")
z, cfg = np.load(os.path.join(gdir, f"ckpt_{run}.npz")), json.load(open(os.path.join(gdir, f"ckpt_{run}.json")))
P = {"params": {"rbf_list": {"centers": z["centers"].astype(np.float32), "log_sigs": z["log_sigs"].astype(np.float32)},
                "linear": {"kernel": z["kernel"].astype(np.float32), "bias": z["bias"].astype(np.float32)}}}
net = WCRBFNet.from_config(cfg)
net.bind(distributed.params_to_device(P))
net.set_options(fwd_kernel=_lib.FWD_K1R)
ns = len(cfg["activation_idx"])
lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
rng = np.random.default_rng(1)
x = torch.from_numpy(np.hstack([rng.uniform(lo, hi, size=(B, ns)), rng.normal(size=(B, cfg["in_features"] - ns)) * 0.1]).astype(np.float32)).cuda()
for _ in range(20):
    net(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 32)()
assert lib.irbfn_debug_sparse_stamps(buf) == 0
t = np.array(buf[:12], dtype=np.int64)
print(run, "B", B, "block 0, wave 0 (s_memtime ticks = 100 MHz constant clock? or shader cycles):")
for i in range(1, 12):
    print(f"  {NAMES[i]:24s} +{t[i] - t[i - 1]:8d}   (cumulative {t[i] - t[0]})")
