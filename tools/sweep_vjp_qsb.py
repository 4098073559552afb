"""K2g: query slices (QSB), config 3 and the reference's trained single-region net at its batch size:
python tools/sweep_vjp_qsb.py"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402


def trained(run, B):
    gdir = os.path.join(ROOT, "tests", "golden")
    z, cfg = np.load(os.path.join(gdir, f"ckpt_{run}.npz")), json.load(open(os.path.join(gdir, f"ckpt_{run}.json")))
    P = {"params": {"rbf_list": {"centers": z["centers"].astype(np.float32), "log_sigs": z["log_sigs"].astype(np.float32)},
                    "linear": {"kernel": z["kernel"].astype(np.float32), "bias": z["bias"].astype(np.float32)}}}
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = torch.from_numpy(np.random.default_rng(0).uniform(lo, hi, size=(B, ns)).astype(np.float32)).cuda()
    return WCRBFNet.from_config(cfg), P, x


cases = []
net = WCRBFNet.from_config(configs.model_card(3)); P = configs.synth_params(3)
for B in (65536, 131072):
    cases.append((f"cfg3 B={B}", net, P, torch.from_numpy(configs.synth_queries(3, B=B)).cuda()))
n1, P1, x1 = trained("dnmpc_1regions_newdata_oldintloss_nomirror_highk", 80000)
cases.append(("1-region N=1000 B=80000", n1, P1, x1))
for name, net, P, x in cases:
    Pd = distributed.params_to_device(P); net.bind(Pd)
    g = torch.randn(x.shape[0], net.out_features, device="cuda")
    row = []
    for qsb in (0, 8, 12, 16, 24, 32, 48, 64, 96):
        net.set_options(vjp_kernel=_lib.VJP_K2G, vjp_qsb=qsb)
        t = min(t_us(lambda: net.vjp(Pd, x, g), 20) for _ in range(2))
        row.append(f"{net.last_launch()['kernel'].split('QSB=')[1][:-1]}: {t:.1f}")
    net.set_options(vjp_kernel=_lib.VJP_AUTO, vjp_qsb=0)
    print(name, " | ".join(row), flush=True)
