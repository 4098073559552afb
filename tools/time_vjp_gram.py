"""K2g (u and the centre gradients on the matrix cores) against K2h at config 3: the four gradient leaves against each other and
against the float64 C restatement on a smaller batch; time per VJP.  python tools/time_vjp_gram.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402

LEAVES = (("rbf_list", "centers"), ("rbf_list", "log_sigs"), ("linear", "kernel"), ("linear", "bias"))
cfg, P = configs.model_card(3), configs.synth_params(3)
net = WCRBFNet.from_config(cfg)
Pd = distributed.params_to_device(P)
net.bind(Pd)
for B in (4096, 65536):
    x = torch.from_numpy(configs.synth_queries(3, B=B)).cuda()
    g = torch.from_numpy(configs.synth_cotangent(3, B=B)).cuda()
    out = {}
    for name, k in (("K2h", _lib.VJP_K2H), ("K2g", _lib.VJP_K2G), ("K2", _lib.VJP_K2)):
        net.set_options(vjp_kernel=k)
        gr = net.vjp(Pd, x, g)["params"]
        torch.cuda.synchronize()
        out[name] = {l: gr[l[0]][l[1]].double().cpu().numpy() for l in LEAVES}
    for l in LEAVES:
        ref = out["K2"][l]
        sc = np.abs(ref).max()
        print(f"B={B} {l[1]:9s}: |K2g - K2| / max {np.abs(out['K2g'][l] - ref).max() / sc:.2e}   |K2h - K2| / max {np.abs(out['K2h'][l] - ref).max() / sc:.2e}", flush=True)
    res = {}
    for rnd in range(3):
        for name, k in (("K2h", _lib.VJP_K2H), ("K2g", _lib.VJP_K2G)):
            net.set_options(vjp_kernel=k)
            res.setdefault(name, []).append(t_us(lambda: net.vjp(Pd, x, g), 20))
    print(f"B={B} VJP us:", {n: [round(v, 1) for v in vs] for n, vs in res.items()}, flush=True)
net.set_options(vjp_kernel=_lib.VJP_AUTO)
