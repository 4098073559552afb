"""K2g against K2h over the batch size (the crossover behind `use_g` in rbf_vjp.hip): python tools/sweep_vjp_batch.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402
P0 = configs.synth_params(3)
for N in (1000, 4096):
    card = dict(configs.model_card(3)); card["num_kernels"] = N
    P = {"params": {"rbf_list": {k: v[:, :N].copy() for k, v in P0["params"]["rbf_list"].items()},
                    "linear": {"kernel": P0["params"]["linear"]["kernel"][:N].copy(), "bias": P0["params"]["linear"]["bias"]}}}
    net = WCRBFNet.from_config(card); Pd = distributed.params_to_device(P); net.bind(Pd)
    for B in (2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768):
        x = torch.from_numpy(configs.synth_queries(3, B=B)).cuda(); g = torch.from_numpy(configs.synth_cotangent(3, B=B)).cuda()
        row = []
        for nm, k in (("K2h", _lib.VJP_K2H), ("K2g", _lib.VJP_K2G), ("auto", _lib.VJP_AUTO)):
            net.set_options(vjp_kernel=k)
            row.append(f"{nm} {min(t_us(lambda: net.vjp(Pd, x, g), 20) for _ in range(3)):.1f}")
        net.set_options(vjp_kernel=_lib.VJP_AUTO)
        print(f"N={N} B={B}", " | ".join(row), flush=True)
