"""K1g launch geometries over (centres N, batch B) -- the data behind gram_geometry (rbf_forward.hip): python tools/sweep_gram_geo.py
Nets: the config-2 card with its first N centres."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402

P0 = configs.synth_params(2)
for N in (256, 1000, 2048, 4096):
    card = dict(configs.model_card(2)); card["num_kernels"] = N
    P = {"params": {"rbf_list": {"centers": P0["params"]["rbf_list"]["centers"][:, :N].copy(), "log_sigs": P0["params"]["rbf_list"]["log_sigs"][:, :N].copy()},
                    "linear": {"kernel": P0["params"]["linear"]["kernel"][:N].copy(), "bias": P0["params"]["linear"]["bias"]}}}
    net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(P))
    for B in (16384, 24576, 32768, 49152, 65536, 80000, 98304, 131072, 196608, 262144):
        x = torch.from_numpy(configs.synth_queries(2, B=B)).cuda()
        row = []
        net.set_options(fwd_kernel=_lib.FWD_AUTO, fwd_f16_s=0, fwd_f16_qg=0)
        row.append(f"auto {min(t_us(lambda: net(x), 30) for _ in range(2)):.1f} ({net.last_launch()['kernel'][-11:]})")
        for S, QG in ((4, 2), (2, 4), (2, 2), (1, 8), (1, 4)):
            net.set_options(fwd_kernel=_lib.FWD_K1G, fwd_f16_s=S, fwd_f16_qg=QG)
            try:
                row.append(f"S{S}Q{QG} {min(t_us(lambda: net(x), 30) for _ in range(2)):.1f}")
            except Exception as e:
                row.append(f"S{S}Q{QG} --")
        print(f"N={N} B={B}", " | ".join(row), flush=True)
