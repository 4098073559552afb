"""K1g against K1h over the batch size at the config-2 net (the crossover behind gram_preferred / gram_geometry in rbf_forward.hip):
python tools/sweep_gram_batch.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402
card = configs.model_card(2); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(2)))
for B in (1024, 4096, 8192, 12288, 16384, 24576, 32768, 65536, 131072, 262144):
    x = torch.from_numpy(configs.synth_queries(2, B=B)).cuda()
    row = []
    for k, nm in ((_lib.FWD_K1H, "K1h"), (_lib.FWD_K1G, "K1g"), (_lib.FWD_AUTO, "auto")):
        net.set_options(fwd_kernel=k, fwd_f16_s=0, fwd_f16_qg=0)
        row.append(f"{nm} {min(t_us(lambda: net(x), 30) for _ in range(2)):.1f} ({net.last_launch()['kernel'].split('<')[0][8:]}{net.last_launch()['kernel'][-11:]})")
    for S, QG in ((4, 2), (2, 4), (1, 8)):
        net.set_options(fwd_kernel=_lib.FWD_K1G, fwd_f16_s=S, fwd_f16_qg=QG)
        row.append(f"S{S}Q{QG} {min(t_us(lambda: net(x), 30) for _ in range(2)):.1f}")
    print(B, " | ".join(row), flush=True)
