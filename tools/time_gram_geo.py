"""K1g launch geometries at config 2 (timing only): python tools/time_gram_geo.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from tools.time_gram import t_us  # noqa: E402

cfg, P = configs.model_card(2), configs.synth_params(2)
net = WCRBFNet.from_config(cfg); net.bind(distributed.params_to_device(P))
x = torch.from_numpy(configs.synth_queries(2)).cuda()
out = []
for S, QG in ((2, 4), (1, 4), (1, 8), (2, 2), (4, 2)):
    net.set_options(fwd_kernel=_lib.FWD_K1G, fwd_f16_s=S, fwd_f16_qg=QG)
    try:
        out.append(f"S={S} QG={QG}: {min(t_us(lambda: net(x)) for _ in range(3)):.1f}")
    except Exception as e:
        out.append(f"S={S} QG={QG}: {str(e)[:60]}")
print(os.environ.get("IRBFN_LIB", "regular"), " | ".join(out), flush=True)
