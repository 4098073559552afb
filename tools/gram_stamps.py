"""Phase timings inside rbf_fwd_f16gram (diagnosis build):
   python tools/build_variant.py gstamps rbf_forward_gram.hip -DIRBFN_GRAM_STAMPS
   IRBFN_LIB=tools/_bin/libirbfn_gstamps.so python tools/gram_stamps.py [S QG]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
cfg, P = configs.model_card(2), configs.synth_params(2)
net = WCRBFNet.from_config(cfg); net.bind(distributed.params_to_device(P))
x = torch.from_numpy(configs.synth_queries(2)).cuda()
lib = _lib.load()
for S, QG in ((2, 4), (1, 4), (2, 8), (1, 8)):
    net.set_options(fwd_kernel=_lib.FWD_K1G, fwd_f16_s=S, fwd_f16_qg=QG)
    for _ in range(10):
        net(x)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    assert lib.irbfn_debug_gram_stamps(buf) == 0
    for b in (0, 1):
        t = np.array(buf[b * 8:b * 8 + 5], dtype=np.float64)
        n = max(t[4], 1)
        print(f"S={S} QG={QG} block {b}: steps {int(t[4])}; ticks per step: A reads + distance MFMAs issued {t[0]/n:.0f} | trans + split + PhiW {t[1]/n:.0f} | "
              f"waitcnt + barrier {t[2]/n:.0f} | DMA issue {t[3]/n:.0f} | total {t[:4].sum()/n:.0f}", flush=True)
