"""K1g's wide form against K1h's wide kernel at config 4 (O = 100): error against float64 on a sample, time per launch, the
one-launch planning tick.  python tools/time_gram_wide.py [B ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from irbfn_amd.planner import plan_batch  # noqa: E402
from tools.time_gram import t_us, ref_f64  # noqa: E402

cfg, P = configs.model_card(4), configs.synth_params(4)
net = WCRBFNet.from_config(cfg)
Pd = distributed.params_to_device(P)
net.bind(Pd)
for B in [int(a) for a in sys.argv[1:]] or [32768, 262144]:
    xq = configs.synth_queries(4, B=B)
    x = torch.from_numpy(xq).cuda()
    ref = ref_f64(cfg, P["params"], xq[:512])
    for name, k in (("K1h", _lib.FWD_K1H), ("K1g", _lib.FWD_K1G)):
        net.set_options(fwd_kernel=k)
        y = net(x); torch.cuda.synchronize()
        err = np.abs(y[:512].double().cpu().numpy() - ref)
        print(f"B={B} {name} {net.last_launch()['kernel']}: max abs err {err.max():.3e}, rms rel-to-scale {np.sqrt((err**2).mean()) / np.sqrt((ref**2).mean()):.3e}", flush=True)
    res = {}
    for rnd in range(3):
        for name, k in (("K1h", _lib.FWD_K1H), ("K1g", _lib.FWD_K1G)):
            net.set_options(fwd_kernel=k)
            res.setdefault(name, []).append(t_us(lambda: net(x), 20))
    print(f"B={B} forward us:", {n: [round(v, 1) for v in vs] for n, vs in res.items()}, flush=True)
    net.set_options(fwd_kernel=_lib.FWD_AUTO)
    s0 = torch.from_numpy(configs.initial_state_from_query(xq)).cuda()
    t = min(t_us(lambda: plan_batch(net, Pd, x, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS, return_controls=False), 20) for _ in range(3))
    print(f"B={B} one-launch tick (AUTO): {t:.1f} us  {net.last_launch()['kernel']}", flush=True)
