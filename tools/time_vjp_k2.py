"""cfg-3 VJP on the VALU kernel K2 (forced) and on K2h (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import _lib, configs, distributed
from irbfn_amd.model import WCRBFNet
card = configs.model_card(3); net = WCRBFNet.from_config(card); P = distributed.params_to_device(configs.synth_params(3))
x = torch.from_numpy(configs.synth_queries(3)).cuda(); g = torch.from_numpy(configs.synth_cotangent(3)).cuda()
for name, k in (("K2", _lib.VJP_K2), ("K2h", _lib.VJP_K2H), ("K2", _lib.VJP_K2), ("K2h", _lib.VJP_K2H)):
    net.set_options(vjp_kernel=k)
    net.vjp(P, x, g); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): net.vjp(P, x, g)
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/20*1e3:.1f} us per call", flush=True)
