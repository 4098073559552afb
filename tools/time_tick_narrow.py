"""Narrow nets (O = 10 = 2 x 5 knots, the reference's trained planners): forward alone vs the fused tick (K1 with the
roll-out in its epilogue, per-lane stores of the T x 7 states) at config 2's size (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import _lib, configs, distributed
from irbfn_amd.model import WCRBFNet
from irbfn_amd.planner import plan_batch
card = configs.model_card(2); net = WCRBFNet.from_config(card); P = distributed.params_to_device(configs.synth_params(2)); net.bind(P)
def timed(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B in (65536, 8192):
    x = configs.synth_queries(2, B=B)
    xt = torch.from_numpy(x).cuda(); st = torch.from_numpy(configs.initial_state_from_query(x)).cuda()
    for rep in range(2):
        f = timed(lambda: net(xt)); kf = net.last_launch()["kernel"]
        net.set_options(fwd_kernel=_lib.FWD_K1)
        f1 = timed(lambda: net(xt))
        net.set_options(fwd_kernel=_lib.FWD_AUTO)
        a = timed(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)); ka = net.last_launch()["kernel"]
        print(f"B={B}: forward {f:.1f} us [{kf}] | forward K1 {f1:.1f} us | fused tick {a:.1f} us [{ka}]", flush=True)
