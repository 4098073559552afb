"""cfg-4 planning tick (forward O = 100 -> 50-step ST-kinematic roll-out): one launch vs separate launches vs the forward
alone, per-GPU share and whole batch (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import _lib, configs, distributed
from irbfn_amd.model import WCRBFNet
from irbfn_amd.planner import plan_batch
card = configs.model_card(4); net = WCRBFNet.from_config(card); P = distributed.params_to_device(configs.synth_params(4)); net.bind(P)
def timed_sync(fn, n):
    import time
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, n in ((32768, 20), (65536, 10), (131072, 10), (262144, 5)):
    x = configs.synth_queries(4, B=B)
    xt = torch.from_numpy(x).cuda(); st = torch.from_numpy(configs.initial_state_from_query(x)).cuda()
    for rep in range(2):
        f = timed(lambda: net(xt), n)
        net.set_options(tick_fused=1)
        a = timed(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS), n); ka = net.last_launch()["kernel"]
        a2 = timed(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS, return_controls=False), n)
        net.set_options(tick_fused=0)
        b = timed(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS), n)
        net.set_options(tick_fused=1)
        net.set_options(tick_fused=1)
        sa = timed_sync(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS), n)
        net.set_options(tick_fused=0)
        sb = timed_sync(lambda: plan_batch(net, P, xt, st, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS), n)
        net.set_options(tick_fused=1)
        print(f"B={B}: host-synchronous tick (launch .. results): one launch {sa:.1f} us, separate launches {sb:.1f} us")
        print(f"B={B}: forward {f:.1f} us | one-launch tick {a:.1f} us (states only {a2:.1f}) [{ka}] | separate launches {b:.1f} us", flush=True)
