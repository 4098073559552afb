import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed, _lib
from irbfn_amd.model import WCRBFNet
from irbfn_amd.planner import plan_batch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); from conftest import load_ckpt_fixture
cfg, params, x0, *_ = load_ckpt_fixture("dnmpc_1regions_newdata_oldintloss_nomirror_highk")
net = WCRBFNet.from_config(cfg)
P = distributed.params_to_device(params)
for B in (1, 8, 64, 256):
    x = torch.from_numpy(np.repeat(x0[:1].astype(np.float32), B, 0)).cuda()
    s0 = torch.from_numpy(configs.initial_state_from_query(x.cpu().numpy())).cuda()
    def tick(): return plan_batch(net, P, x, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_SELECT)
    def fwd(): return net.apply(P, x)
    from irbfn_amd import dynamics
    def two():
        u = net.apply(P, x)
        return dynamics.integrate_st_mult(torch.cat([s0, u], dim=1), configs.DYN_PARAMS)
    for name, fn in (("forward only", fwd), ("fused tick", tick), ("forward + cat + stand-alone roll-out", two)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1)/200*1e3:.1f} us", net.last_launch()["kernel"])
