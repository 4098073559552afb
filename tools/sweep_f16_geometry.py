import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
card = configs.model_card(2); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(2)))
for B in (65536, 262144):
    x = torch.from_numpy(configs.synth_queries(2, B=B)).cuda()
    for S, QG in ((8,1),(4,2),(2,4),(1,8),(4,1),(2,2)):
        net.set_options(fwd_f16_s=S, fwd_f16_qg=QG)
        net(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): net(x)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} S={S} QG={QG}: {e0.elapsed_time(e1)/100*1e3:.1f} us")
