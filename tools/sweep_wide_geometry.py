import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
card = configs.model_card(4); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(4)))
for B in (32768,):
    x = torch.from_numpy(configs.synth_queries(4, B=B)).cuda()
    for S, QG in ((2,4),(2,2),(1,4),(1,2),(4,2),(1,8)):
        net.set_options(fwd_f16_s=S, fwd_f16_qg=QG)
        net(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): net(x)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} SW={S} QG={QG}: {e0.elapsed_time(e1)/20*1e3:.1f} us", net.last_launch())
