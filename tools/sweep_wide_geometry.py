"""cfg-4 forward (O = 100, wide K1h): pipelined vs two-buffer kernel over batch sizes and block geometries (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
card = configs.model_card(4); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(4)))
ref = {}
for B in (32768, 8192, 2048, 262144):
    x = torch.from_numpy(configs.synth_queries(4, B=B)).cuda()
    for pipe in (0, 1):
        for S, QG in ((0, 0), (2, 4), (1, 8), (4, 2)):
            if B == 262144 and (S, QG) not in ((0, 0), (1, 8)):
                continue
            net.set_options(fwd_wide_pipe=pipe, fwd_f16_s=S, fwd_f16_qg=QG)
            out = net(x); torch.cuda.synchronize()
            if B not in ref:
                ref[B] = out.clone()
            dev = float((out - ref[B]).abs().max() / ref[B].abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): net(x)
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} pipe={pipe} SW={S} QG={QG}: {e0.elapsed_time(e1)/20*1e3:.1f} us  dev vs first {dev:.1e}", net.last_launch(), flush=True)
