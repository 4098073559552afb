"""cfg-2 forward (narrow K1h) at a chosen batch in a loop: profiling target (effective-clock measurements need long dispatches)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
card = configs.model_card(2); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(2)))
x = torch.from_numpy(configs.synth_queries(2, B=B)).cuda()
net(x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): net(x)
e1.record(); torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1)/n*1e3:.1f} us", net.last_launch())
