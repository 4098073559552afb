"""cfg-4 forward at the per-GPU share (B = 32768, O = 100) in a loop: the profiling target for the wide K1h kernel."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
card = configs.model_card(4); net = WCRBFNet.from_config(card); net.bind(distributed.params_to_device(configs.synth_params(4)))
x = torch.from_numpy(configs.synth_queries(4, B=B)).cuda()
net(x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): net(x)
e1.record(); torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1)/n*1e3:.1f} us", net.last_launch())
