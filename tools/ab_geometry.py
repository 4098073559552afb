"""Interleaved A/B of K1h block geometries on the headline config (GPU box): rounds x geometries, 50 launches each."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
net = WCRBFNet.from_config(configs.model_card(2)); net.bind(distributed.params_to_device(configs.synth_params(2)))
xt = torch.from_numpy(configs.synth_queries(2)).cuda()
geos = [(8, 1), (4, 1), (4, 2), (2, 1), (2, 2), (2, 4)]
def timed(n=50):
    net(xt); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): net(xt)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = {g: [] for g in geos}
for r in range(6):
    for g in geos:
        net.set_options(fwd_f16_s=g[0], fwd_f16_qg=g[1])
        res[g].append(timed())
for g in geos:
    v = sorted(res[g])
    print(f"S{g[0]}xQG{g[1]}: median {v[len(v)//2]:.1f} us  min {v[0]:.1f}  all {[round(t,1) for t in res[g]]}", flush=True)
