"""Forward time vs batch size (cfg-2 net): where does the single-launch geometry stop scaling?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
card = configs.model_card(idx)
net = WCRBFNet.from_config(card)
net.bind(distributed.params_to_device(configs.synth_params(idx)))
N = card["num_kernels"]
for B in (1, 16, 64, 128, 256, 512, 1024, 2048, 4096, 16384, 65536, 262144, 1048576):
    x = torch.from_numpy(configs.synth_queries(idx, B=B)).cuda()
    net(x)
    torch.cuda.synchronize()
    reps = 50 if B <= 65536 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        net(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ll = net.last_launch()
    print(f"B={B:8d}: {us:9.1f} us  {B / us:9.2f} Mevals/s  {B * N / us / 1e6:7.3f} Tpairs/s  grid={ll['grid']} block={ll['block']} {ll['kernel']}")
