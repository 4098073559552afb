"""Times net.vjp (cfg-3) and prints per-call time; run under rocprofv3 --stats for the kernel split."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    B = int(sys.argv[2]) if len(sys.argv) > 2 else configs.batch_size(idx)
    card = configs.model_card(idx)
    net = WCRBFNet.from_config(card)
    P = distributed.params_to_device(configs.synth_params(idx))
    x = torch.from_numpy(configs.synth_queries(idx, B=B)).cuda()
    g = torch.from_numpy(configs.synth_cotangent(idx, B=B)).cuda()
    net.vjp(P, x, g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net.vjp(P, x, g)
    e1.record()
    torch.cuda.synchronize()
    print(f"vjp cfg{idx} B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")


if __name__ == "__main__":
    main()
