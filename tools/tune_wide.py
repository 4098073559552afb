"""cfg-4 forward (O = 100) on one GPU share: K1h-wide geometries vs K1m.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    card = configs.model_card(4)
    net = WCRBFNet.from_config(card)
    P = distributed.params_to_device(configs.synth_params(4))
    for B in (32768, 262144):
        x = torch.from_numpy(configs.synth_queries(4, B=B)).cuda()
        net.bind(P)
        for env in ({"IRBFN_FWD_F16": "0"}, {"IRBFN_FWD_F16": "1"}, {"IRBFN_FWD_F16": "1", "IRBFN_FWD_F16_S": "1"},
                    {"IRBFN_FWD_F16": "1", "IRBFN_FWD_F16_S": "2"}, {"IRBFN_FWD_F16": "1", "IRBFN_FWD_F16_S": "4"}):
            old = {k: os.environ.get(k) for k in ("IRBFN_FWD_F16", "IRBFN_FWD_F16_S")}
            for k in old:
                os.environ.pop(k, None)
            os.environ.update(env)
            us = timeit(lambda: net(x))
            k = net.last_launch()
            pairs = B * 4096.0
            print(f"B={B} {env}: {us:.1f} us  {k['kernel']} grid {k['grid']}  {pairs * 223 / us / 1e6:.1f} fp32-equiv TFLOP/s")


if __name__ == "__main__":
    main()
