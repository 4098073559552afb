"""Sweeps the launch geometry of the fused forward kernel (IRBFN_FWD_Q / IRBFN_FWD_NW env overrides,
read per launch by the dispatcher) and prints the HIP-event time per launch.  Interleaved rounds in
one process (cdna_hip_programming.md section 5.4 rule 24)."""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else configs.batch_size(idx)
    card = configs.model_card(idx)
    net = WCRBFNet.from_config(card)
    P = distributed.params_to_device(configs.synth_params(idx))
    net.bind(P)
    x = torch.from_numpy(configs.synth_queries(idx, B=B)).cuda()
    combos = [(q, nw) for q in (1, 2) for nw in (1, 2, 4, 8, 16)]
    res = {c: [] for c in combos}
    for rnd in range(5):
        for (q, nw) in combos:
            os.environ["IRBFN_FWD_Q"], os.environ["IRBFN_FWD_NW"] = str(q), str(nw)
            net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                net(x)
            e1.record()
            torch.cuda.synchronize()
            res[(q, nw)].append(e0.elapsed_time(e1) / 20 * 1e3)
    N = card["num_kernels"] * card["num_regions"]
    fl = B * N * (3 * card["in_features"] + 2 + 2 * card["out_features"])
    for c in combos:
        v = np.array(res[c])
        print(f"Q={c[0]} NW={c[1]:2d}: median {np.median(v):8.1f} us  min {v.min():8.1f} us  "
              f"{fl / np.median(v) / 1e6:6.1f} TFLOP/s  launch={net.last_launch() if c == combos[-1] else ''}")


if __name__ == "__main__":
    main()
