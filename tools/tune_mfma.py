"""A/B of the forward kernels on one config: K1 (qlane) vs K1m (mfma) over launch geometries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from oracle import c_oracle as co  # noqa: E402


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else configs.batch_size(idx)
    card = configs.model_card(idx)
    net = WCRBFNet.from_config(card)
    Pn = configs.synth_params(idx)
    P = distributed.params_to_device(Pn)
    net.bind(P)
    xn = configs.synth_queries(idx, B=B)
    x = torch.from_numpy(xn).cuda()
    ref = co.wcrbf_forward(card, Pn, xn[:512], np.float64)
    combos = [("0", "1", "16", "4")] + [("1", "1", qj_nw[1], qj_nw[0]) for qj_nw in
                                         (("4", "16"), ("4", "8"), ("4", "4"), ("2", "16"), ("2", "8"), ("1", "16"))]
    res = {c: [] for c in combos}
    err = {}
    for rnd in range(5):
        for c in combos:
            os.environ["IRBFN_FWD_MFMA"], os.environ["IRBFN_FWD_Q"], os.environ["IRBFN_FWD_NW"], os.environ["IRBFN_FWD_QJ"] = c
            out = net(x)
            torch.cuda.synchronize()
            if rnd == 0:
                err[c] = float(np.abs(out[:512].cpu().numpy() - ref).max() / np.abs(ref).max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                net(x)
            e1.record()
            torch.cuda.synchronize()
            res[c].append(e0.elapsed_time(e1) / 20 * 1e3)
    N = card["num_kernels"] * card["num_regions"]
    fl = B * N * (3 * card["in_features"] + 2 + 2 * card["out_features"])
    for c in combos:
        v = np.array(res[c])
        os.environ["IRBFN_FWD_MFMA"], os.environ["IRBFN_FWD_Q"], os.environ["IRBFN_FWD_NW"], os.environ["IRBFN_FWD_QJ"] = c
        net(x)
        print(f"mfma={c[0]} Q={c[1]} NW={c[2]:>2} QJ={c[3]}: median {np.median(v):8.1f} us  min {v.min():8.1f}  "
              f"{fl / np.median(v) / 1e6:6.1f} TFLOP/s  relerr {err[c]:.2e}  {net.last_launch()['kernel']}")


if __name__ == "__main__":
    main()
