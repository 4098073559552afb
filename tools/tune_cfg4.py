"""cfg-4 forward (O = 100): K1 vs K1m (f32 MFMA) over geometries."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import configs, distributed
from irbfn_amd.model import WCRBFNet
from oracle import c_oracle as co
card = configs.model_card(4); net = WCRBFNet.from_config(card)
Pn = configs.synth_params(4); net.bind(distributed.params_to_device(Pn))
B = 32768
xn = configs.synth_queries(4, B=B); x = torch.from_numpy(xn).cuda()
ref = co.wcrbf_forward(card, Pn, xn[:256], np.float64)
for c in [("0", "1", "0"), ("0", "2", "0"), ("0", "4", "0"), ("0", "8", "0"), ("1", "8", "2"), ("1", "4", "2"), ("1", "8", "1"), ("1", "16", "1"), ("1", "2", "2")]:
    os.environ["IRBFN_FWD_MFMA"], os.environ["IRBFN_FWD_NW"], os.environ["IRBFN_FWD_QJ"] = c
    if c[2] == "0": os.environ.pop("IRBFN_FWD_QJ")
    try:
        out = net(x)
    except Exception as e:
        print(c, "ERR", e); continue
    torch.cuda.synchronize()
    err = float(np.abs(out[:256].cpu().numpy() - ref).max() / np.abs(ref).max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): net(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"mfma={c[0]} NW={c[1]:>2} QJ={c[2]}: {us:8.1f} us  {B*4096*223/us/1e6:6.1f} TFLOP/s  relerr {err:.2e}  {net.last_launch()}")
