"""K1g (distances on the matrix cores as a Gram expansion) against K1h on one lease: error against the float64 oracle on a
sample, and time per launch, interleaved.  python tools/time_gram.py [cfg ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402


def t_us(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def ref_f64(cfg, p, x):
    """float64 evaluation of the single-region net (src/irbfn_mpc/model.py:169-198) for the error columns"""
    x = x.astype(np.float64)
    c, ls = p["rbf_list"]["centers"][0].astype(np.float64), p["rbf_list"]["log_sigs"][0].astype(np.float64)
    d2 = ((x[:, None, :] - c[None]) ** 2).sum(-1) / np.exp(ls)[None] ** 2
    phi = {"gaussian": lambda t: np.exp(-t), "inverse_quadratic": lambda t: 1 / (1 + t), "inverse_multiquadric": lambda t: 1 / np.sqrt(1 + t)}[cfg["basis_func"]](d2)
    lo, hi = np.array([b[0] for b in cfg["lower_bounds"]]), np.array([b[0] for b in cfg["upper_bounds"]])
    dl = np.array(cfg["delta"])
    gam = np.prod((np.tanh(dl * (x - lo)) + 1) / 2 * (np.tanh(dl * (hi - x)) + 1) / 2, axis=1)
    return gam[:, None] * (phi @ p["linear"]["kernel"].astype(np.float64)) + p["linear"]["bias"].astype(np.float64)


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [2]
    for ci in cfgs:
        cfg, P = configs.model_card(ci), configs.synth_params(ci)
        net = WCRBFNet.from_config(cfg)
        net.bind(distributed.params_to_device(P))
        B = min(configs.batch_size(ci), 262144)
        xq = configs.synth_queries(ci, B=B)
        x = torch.from_numpy(xq).cuda()
        ns = 1024
        ref = ref_f64(cfg, P["params"], xq[:ns])
        for name, k in (("K1h", _lib.FWD_K1H), ("K1g", _lib.FWD_K1G)):
            net.set_options(fwd_kernel=k)
            try:
                y = net(x)
            except Exception as e:
                print(name, "failed:", repr(e)[:300]); continue
            torch.cuda.synchronize()
            got = y[:ns].double().cpu().numpy()
            err = np.abs(got - ref)
            print(f"cfg {ci} {name} {net.last_launch()['kernel']}: max abs err {err.max():.3e}  max rel {np.max(err / np.maximum(np.abs(ref), 1e-30)):.3e}"
                  f"  rms rel-to-scale {np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()):.3e}", flush=True)
        res = {}
        for rnd in range(3):
            for name, k in (("K1h", _lib.FWD_K1H), ("K1g", _lib.FWD_K1G)):
                net.set_options(fwd_kernel=k)
                try:
                    res.setdefault(name, []).append(t_us(lambda: net(x)))
                except Exception as e:
                    res.setdefault(name, []).append(float("nan"))
        print(f"cfg {ci} B={B} us per launch:", {n: [round(v, 1) for v in vs] for n, vs in res.items()}, flush=True)
        for S, QG in ((1, 8), (2, 4), (4, 2), (2, 2), (1, 4)):
            net.set_options(fwd_kernel=_lib.FWD_K1G, fwd_f16_s=S, fwd_f16_qg=QG)
            try:
                print(f"   K1g S={S} QG={QG}: {t_us(lambda: net(x)):.1f} us", flush=True)
            except Exception as e:
                print(f"   K1g S={S} QG={QG}: {repr(e)[:100]}")
        net.set_options(fwd_kernel=_lib.FWD_AUTO, fwd_f16_s=0, fwd_f16_qg=0)


if __name__ == "__main__":
    main()
