"""Region-sparse (K1r / K2r) vs dense gated (K1 / K2) kernels on the reference's trained multi-region planners, interleaved
on one lease: forward at B = 65536, planning tick, train_step_fullint at B = 80000.  python tools/time_sparse.py [out.txt]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed, train  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from irbfn_amd.planner import plan_batch  # noqa: E402


def t_us(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    out = []
    gdir = os.path.join(ROOT, "tests", "golden")
    for run in ("dnmpc_128regions", "dnmpc_12regions_frenet_l1_bigdata"):
        z, cfg = np.load(os.path.join(gdir, f"ckpt_{run}.npz")), json.load(open(os.path.join(gdir, f"ckpt_{run}.json")))
        P = {"params": {"rbf_list": {"centers": z["centers"].astype(np.float32), "log_sigs": z["log_sigs"].astype(np.float32)},
                        "linear": {"kernel": z["kernel"].astype(np.float32), "bias": z["bias"].astype(np.float32)}}}
        net = WCRBFNet.from_config(cfg)
        Pd = distributed.params_to_device(P)
        net.bind(Pd)
        ns = len(cfg["activation_idx"])
        lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
        rng = np.random.default_rng(1)
        for B in (65536, 8192):
            xq = np.hstack([rng.uniform(lo, hi, size=(B, ns)), rng.normal(size=(B, cfg["in_features"] - ns)) * 0.1]).astype(np.float32)
            x = torch.from_numpy(xq).cuda()
            g = net.gate(x)
            nz = (g != 0).sum(dim=1).float()
            line = f"{run} B={B}: regions with gamma != 0 per query: mean {nz.mean().item():.2f} max {int(nz.max().item())} of {cfg['num_regions']}"
            res = {}
            for rnd in range(3):
                for name, k in (("dense_K1", _lib.FWD_K1), ("sparse_K1r", _lib.FWD_K1R)):
                    net.set_options(fwd_kernel=k)
                    res.setdefault(name, []).append(t_us(lambda: net(x)))
            line += " | forward us: " + ", ".join(f"{n} {sorted(v)[1]:.1f}" for n, v in res.items())
            if cfg["out_features"] == 10 and cfg["in_features"] == 7:
                s0 = torch.from_numpy(configs.initial_state_from_query(xq)).cuda()
                res = {}
                for rnd in range(3):
                    for name, k in (("dense_K1", _lib.FWD_K1), ("sparse_K1r", _lib.FWD_K1R)):
                        net.set_options(fwd_kernel=k)
                        res.setdefault(name, []).append(t_us(lambda: plan_batch(net, Pd, x, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)))
                line += " | tick(T=5) us: " + ", ".join(f"{n} {sorted(v)[1]:.1f}" for n, v in res.items())
            net.set_options(fwd_kernel=_lib.FWD_AUTO)
            out.append(line)
            print(line, flush=True)
        if cfg["out_features"] == 10 and cfg["in_features"] == 7 and hasattr(_lib, "VJP_K2R"):
            Bt = 80000
            xb = torch.from_numpy(rng.uniform(lo, hi, size=(Bt, 7)).astype(np.float32)).cuda()
            yb = torch.from_numpy(np.hstack([rng.normal(size=(Bt, 5)) * 2, rng.normal(size=(Bt, 5)) * 0.5]).astype(np.float32)).cuda()
            gy = torch.randn(Bt, 10, device="cuda")
            res, resv = {}, {}
            for rnd in range(3):
                for name, kf, kv in (("dense", _lib.FWD_K1, _lib.VJP_K2), ("sparse", _lib.FWD_K1R, _lib.VJP_K2R)):
                    try:
                        net.set_options(fwd_kernel=kf, vjp_kernel=kv)
                        st = [train.TrainState.create(net, P, lr=1e-3, max_grad_norm=1.0)]
                        def step():
                            st[0], _ = train.train_step_fullint(st[0], xb, yb)
                        res.setdefault(name, []).append(t_us(step, 20))
                        resv.setdefault(name, []).append(t_us(lambda: net.vjp(st[0].params, xb, gy), 20))
                    except Exception as e:      # K2r not built yet
                        res.setdefault(name, []).append(float("nan")); resv.setdefault(name, []).append(float("nan"))
                        print("  ", name, repr(e)[:200])
            line = f"{run} train_step_fullint B={Bt} us: " + ", ".join(f"{n} {sorted(v)[1]:.1f}" for n, v in res.items())
            line += " | parameter VJP alone us: " + ", ".join(f"{n} {sorted(v)[1]:.1f}" for n, v in resv.items())
            net.set_options(fwd_kernel=_lib.FWD_AUTO, vjp_kernel=_lib.VJP_AUTO)
            out.append(line)
            print(line, flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
