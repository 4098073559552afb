"""ONE kernel at ONE batch size, N launches: the unit of a per-(kernel, batch size) rocprofv3 summary
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/run_one.py <what> <B> [launches]
what: fwd_cfg2[_k1h|_k1g] | fwd_cfg4_wide | tick_cfg4 | vjp_cfg3 | train_cfg3 | roll_<mode> | rollvjp_<mode> | spiral | spiralvjp | sparse_fwd |
      sparse_tick | sparse_vjp | sparse_train | train_1region      (mode: st_ks, st_select, fullint, frenet; T = 50, spiral N = 9)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from irbfn_amd import _lib, configs, distributed, dynamics, train  # noqa: E402
from irbfn_amd.model import WCRBFNet  # noqa: E402
from irbfn_amd.planner import plan_batch  # noqa: E402

what, B = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(0)
DP = configs.DYN_PARAMS
MODES = {"st_ks": (_lib.ROLLOUT_ST_KS, 7, 7), "st_select": (_lib.ROLLOUT_ST_SELECT, 7, 7), "fullint": (_lib.ROLLOUT_FULLINT, 5, 1),
         "frenet": (_lib.ROLLOUT_FRENET_LS, 8, 8)}


def net_of(idx):
    net = WCRBFNet.from_config(configs.model_card(idx))
    P = distributed.params_to_device(configs.synth_params(idx))
    net.bind(P)
    return net, P


def trained(run="dnmpc_128regions"):
    gdir = os.path.join(ROOT, "tests", "golden")
    z, cfg = np.load(os.path.join(gdir, f"ckpt_{run}.npz")), json.load(open(os.path.join(gdir, f"ckpt_{run}.json")))
    P = {"params": {"rbf_list": {"centers": z["centers"].astype(np.float32), "log_sigs": z["log_sigs"].astype(np.float32)},
                    "linear": {"kernel": z["kernel"].astype(np.float32), "bias": z["bias"].astype(np.float32)}}}
    ns = len(cfg["activation_idx"])
    lo = np.array([min(cfg["lower_bounds"][d]) for d in range(ns)]); hi = np.array([max(cfg["upper_bounds"][d]) for d in range(ns)])
    x = torch.from_numpy(rng.uniform(lo, hi, size=(B, ns)).astype(np.float32)).cuda()
    return WCRBFNet.from_config(cfg), P, x


if what in ("fwd_cfg2", "fwd_cfg2_k1h", "fwd_cfg2_k1g"):
    net, P = net_of(2); x = torch.from_numpy(configs.synth_queries(2, B=B)).cuda(); fn = lambda: net(x)
    if what != "fwd_cfg2":
        net.set_options(fwd_kernel=_lib.FWD_K1H if what.endswith("k1h") else _lib.FWD_K1G)
elif what == "fwd_cfg4_wide":
    net, P = net_of(4); x = torch.from_numpy(configs.synth_queries(4, B=B)).cuda(); fn = lambda: net(x)
elif what == "tick_cfg4":
    net, P = net_of(4); xq = configs.synth_queries(4, B=B); x = torch.from_numpy(xq).cuda()
    s0 = torch.from_numpy(configs.initial_state_from_query(xq)).cuda()
    fn = lambda: plan_batch(net, P, x, s0, DP, mode=_lib.ROLLOUT_ST_KS, return_controls=False)
elif what == "train_cfg3":
    net, P = net_of(3); x = torch.from_numpy(configs.synth_queries(3, B=B)).cuda(); y = torch.from_numpy(configs.synth_cotangent(3, B=B)).cuda()
    st = [train.TrainState.create(net, configs.synth_params(3), lr=1e-3, max_grad_norm=1.0)]
    def fn():
        st[0], _ = train.train_step_oneint(st[0], x, y, DP)
elif what == "vjp_cfg3":
    net, P = net_of(3); x = torch.from_numpy(configs.synth_queries(3, B=B)).cuda(); g = torch.from_numpy(configs.synth_cotangent(3, B=B)).cuda()
    fn = lambda: net.vjp(P, x, g)
elif what.startswith("roll_") or what.startswith("rollvjp_"):
    name = what.split("_", 1)[1]
    mode, S, S0 = MODES[name]
    T = 50
    x0 = rng.uniform(0, 7, (B, 1)) if name == "fullint" else (np.hstack([rng.uniform(0, 1, (B, 1)), rng.uniform(-0.2, 0.2, (B, 1)), rng.uniform(-0.3, 0.3, (B, 1)),
         rng.uniform(1, 6, (B, 1)), np.zeros((B, 2)), rng.uniform(-0.3, 0.3, (B, 1)), rng.uniform(-0.2, 0.2, (B, 1))]) if name == "frenet"
         else configs.initial_state_from_query(configs.synth_queries(4, B=B)))
    xu = torch.from_numpy(np.hstack([x0, rng.normal(0, 2.0, size=(B, 2 * T))]).astype(np.float32)).cuda()
    if what.startswith("roll_"):
        fn = lambda: dynamics.rollout_forward(mode, xu, DP, T)
    else:
        gs = torch.from_numpy(rng.normal(size=(B, T, S)).astype(np.float32)).cuda()
        fn = lambda: dynamics.rollout_vjp(mode, xu, DP, gs, T)
elif what in ("spiral", "spiralvjp"):
    N = 9
    q = torch.from_numpy(np.hstack([rng.normal(0, 0.3, (B, 4)), rng.uniform(2, 10, (B, 1))]).astype(np.float32)).cuda()
    gs = torch.from_numpy(rng.normal(size=(B, N, 6)).astype(np.float32)).cuda()
    fn = (lambda: dynamics.rollout_forward(_lib.ROLLOUT_SPIRAL, q, None, N)) if what == "spiral" else (lambda: dynamics.rollout_vjp(_lib.ROLLOUT_SPIRAL, q, None, gs, N))
elif what.startswith("sparse_") or what == "train_1region":
    net, P, x = trained("dnmpc_1regions_newdata_oldintloss_nomirror_highk") if what == "train_1region" else trained()
    Pd = distributed.params_to_device(P)
    net.bind(Pd)
    if what == "sparse_fwd":
        fn = lambda: net(x)
    elif what == "sparse_tick":
        s0 = torch.from_numpy(configs.initial_state_from_query(x.cpu().numpy())).cuda()
        fn = lambda: plan_batch(net, Pd, x, s0, DP, mode=_lib.ROLLOUT_ST_KS)
    elif what == "sparse_vjp":
        g = torch.randn(B, 10, device="cuda")
        fn = lambda: net.vjp(Pd, x, g)
    else:
        y = torch.from_numpy(np.hstack([rng.normal(size=(B, 5)) * 2, rng.normal(size=(B, 5)) * 0.5]).astype(np.float32)).cuda()
        st = [train.TrainState.create(net, P, lr=1e-3, max_grad_norm=1.0)]
        def fn():
            st[0], _ = train.train_step_fullint(st[0], x, y)
else:
    raise SystemExit(__doc__)
for _ in range(n):
    fn()
torch.cuda.synchronize()
print("ran", n, "x", what, "B =", B)
