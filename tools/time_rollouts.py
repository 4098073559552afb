"""The roll-out family on one lease: forward (ST kinematic / ST select / inline bicycle / Frenet / spiral) and the VJPs, at
the config-4 sizes.  python tools/time_rollouts.py [out.txt] [only=substring]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irbfn_amd import _lib, configs, dynamics, planner_utils  # noqa: E402

DP = configs.DYN_PARAMS
rng = np.random.default_rng(0)
only = next((a[5:] for a in sys.argv[1:] if a.startswith("only=")), "")
outp = next((a for a in sys.argv[1:] if not a.startswith("only=")), None)
lines = []


def t_us(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(best)[1]


def report(name, B, nbytes, us):
    line = f"{name:44s} B={B:7d}: {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s algorithmic ({nbytes / us / 1e6 / 8 * 100:5.1f} % of 8 TB/s)"
    print(line, flush=True)
    lines.append(line)


MODES = {"st_ks": (_lib.ROLLOUT_ST_KS, 7, 7), "st_select": (_lib.ROLLOUT_ST_SELECT, 7, 7), "fullint": (_lib.ROLLOUT_FULLINT, 5, 1),
         "frenet": (_lib.ROLLOUT_FRENET_LS, 8, 8)}
for T in (50, 5):
    for B in (262144, 32768):
        for name, (mode, S, S0) in MODES.items():
            if only and only not in name:
                continue
            if name == "frenet":
                x0 = np.hstack([rng.uniform(0, 1, (B, 1)), rng.uniform(-0.2, 0.2, (B, 1)), rng.uniform(-0.3, 0.3, (B, 1)), rng.uniform(1, 6, (B, 1)),
                                np.zeros((B, 2)), rng.uniform(-0.3, 0.3, (B, 1)), rng.uniform(-0.2, 0.2, (B, 1))])
            elif name == "fullint":
                x0 = rng.uniform(0, 7, (B, 1))
            else:
                x0 = configs.initial_state_from_query(configs.synth_queries(4, B=B))
            u = rng.normal(0, 2.0, size=(B, 2 * T))
            xu = torch.from_numpy(np.hstack([x0, u]).astype(np.float32)).cuda()
            gs = torch.from_numpy(rng.normal(size=(B, T, S)).astype(np.float32)).cuda()
            us = t_us(lambda: dynamics.rollout_forward(mode, xu, DP, T))
            report(f"forward {name} T={T}", B, 4 * B * (S0 + 2 * T + T * S), us)
            if name != "st_select":
                us = t_us(lambda: dynamics.rollout_vjp(mode, xu, DP, gs, T))
                report(f"vjp     {name} T={T}", B, 4 * B * (2 * (S0 + 2 * T) + T * S), us)
            del xu, gs
for N in (9, 100):
    for B in (262144, 32768):
        if only and only not in "spiral":
            continue
        q = torch.from_numpy(np.hstack([rng.normal(0, 0.3, (B, 4)), rng.uniform(2, 10, (B, 1))]).astype(np.float32)).cuda()
        gs = torch.from_numpy(rng.normal(size=(B, N, 6)).astype(np.float32)).cuda()
        us = t_us(lambda: dynamics.rollout_forward(_lib.ROLLOUT_SPIRAL, q, None, N))
        report(f"forward spiral N={N}", B, 4 * B * (5 + N * 6), us)
        us = t_us(lambda: dynamics.rollout_vjp(_lib.ROLLOUT_SPIRAL, q, None, gs, N))
        report(f"vjp     spiral N={N}", B, 4 * B * (10 + N * 6), us)
        del q, gs
if outp:
    open(outp, "w").write("\n".join(lines) + "\n")
