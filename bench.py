#!/usr/bin/env python3
"""Headline benchmark of the IRBFN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): RBF-net evals/s.  One "step" = one fused forward pass of the WCRBFNet
(region gate + 4096 Gaussian centres + Dense) over one batch of 65 536 synthetic 7-D queries
(BASELINE config 2), inputs and parameters resident in HBM.  Weak scaling: every rank processes its
own 65 536-query shard; rank 0's parameters are broadcast once over RCCL before the timed region and
there is no collective in the steady state.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel, HIP-event timing on the launch stream) and ``cpu_baseline`` (the C
restatement of the reference path, oracle/, timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector peak == dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, default=2, help="BASELINE config index (2 = headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=65536)
    return ap.parse_args()


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal knobs (one-GPU box): IRBFN_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
        # IRBFN_DIST_BACKEND=gloo avoids RCCL's one-rank-per-device rule; the driver uses neither.
        dev = 0 if os.environ.get("IRBFN_BENCH_SAME_DEVICE") == "1" else local
        backend = os.environ.get("IRBFN_DIST_BACKEND", "nccl")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    return rank, world, local


def timed_region(fn, steps, warmup, world):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks.
    Also returns the HIP-event time of the same K launches on the launch stream."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, ev_ms


def main():
    args = parse()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (irbfn_amd has no CPU fallback)")
    from irbfn_amd import configs, distributed
    from irbfn_amd.model import WCRBFNet
    rank, world, local = dist_setup(args.gpus)
    idx = args.config
    card = configs.model_card(idx)
    net = WCRBFNet.from_config(card)
    B = configs.batch_size(idx)
    D, K, O = card["in_features"], card["num_kernels"], card["out_features"]
    N = K * card["num_regions"]

    # rank 0 owns the parameters; one RCCL broadcast puts them on every GPU (no further collectives)
    params = configs.synth_params(idx) if rank == 0 else None
    params = distributed.broadcast_params(net, params, src=0)
    net.bind(params)
    # each rank's shard of the (weak-scaled) global batch: different seed per rank
    x = torch.from_numpy(configs.synth_queries(idx, seed=1123 + rank)).cuda()
    torch.cuda.synchronize()

    def step():
        return net(x)

    wall, ev_ms = timed_region(step, args.steps, args.warmup, world)
    ms_per_step = wall / args.steps * 1e3
    value = B * world * args.steps / wall

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if world > 1:
        # N > 1: the other ranks are gone after the timed region -- nothing below may issue a collective (the
        # training-step extra all-reduces its gradients when a process group is up); CPU baseline and extras are
        # reported at N = 1 only
        args.no_cpu_baseline = True
        args.no_extras = True

    # ---- roofline of the dominant kernel (the fused forward): algorithmic work / event time
    kern_s = ev_ms / 1e3 / args.steps
    flops = B * N * (3 * D + 2 + 2 * O)                          # per pair 3D + 2 + 2O
    abytes = 4 * (B * D + N * D + N + N * O + O + B * O)         # every tensor touched once
    launch = net.last_launch()
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):                      # measured by rocprofv3 PMC passes of this same command
        rec = json.load(open(tpath)).get(launch["kernel"])
        if rec:
            traffic, traffic_src = rec["bytes"], "profiles/r01_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE, per launch)"
    roofline = {
        "bound": "mfma", "achieved": flops / kern_s / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
        "frac": flops / kern_s / 1e12 / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
        "kernel": launch["kernel"], "grid": launch["grid"], "block": launch["block"],
        "avg_launch_us": kern_s * 1e6, "algorithmic_flops": flops, "algorithmic_bytes": abytes,
        "note": "priced against the fp32 peak 157.3 TFLOP/s (fp32 vector == dense f32-input MFMA peak): distances, basis and the operand splits run on the f32 VALU, the centre x weight reduction on the f16 matrix cores with hi/lo operand pairs (float32-equivalent result, same error as the all-f32 kernel); VALU/transcendental-bound (2400 flop/B), not HBM-bound; transcendental count = B*N",
        "hbm": {"achieved": abytes / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": abytes / kern_s / 1e9 / PEAK_HBM_GBS},
    }

    # ---- CPU baseline: C restatement of the reference path (oracle/), bounded sample, same box
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import c_oracle as co          # checker / baseline leg only
        ns = min(args.cpu_sample, B)
        co.set_num_threads(available_cores())
        xs = x[:ns].cpu().numpy()
        co.wcrbf_forward(card, params_np(params), xs[:4096], np.float32)      # warm-up (thread pool)
        dts = []
        for _ in range(5):                         # ~10-30 core-seconds of CPU work in total
            t0 = time.perf_counter()
            ref = co.wcrbf_forward(card, params_np(params), xs, np.float32)
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[2]
        got = net(x[:ns]).cpu().numpy()
        ref64 = co.wcrbf_forward(card, params_np(params), xs[:1024], np.float64)
        cpu = {"value": ns / dt, "unit": "evals/s", "cores": co.num_threads(), "kind": "port",
               "sample": f"{ns} of the {B} queries of the same workload x 5 repeats (median), float32, OpenMP C "
                         f"restatement of the reference path (oracle/irbfn_oracle.c), {dt:.3f} s wall per repeat "
                         f"(~{5 * dt * co.num_threads():.0f} core-seconds of CPU work in total)",
               "parity_rel_err_vs_f64": float(np.abs(got[:1024] - ref64).max() / np.abs(ref64).max()),
               "parity_rel_err_vs_cpu_f32": float(np.abs(got - ref).max() / np.abs(ref).max())}

    extras = None
    if not args.no_extras and idx == 2:
        extras = run_extras(net, params, x, configs, torch)

    line = {
        "metric": "RBF-net evals/s (B queries x N centres), forward", "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config {idx}: {N} centres, d={D}, O={O}, batch={B} per GPU, "
                               f"{card['basis_func']} RBF forward (gate + RBF + Dense fused)",
                   "centres": N, "in_features": D, "out_features": O, "batch_per_gpu": B,
                   "global_batch": B * world, "basis": card["basis_func"], "parallelism": f"dp{world} (query shards)"},
        "pair_evals_per_s": value * N,
        "roofline": roofline, "cpu_baseline": cpu,
    }
    if extras:
        line["extras"] = extras
    print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def available_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box
    exposes 128 hardware threads but gives a one-GPU job a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def params_np(params):
    p = params["params"]
    return {"params": {"rbf_list": {k: v.cpu().numpy() for k, v in p["rbf_list"].items()},
                       "linear": {k: v.cpu().numpy() for k, v in p["linear"].items()}}}


def _time(fn, reps, torch):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 1e3 / reps


def run_extras(net, params, x, configs, torch):
    """Secondary numbers (outside the timed region): fwd+VJP (cfg-3), stand-alone roll-out and fused
    planning tick (per-GPU share of cfg-4)."""
    from irbfn_amd import _lib, dynamics
    from irbfn_amd.model import WCRBFNet
    from irbfn_amd.planner import plan_batch
    out = {}
    B = x.shape[0]
    g = torch.from_numpy(configs.synth_cotangent(3)).cuda()
    t = _time(lambda: (net(x), net.vjp(params, x, g)), 20, torch)
    out["cfg3_fwd_plus_vjp"] = {"evals_per_s": B / t, "ms": t * 1e3, "batch": B}
    # full training step (scripts/train_nmpc.py:258-300): forward + loss/seeds + VJP + clip/Adam, on device
    from irbfn_amd import train
    state = train.TrainState.create(net, configs.synth_params(3), lr=1e-3, max_grad_norm=1.0)
    yt = torch.from_numpy(configs.synth_cotangent(3)).cuda()
    t = _time(lambda: train.train_step_oneint(state, x, yt, configs.DYN_PARAMS), 20, torch)
    out["cfg3_train_step_oneint"] = {"evals_per_s": B / t, "ms": t * 1e3, "batch": B,
                                     "what": "fwd + loss seeds + param VJP + clip_by_global_norm + adam, no host sync"}
    net.bind(params)
    # roll-out: 32768 trajectories (cfg-4 per-GPU share), T = 50, kinematic single track
    Bt, T = 32768, 50
    st0 = configs.initial_state_from_query(x[:Bt].cpu().numpy())
    u = np.random.default_rng(5).normal(0, 2.0, size=(Bt, 2 * T)).astype(np.float32)
    xu = torch.from_numpy(np.hstack([st0, u])).cuda()
    t = _time(lambda: dynamics.integrate_st_ks_mult(xu, configs.DYN_PARAMS), 50, torch)
    rbytes = 4 * Bt * (7 + 2 * T + T * 7)
    out["rollout_st_ks_T50"] = {"traj_per_s": Bt / t, "us": t * 1e6, "batch": Bt,
                                "hbm_GBs": rbytes / t / 1e9, "hbm_frac": rbytes / t / 1e9 / PEAK_HBM_GBS,
                                "algorithmic_bytes": rbytes,
                                "note": "per-GPU share of cfg-4: 512 waves on 1024 SIMDs -> bound by one wave's serial "
                                        "latency (50 dependent steps), not by HBM; see the whole-batch entry"}
    # the whole cfg-4 batch on ONE GPU: the size at which the roll-out is actually HBM-bound
    Bw = 262144
    xw = configs.synth_queries(4, B=Bw)
    stw = configs.initial_state_from_query(xw)
    uw = np.random.default_rng(6).normal(0, 2.0, size=(Bw, 2 * T)).astype(np.float32)
    xuw = torch.from_numpy(np.hstack([stw, uw])).cuda()
    t = _time(lambda: dynamics.integrate_st_ks_mult(xuw, configs.DYN_PARAMS), 20, torch)
    wbytes = 4 * Bw * (7 + 2 * T + T * 7)
    out["rollout_st_ks_T50_whole_cfg4_batch"] = {"traj_per_s": Bw / t, "us": t * 1e6, "batch": Bw,
                                                 "hbm_GBs": wbytes / t / 1e9, "hbm_frac": wbytes / t / 1e9 / PEAK_HBM_GBS,
                                                 "algorithmic_bytes": wbytes}
    del xuw
    # fused planning tick, cfg-4 per-GPU share: 4096 centres, O = 100, B = 32768
    card4 = configs.model_card(4)
    net4 = WCRBFNet.from_config(card4)
    from irbfn_amd import distributed
    p4 = distributed.params_to_device(configs.synth_params(4))
    x4 = x[:Bt].contiguous()
    s0 = torch.from_numpy(st0).cuda()
    net4.bind(p4)
    t = _time(lambda: plan_batch(net4, p4, x4, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS,
                                 return_controls=False), 10, torch)
    out["cfg4_fused_plan_tick"] = {"traj_per_s": Bt / t, "ms": t * 1e3, "batch": Bt,
                                   "tflops": Bt * 4096 * (3 * 7 + 2 + 200) / t / 1e12}
    # BASELINE config 5: 16384-centre inverse-multiquadric net, B = 2^20 -- fp32 VALU kernel (K1) vs the reduction
    # "cast as MFMA GEMM": K1h at float32 accuracy (hi/lo f16 operand split) and with plain f16 operands
    card5 = configs.model_card(5)
    net5 = WCRBFNet.from_config(card5)
    p5 = distributed.params_to_device(configs.synth_params(5))
    x5 = torch.from_numpy(configs.synth_queries(5)).cuda()
    net5.bind(p5)
    B5, N5 = x5.shape[0], card5["num_kernels"]
    pairs = float(B5) * N5
    res5 = {"batch": B5, "centres": N5, "basis": card5["basis_func"]}
    ref_out = None
    for key, env in (("fp32_valu_K1", {"IRBFN_FWD_F16": "0"}),
                     ("f16x3_mfma_K1h_fp32_accurate", {"IRBFN_FWD_F16": "1", "IRBFN_FWD_F16_TERMS": "3"}),
                     ("f16_mfma_K1h_reduced_precision", {"IRBFN_FWD_F16": "1", "IRBFN_FWD_F16_TERMS": "1"})):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            t = _time(lambda: net5(x5), 3, torch)
            o5 = net5(x5)[:4096].float().cpu().numpy()
            kern = net5.last_launch()["kernel"]
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if ref_out is None:
            ref_out = o5
        entry = {"ms": t * 1e3, "evals_per_s": B5 / t, "kernel": kern,
                 "fp32_equiv_tflops": pairs * (3 * 7 + 2 + 2 * 10) / t / 1e12,
                 "max_rel_dev_vs_fp32_kernel": float(np.abs(o5 - ref_out).max() / np.abs(ref_out).max())}
        if "K1h" in key:
            # matrix-core share: MFMA instructions of 16 cycles per 16x16x32 tile product, over the kernel's SIMD time
            terms = 3 if "x3" in key else 1
            mfma_cycles = pairs / (16 * 32) * terms * 16.0
            entry["mfma_busy_frac"] = mfma_cycles / (t * 2.4e9 * 1024)
        res5[key] = entry
    out["cfg5_imq_16384_centres"] = res5
    return out


if __name__ == "__main__":
    main()
