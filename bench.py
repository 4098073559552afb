#!/usr/bin/env python3
"""Headline benchmark of the IRBFN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks as a child process tree)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): RBF-net evals/s (+ trajectories/s, fwd+bwd in ``extras``).  One "step" = one fused forward
pass of the WCRBFNet (region gate + 4096 Gaussian centres + Dense) over one batch of 65 536 synthetic 7-D queries
(BASELINE config 2), inputs and parameters resident in HBM.  Weak scaling: every rank processes its own
65 536-query shard; rank 0's parameters are broadcast once over RCCL before the timed region and there is no
collective in the steady state.

Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` (dominant kernel, HIP-event
timing on the launch stream), ``cpu_baseline`` (the C restatement of the reference path, oracle/, timed on this
box's host cores on a bounded sample; N = 1 only) and ``extras``.  At every N the extras hold the multi-GPU
numbers the metric names, each timed like the headline (barrier + synchronize on both sides, MAX over ranks):
config 4 STRONG-scaled (262 144 planning ticks split over the ranks: forward O = 100 + 50-step roll-out ->
trajectories/s), config 3 weak-scaled forward + parameter VJP + ONE gradient all-reduce (evals/s), and the
parameter broadcast.  At N = 1 the single-GPU kernel numbers (roll-outs, config 5 variants, ...) follow.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector peak == dense f32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (the 5 PF headline figure includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E spec peak
DTYPE = "f32 (distances and Phi x W on f16 MFMA: exact fixed-point heads + float tails, hi/lo operand pairs; f32 accumulate)"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, default=2, help="BASELINE config index (2 = headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--scaling-extras-only", action="store_true",
                    help="extras: only the multi-GPU numbers (broadcast, cfg-4 strong, cfg-3 weak), also at N = 1")
    ap.add_argument("--cpu-sample", type=int, default=65536)
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_cmd(args, argv):
    """``python bench.py --gpus N`` typed directly (no RANK in the environment) with N > 1: the command that starts the
    N ranks -- one process per GPU under ``torch.distributed.run`` -- or None when this process is itself a rank (or
    N = 1).  Pure function of (args, argv, environment) so that the CPU suite can check it."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]


def self_launch(args, argv) -> None:
    """Runs the ranks as a FRESH child process tree and exits with its return code.  Must be called before anything
    touches the GPU (never exec after HIP is initialised; this parent never initialises it).  Rank 0's JSON line goes
    straight to the inherited stdout."""
    cmd = self_launch_cmd(args, argv)
    if cmd is None:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, available_cores() // args.gpus)))
    sys.stdout.flush()
    r = subprocess.run(cmd, env=env, cwd=ROOT)
    raise SystemExit(r.returncode)


def _dist_on() -> bool:
    import torch
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world} (RANK is set: this process was started as a rank of "
                         f"another job size; start `python bench.py --gpus {n_gpus}` without RANK/WORLD_SIZE and it "
                         f"launches its own ranks)")
    # IRBFN_BENCH_FORCE_DIST=1 (rehearsal knob): keep a process group alive at world size 1 too, so that one rank on a
    # one-GPU box executes the whole RCCL branch (init with device_id, broadcast, all-reduce, max-over-ranks, barrier)
    if world > 1 or os.environ.get("IRBFN_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal knobs (one-GPU box): IRBFN_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
        # IRBFN_DIST_BACKEND=gloo avoids RCCL's one-rank-per-device rule; the driver uses neither.
        dev = 0 if os.environ.get("IRBFN_BENCH_SAME_DEVICE") == "1" else local
        backend = os.environ.get("IRBFN_DIST_BACKEND", "nccl")
        if dev >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local} has no GPU (this node shows {torch.cuda.device_count()})")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    return rank, world, local


def _barrier(world):
    import torch
    torch.cuda.synchronize()
    if _dist_on():
        torch.distributed.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(v, world):
    import torch
    if not _dist_on():
        return v
    t = torch.tensor([v], dtype=torch.float64)
    if torch.distributed.get_backend() == "nccl":
        t = t.cuda()
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def timed_region(fn, steps, warmup, world):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks.
    Also returns the HIP-event time of the same K launches on the launch stream (this rank)."""
    import torch
    for _ in range(warmup):
        fn()
    _barrier(world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    _barrier(world)
    wall = time.perf_counter() - t0
    return _max_over_ranks(wall, world), e0.elapsed_time(e1)


def main():
    args = parse()
    self_launch(args, sys.argv[1:])              # --gpus N > 1 typed directly: start the ranks, relay, exit
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

    wall, ev_ms = timed_region(lambda: net(x), args.steps, args.warmup, world)
    launch = net.last_launch()                   # the kernel of the timed region (the extras launch others)
    ms_per_step = wall / args.steps * 1e3
    value = B * world * args.steps / wall

    # every rank stays in the group until the end: the multi-GPU extras below are collective
    extras = None
    if not args.no_extras and idx == 2:
        extras = scaling_extras(net, params, x, rank, world, configs, torch)
        if world == 1 and not args.scaling_extras_only:
            extras.update(single_gpu_extras(net, params, x, configs, torch))
            net.bind(params)

    if rank == 0:
        roofline = forward_roofline(launch, B, N, D, O, ev_ms / 1e3 / args.steps)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(net, card, params, x, B, args.cpu_sample)
        line = {
            "metric": "RBF-net evals/s (B queries x N centres), forward", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic",
            "config": {"workload": f"BASELINE config {idx}: {N} centres, d={D}, O={O}, batch={B} per GPU, "
                                   f"{card['basis_func']} RBF forward (gate + RBF + Dense fused)",
                       "centres": N, "in_features": D, "out_features": O, "batch_per_gpu": B,
                       "global_batch": B * world, "basis": card["basis_func"], "parallelism": f"dp{world} (query shards)"},
            "pair_evals_per_s": value * N,
            "roofline": roofline, "cpu_baseline": cpu,
            "process_group": ({"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size()}
                              if _dist_on() else None),
        }
        if extras:
            line["extras"] = extras
        print(json.dumps(line), flush=True)
    if _dist_on():
        _barrier(world)
        torch.distributed.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
def kernel_fingerprint(names=("rbf_forward_gram.hip", "rbf_forward_gram.h", "pack_all.hip", "rbf_forward_f16_narrow.h", "rbf_forward_f16.hip", "rbf_forward_f16_wide.h", "f16_split.h",
                              "rbf_forward.h", "rbf_forward.hip")) -> str:
    """Hash of the sources the headline kernel is built from: profiles/*_traffic.json is only trusted for the
    code it was measured on (tools/measure_traffic.py stamps it)."""
    h = hashlib.sha1()
    for n in names:
        with open(os.path.join(ROOT, "irbfn_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def forward_roofline(launch, B, N, D, O, kern_s):
    """Roofline object of the fused forward: algorithmic work / HIP-event time per launch."""
    flops = B * N * (3 * D + 2 + 2 * O)                          # per pair 3D + 2 + 2O
    abytes = 4 * (B * D + N * D + N + N * O + O + B * O)         # every tensor touched once
    traffic, traffic_src = None, "none measured for this build (run tools/measure_traffic.py on the GPU box)"
    import glob
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):      # newest round first
        rec = json.load(open(tpath)).get(launch["kernel"])
        tname = "profiles/" + os.path.basename(tpath)
        if rec and rec.get("fingerprint") == kernel_fingerprint() and rec.get("grid") == launch["grid"]:
            traffic = rec["bytes"]
            traffic_src = (f"{tname}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this "
                           "command, per launch; source fingerprint and launch geometry match this build")
            break
        elif rec:
            traffic_src = f"{tname} is stale (kernel sources or launch geometry changed since it was measured)"
    # the kernel's work by execution unit
    gram = launch["kernel"].startswith("rbf_fwd_f16gram")
    if gram:
        # K1g: per 32 x 32 pairs 4 head MFMAs 16x16x16 + 8 tail MFMAs 16x16x32 (the squared distances as a Gram expansion:
        # 10 + 42 of the 16 + 64 k-slots carry products) + 6 MFMAs 16x16x32 (Phi x W as 3 f16 products, O padded to 16);
        # the VALU keeps one transcendental and the hi/lo split (3 instructions) per pair -- no algorithmic flops
        valu_flops = 0.0
        mfma_flops = B * N * (4 * 8192 + 14 * 16384) / 1024.0
        detail = "compute roof; issue bound: the MFMA issue and the VALU issue of a SIMD add up (see note)"
        note = ("`bound` takes the contract's two values: 'mfma' here means the COMPUTE roof (as opposed to 'hbm'), priced as "
                "SURVEY 8d prescribes: algorithmic f32 flops (B*N*(3D+2+2O), transcendental count B*N) over the fp32 peak "
                "157.3 TFLOP/s.  K1g evaluates BOTH GEMM-shaped pieces on the f16 matrix cores (the squared distances as an "
                "exactly-cancelling Gram expansion, Phi x W as (hi, lo) pairs: 256 issued f16 flops per pair, 18 MFMAs per "
                "32 x 32 pairs) and keeps one transcendental + a 3-instruction operand split per pair on the VALU; on a SIMD the "
                "two issue streams add up (PMC: profiles/r03_gram_pmc.txt), so the kernel is bound by their sum, not by HBM "
                "(2400 flop/B) and not by the matrix cores alone; `by_unit` prices the two pipes separately")
        what_valu = "no algorithmic flops: B*N transcendentals and the hi/lo operand split (3 VALU instructions per pair)"
        what_mfma = "issued f16 MFMA flops: distances (4 x 16x16x16 + 8 x 16x16x32 per 1024 pairs) + Phi x W (6 x 16x16x32)"
    else:
        # K1h: the Phi x W products run on the f16 matrix cores as 3 f16 products per f32 product (ph*wh, pl*wh, ph*wl), the
        # outputs padded to one 16-wide tile; the rest is f32 VALU + 1 transcendental
        valu_flops = B * N * (3 * D + 2)
        mfma_flops = 2.0 * B * N * 16 * 3
        detail = "compute roof; VALU-issue bound in fact (see note)"
        note = ("`bound` takes the contract's two values: 'mfma' here means the COMPUTE roof (as opposed to 'hbm'), priced as "
                "SURVEY 8d prescribes: algorithmic f32 flops (B*N*(3D+2+2O), transcendental count B*N) over the fp32 peak "
                "157.3 TFLOP/s (fp32 vector == dense f32-input MFMA peak).  Within that roof the kernel is VALU/"
                "transcendental-ISSUE bound (2400 flop/B, ~21.5 VALU instructions per pair; the matrix cores are ~7 % busy), "
                "not MFMA-throughput bound and not HBM-bound; `by_unit` prices the two pipes it runs on separately")
        what_valu = ("distances + basis argument (3D+2 per pair) on the f32 VALU; + B*N transcendentals and "
                     "the hi/lo operand split (3.5 VALU instructions per pair, no algorithmic flops)")
        what_mfma = "issued f16 MFMA flops: 3 products x 16-wide output tile (O = 10 padded) per pair"
    return {
        "bound": "mfma", "bound_detail": detail, "achieved": flops / kern_s / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
        "frac": flops / kern_s / 1e12 / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
        "kernel": launch["kernel"], "grid": launch["grid"], "block": launch["block"],
        "avg_launch_us": kern_s * 1e6, "algorithmic_flops": flops, "algorithmic_bytes": abytes,
        "note": note,
        "by_unit": {
            "valu_f32": {"flops": valu_flops, "tflops": valu_flops / kern_s / 1e12, "peak": PEAK_FP32_TFLOPS,
                         "frac": valu_flops / kern_s / 1e12 / PEAK_FP32_TFLOPS, "what": what_valu},
            "mfma_f16": {"flops": mfma_flops, "tflops": mfma_flops / kern_s / 1e12, "peak": PEAK_F16_MFMA_TFLOPS,
                         "frac": mfma_flops / kern_s / 1e12 / PEAK_F16_MFMA_TFLOPS, "what": what_mfma}},
        "hbm": {"achieved": abytes / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": abytes / kern_s / 1e9 / PEAK_HBM_GBS},
    }


def cpu_baseline(net, card, params, x, B, sample):
    """C/OpenMP restatement of the reference path (oracle/), bounded sample, same box (kind = "port": the
    reference itself is JAX and cannot run here), + torch-CPU broadcast form."""
    import torch
    from oracle import c_oracle as co          # checker / baseline leg only
    ns = min(sample, B)
    co.set_num_threads(available_cores())
    xs = x[:ns].cpu().numpy()
    pn = params_np(params)
    co.wcrbf_forward(card, pn, xs[:4096], np.float32)      # warm-up (thread pool)
    dts = []
    for _ in range(5):                         # ~10-30 core-seconds of CPU work in total
        t0 = time.perf_counter()
        ref = co.wcrbf_forward(card, pn, xs, np.float32)
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[2]
    got = net(x[:ns]).cpu().numpy()
    ref64 = co.wcrbf_forward(card, pn, xs[:1024], np.float64)
    out = {"value": ns / dt, "unit": "evals/s", "cores": co.num_threads(), "kind": "port",
           "sample": f"{ns} of the {B} queries of the same workload x 5 repeats (median), float32, OpenMP C "
                     f"restatement of the reference path (oracle/irbfn_oracle.c), {dt:.3f} s wall per repeat "
                     f"(~{5 * dt * co.num_threads():.0f} core-seconds of CPU work in total)",
           "parity_rel_err_vs_f64": float(np.abs(got[:1024] - ref64).max() / np.abs(ref64).max()),
           "parity_rel_err_vs_cpu_f32": float(np.abs(got - ref).max() / np.abs(ref).max())}
    # XLA-like vectorised leg: torch-CPU float32, the explicit broadcast of flax_rbf.py:275-283 chunked over B
    try:
        from oracle import irbfn_oracle as orc
        torch.set_num_threads(co.num_threads())
        nt = min(ns, 8192)
        xt = torch.from_numpy(xs[:nt])
        pt = {"params": {g: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()} for g, d in pn["params"].items()}}
        with torch.no_grad():
            orc.wcrbfnet_apply(card, pt, xt[:1024])
            t0 = time.perf_counter()
            for i in range(0, nt, 1024):
                orc.wcrbfnet_apply(card, pt, xt[i:i + 1024])
            dtt = time.perf_counter() - t0
        out["torch_cpu_broadcast_form"] = {"value": nt / dtt, "unit": "evals/s", "cores": co.num_threads(),
                                           "sample": f"{nt} queries, torch-CPU float32, chunks of 1024"}
    except Exception as e:                      # the secondary leg must never sink the line
        out["torch_cpu_broadcast_form"] = {"error": repr(e)[:200]}
    return out


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


# ---------------------------------------------------------------------------------------------------------------
def scaling_extras(net, params, x, rank, world, configs, torch):
    """The multi-GPU numbers BASELINE.json's metric names, measured at EVERY N (all ranks take part):
    broadcast of the parameters, config 4 strong-scaled planning ticks, config 3 weak-scaled fwd + VJP + all-reduce."""
    from irbfn_amd import _lib, distributed
    from irbfn_amd.model import WCRBFNet
    from irbfn_amd.planner import plan_batch
    out = {}
    # --- parameter broadcast (RCCL over xGMI at N > 1): the one collective of the forward path, at upload time
    card4 = configs.model_card(4)
    net4 = WCRBFNet.from_config(card4)
    p4_host = configs.synth_params(4) if rank == 0 else None
    distributed.broadcast_params(net4, p4_host, src=0)            # warm-up (communicator set-up)
    _barrier(world)
    t0 = time.perf_counter()
    p4 = distributed.broadcast_params(net4, p4_host, src=0)
    _barrier(world)
    out["broadcast_params_cfg4"] = {"ms": _max_over_ranks(time.perf_counter() - t0, world) * 1e3,
                                    "bytes": 4 * distributed.flat_param_count(net4),
                                    "what": "host pytree -> flat device buffer -> ONE broadcast -> views (includes the H2D copy on rank 0)"}
    net4.bind(p4)
    # --- config 4, STRONG scaling: 262144 (start, goal) pairs split over the ranks; one tick = forward (4096
    #     centres, O = 100) + 50-step kinematic single-track roll-out; no collective in the steady state
    Bt = configs.batch_size(4)
    lo, hi = distributed.shard_range(Bt, rank, world)
    x4 = torch.from_numpy(configs.synth_queries(4, B=Bt)[lo:hi]).cuda()
    s0 = torch.from_numpy(configs.initial_state_from_query(x4.cpu().numpy())).cuda()
    steps4 = 20 if world > 1 else 10
    tick = lambda: plan_batch(net4, p4, x4, s0, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS, return_controls=False)
    wall, _ = timed_region(tick, steps4, 3, world)
    out["cfg4_plan_tick_strong"] = {"traj_per_s": Bt * steps4 / wall, "ms_per_tick": wall / steps4 * 1e3,
                                    "global_batch": Bt, "batch_per_gpu": hi - lo, "scaling": "strong", "steps": steps4,
                                    "what": "forward O=100 (f16-MFMA wide kernel) + 50-step ST-kinematic roll-out per rank, one launch; "
                                            "barrier + synchronize on both sides, max over ranks"}
    del x4, s0
    # --- config 3, WEAK scaling: forward + parameter VJP on each rank's 65536-query shard + ONE all-reduce of the
    #     flat gradient buffer (0.3 MB)
    B = x.shape[0]
    g = torch.from_numpy(configs.synth_cotangent(3, seed=2123 + rank)).cuda()
    steps3 = 20

    def fwd_bwd():
        net(x)
        return distributed.allreduce_grads(net, net.vjp(params, x, g))
    wall, _ = timed_region(fwd_bwd, steps3, 3, world)
    out["cfg3_fwd_vjp_allreduce_weak"] = {"evals_per_s": B * world * steps3 / wall, "ms_per_step": wall / steps3 * 1e3,
                                          "batch_per_gpu": B, "global_batch": B * world, "scaling": "weak", "steps": steps3,
                                          "what": "fused forward + parameter VJP (centres, widths, weights, bias) + one "
                                                  "all-reduce(sum) of the flat gradient buffer (identity at N = 1)"}
    return out


def single_gpu_extras(net, params, x, configs, torch):
    """Secondary single-GPU kernel numbers (outside the timed region)."""
    from irbfn_amd import _lib, distributed, dynamics, train
    from irbfn_amd.model import WCRBFNet
    from irbfn_amd.planner import plan_batch
    out = {}
    B = x.shape[0]
    N, D, O = 4096, 7, 10
    pairs2 = float(B) * N
    # the all-float32 VALU kernel on the headline workload (the kernel the f16-MFMA one replaced)
    net.set_options(fwd_kernel=_lib.FWD_K1)
    t = _time(lambda: net(x), 50, torch)
    out["cfg2_fp32_valu_K1"] = {"us": t * 1e6, "evals_per_s": B / t, "kernel": net.last_launch()["kernel"],
                                "fp32_tflops": pairs2 * (3 * D + 2 + 2 * O) / t / 1e12,
                                "frac_of_fp32_peak": pairs2 * (3 * D + 2 + 2 * O) / t / 1e12 / PEAK_FP32_TFLOPS}
    # K1h: Phi x W on the matrix cores, the distances on the VALU (the headline kernel until K1g)
    net.set_options(fwd_kernel=_lib.FWD_K1H)
    t = _time(lambda: net(x), 50, torch)
    out["cfg2_valu_distances_K1h"] = {"us": t * 1e6, "evals_per_s": B / t, "kernel": net.last_launch()["kernel"],
                                      "frac_of_fp32_peak": pairs2 * (3 * D + 2 + 2 * O) / t / 1e12 / PEAK_FP32_TFLOPS}
    net.set_options(fwd_kernel=_lib.FWD_AUTO)
    g = torch.from_numpy(configs.synth_cotangent(3)).cuda()
    t = _time(lambda: (net(x), net.vjp(params, x, g)), 20, torch)
    out["cfg3_fwd_plus_vjp"] = {"evals_per_s": B / t, "ms": t * 1e3, "batch": B}
    # full training step (scripts/train_nmpc.py:258-300): forward + loss/seeds + VJP + clip/Adam, on device
    state = train.TrainState.create(net, configs.synth_params(3), lr=1e-3, max_grad_norm=1.0)
    yt = torch.from_numpy(configs.synth_cotangent(3)).cuda()
    t = _time(lambda: train.train_step_oneint(state, x, yt, configs.DYN_PARAMS), 20, torch)
    out["cfg3_train_step_oneint"] = {"evals_per_s": B / t, "ms": t * 1e3, "batch": B,
                                     "what": "fwd + loss seeds + param VJP + clip_by_global_norm + adam, no host sync"}
    net.bind(params)
    # the planning tick of a narrow net (O = 10 = 2 x 5 knots, what the reference's trained planners are): forward +
    # sign flip + 5-step kinematic roll-out in ONE launch of the matrix-core kernel, beside the forward alone
    s2 = torch.from_numpy(configs.initial_state_from_query(x.cpu().numpy())).cuda()
    tick2 = lambda: plan_batch(net, params, x, s2, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    tfw, ttk = [], []
    for _ in range(3):
        tfw.append(_time(lambda: net(x), 30, torch))
        ttk.append(_time(tick2, 30, torch))
    k2 = net.last_launch()["kernel"]
    out["cfg2_plan_tick_T5"] = {"batch": B, "forward_us": sorted(tfw)[1] * 1e6, "one_launch_us": sorted(ttk)[1] * 1e6,
                                "traj_per_s": B / sorted(ttk)[1], "kernel": k2,
                                "what": "config-2 net (O = 10): forward + 5-step ST-kinematic roll-out, controls and states written; "
                                        "medians of 3 interleaved rounds x 30"}
    del s2
    # roll-out alone (the HBM-bound kernel): T = 50, kinematic single track; per-GPU share and whole cfg-4 batch
    T = 50
    for key, Bt, reps in (("rollout_st_ks_T50_per_gpu_share", 32768, 50), ("rollout_st_ks_T50_whole_cfg4_batch", 262144, 20)):
        xq = configs.synth_queries(4, B=Bt)
        st0 = configs.initial_state_from_query(xq)
        u = np.random.default_rng(5).normal(0, 2.0, size=(Bt, 2 * T)).astype(np.float32)
        xu = torch.from_numpy(np.hstack([st0, u])).cuda()
        states = dynamics.integrate_st_ks_mult(xu, configs.DYN_PARAMS)
        t = _time(lambda: dynamics.integrate_st_ks_mult(xu, configs.DYN_PARAMS), reps, torch)
        rbytes = 4 * Bt * (7 + 2 * T + T * 7)
        out[key] = {"traj_per_s": Bt / t, "us": t * 1e6, "batch": Bt, "algorithmic_bytes": rbytes,
                    "roofline": {"bound": "hbm", "achieved": rbytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": rbytes / t / 1e9 / PEAK_HBM_GBS},
                    "kernel": "rollout_fwd_pair_kernel<ST_KS> (2 lanes per trajectory, whole-line non-temporal stores)"}
        # "also report ST-select" (integrate_st_mult: both right-hand sides evaluated, selected by V > 3)
        t_sel = _time(lambda: dynamics.integrate_st_mult(xu, configs.DYN_PARAMS), max(reps // 2, 5), torch)
        out[key]["st_select"] = {"traj_per_s": Bt / t_sel, "us": t_sel * 1e6, "hbm_GBs": rbytes / t_sel / 1e9,
                                 "hbm_frac": rbytes / t_sel / 1e9 / PEAK_HBM_GBS}
        del xu, states
    # config 4 forward alone at the per-GPU share, by execution unit
    card4 = configs.model_card(4)
    net4 = WCRBFNet.from_config(card4)
    p4 = distributed.params_to_device(configs.synth_params(4))
    net4.bind(p4)
    x4 = torch.from_numpy(configs.synth_queries(4, B=32768)).cuda()
    t = _time(lambda: net4(x4), 20, torch)
    pairs4 = 32768.0 * 4096
    kern4 = net4.last_launch()["kernel"]
    gram4 = kern4.startswith("rbf_fwd_f16gram_wide")
    # issued f16 MFMA flops per pair: Phi x W 3 products x 7 column tiles of 16 (O = 100 padded to 112); K1g adds the distances
    # (4 x 16x16x16 + 8 x 16x16x32 per 32 x 32 pairs = 160 per pair) and takes the 3D + 2 distance flops off the VALU
    mf4 = pairs4 * (2 * 112 * 3 + (160 if gram4 else 0))
    out["cfg4_forward_O100_per_gpu_share"] = {
        "us": t * 1e6, "evals_per_s": 32768 / t, "kernel": kern4,
        "valu_f32": {"tflops": (0.0 if gram4 else pairs4 * 23 / t / 1e12), "frac": (0.0 if gram4 else pairs4 * 23 / t / 1e12 / PEAK_FP32_TFLOPS)},
        "mfma_f16": {"tflops": mf4 / t / 1e12, "frac": mf4 / t / 1e12 / PEAK_F16_MFMA_TFLOPS,
                     "what": "issued f16 MFMA flops: 3 products x 7 column tiles of 16 (O = 100 padded to 112)"
                             + (" + the squared distances as a Gram expansion (K1g)" if gram4 else "")},
        "algorithmic_fp32_tflops": pairs4 * (3 * 7 + 2 + 200) / t / 1e12}
    net4.set_options(fwd_kernel=_lib.FWD_K1H)
    t_h = _time(lambda: net4(x4), 20, torch)
    out["cfg4_forward_O100_per_gpu_share"]["K1h_wide_us"] = t_h * 1e6
    net4.set_options(fwd_kernel=_lib.FWD_AUTO)
    # the planning tick at the per-GPU share: forward + 50-step roll-out in ONE launch (controls stay in LDS) against
    # forward -> roll-out as separate launches through a controls buffer, and the forward alone (interleaved repeats)
    s4 = torch.from_numpy(configs.initial_state_from_query(x4.cpu().numpy())).cuda()
    tick4 = lambda: plan_batch(net4, p4, x4, s4, configs.DYN_PARAMS, mode=_lib.ROLLOUT_ST_KS)
    tf, t1, t2 = [], [], []
    for _ in range(3):
        tf.append(_time(lambda: net4(x4), 20, torch))
        net4.set_options(tick_fused=1)
        t1.append(_time(tick4, 20, torch))
        k1 = net4.last_launch()["kernel"]
        net4.set_options(tick_fused=0)
        t2.append(_time(tick4, 20, torch))
        net4.set_options(tick_fused=1)
    med = lambda v: sorted(v)[len(v) // 2]
    out["cfg4_plan_tick_per_gpu_share"] = {
        "batch": 32768, "forward_us": med(tf) * 1e6, "one_launch_us": med(t1) * 1e6, "separate_launches_us": med(t2) * 1e6,
        "one_launch_minus_forward_us": (med(t1) - med(tf)) * 1e6, "traj_per_s": 32768 / med(t1), "kernel": k1,
        "what": "forward O=100 + 50-step ST-kinematic roll-out, controls and states written; medians of 3 interleaved rounds x 20"}
    del x4, s4
    # the reference's own trained planners (decoded checkpoints committed as fixtures under tests/golden/; 1000-1280 centres,
    # one / 12 / 128 regions): forward at B = 65536 uniform in-range queries
    import json as _json
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    trained = {}
    for run in ("dnmpc_1regions_newdata_oldintloss_nomirror_highk", "dnmpc_128regions", "dnmpc_12regions_frenet_l1_bigdata"):
        fz, fj = os.path.join(gdir, f"ckpt_{run}.npz"), os.path.join(gdir, f"ckpt_{run}.json")
        if not (os.path.exists(fz) and os.path.exists(fj)):
            continue
        z, cfg_t = np.load(fz), _json.load(open(fj))
        pt = {"params": {"rbf_list": {"centers": z["centers"].astype(np.float32), "log_sigs": z["log_sigs"].astype(np.float32)},
                         "linear": {"kernel": z["kernel"].astype(np.float32), "bias": z["bias"].astype(np.float32)}}}
        nt = WCRBFNet.from_config(cfg_t)
        nt.bind(distributed.params_to_device(pt))
        ns = len(cfg_t["activation_idx"])
        lo_t = np.array([min(cfg_t["lower_bounds"][d]) for d in range(ns)]); hi_t = np.array([max(cfg_t["upper_bounds"][d]) for d in range(ns)])
        rng_t = np.random.default_rng(1)
        xq = np.hstack([rng_t.uniform(lo_t, hi_t, size=(65536, ns)),
                        rng_t.normal(size=(65536, cfg_t["in_features"] - ns)) * 0.1]).astype(np.float32)
        xt_ = torch.from_numpy(xq).cuda()
        tt_ = _time(lambda: nt(xt_), 30, torch)
        trained[run] = {"us": tt_ * 1e6, "evals_per_s": 65536 / tt_, "regions": cfg_t["num_regions"],
                        "centres": cfg_t["num_regions"] * cfg_t["num_kernels"], "kernel": nt.last_launch()["kernel"]}
        if cfg_t["num_regions"] > 1:
            # the dense gated kernel on the same net (what the reference's model.py:187-193 evaluates: every region for every
            # query), beside the automatic choice above (region-sparse K1r where the gate is sparse)
            nt.set_options(fwd_kernel=_lib.FWD_K1)
            td_ = _time(lambda: nt(xt_), 30, torch)
            nt.set_options(fwd_kernel=_lib.FWD_AUTO)
            trained[run]["dense_K1_us"] = td_ * 1e6
            trained[run]["live_regions_per_query"] = float((nt.gate(xt_) != 0).sum(dim=1).float().mean().item())
        if cfg_t["out_features"] == 10 and cfg_t["in_features"] == 7:
            # the reference's training step at its own batch size (batch_size: 80000 in scripts/configs/*.yaml):
            # train_step_fullint = forward, loss seeds through the 5-step bicycle, parameter VJP, clip + Adam
            Bt_ = 80000
            xb = torch.from_numpy(rng_t.uniform(lo_t, hi_t, size=(Bt_, 7)).astype(np.float32)).cuda()
            yb = torch.from_numpy(np.hstack([rng_t.normal(size=(Bt_, 5)) * 2, rng_t.normal(size=(Bt_, 5)) * 0.5]).astype(np.float32)).cuda()
            st_box = [train.TrainState.create(nt, pt, lr=1e-3, max_grad_norm=1.0)]
            def _step():
                st_box[0], _ = train.train_step_fullint(st_box[0], xb, yb)
            ts_ = _time(_step, 20, torch)
            trained[run]["train_step_fullint_B80000"] = {"us": ts_ * 1e6, "evals_per_s": Bt_ / ts_}
            if cfg_t["num_regions"] > 1:
                nt.set_options(fwd_kernel=_lib.FWD_K1, vjp_kernel=_lib.VJP_K2)
                st_box[0] = train.TrainState.create(nt, pt, lr=1e-3, max_grad_norm=1.0)
                trained[run]["train_step_fullint_B80000"]["dense_K1_K2_us"] = _time(_step, 20, torch) * 1e6
                nt.set_options(fwd_kernel=_lib.FWD_AUTO, vjp_kernel=_lib.VJP_AUTO)
            del xb, yb, st_box
        del xt_, nt
    out["reference_trained_checkpoints_forward_B65536"] = trained
    # BASELINE config 5 ("fp32 vs bf16, reduction cast as MFMA GEMM, utilisation reported"): 16384-centre inverse-
    # multiquadric net, B = 2^20 -- fp32 VALU kernel (K1) vs K1h at float32 accuracy (hi/lo f16 operand pairs), with
    # plain f16 operands and with plain bf16 operands
    card5 = configs.model_card(5)
    net5 = WCRBFNet.from_config(card5)
    p5 = distributed.params_to_device(configs.synth_params(5))
    x5 = torch.from_numpy(configs.synth_queries(5)).cuda()
    net5.bind(p5)
    B5, N5 = x5.shape[0], card5["num_kernels"]
    pairs = float(B5) * N5
    res5 = {"batch": B5, "centres": N5, "basis": card5["basis_func"]}
    ref_out = None
    for key, opts in (("fp32_valu_K1", {"fwd_kernel": _lib.FWD_K1}),
                      ("f16_gram_K1g_fp32_accurate", {"fwd_kernel": _lib.FWD_K1G}),
                      ("f16x3_mfma_K1h_fp32_accurate", {"fwd_kernel": _lib.FWD_K1H, "fwd_f16_terms": 3}),
                      ("f16_mfma_K1h_reduced_precision", {"fwd_kernel": _lib.FWD_K1H, "fwd_f16_terms": 1}),
                      ("bf16_mfma_K1h_reduced_precision", {"fwd_kernel": _lib.FWD_K1H, "fwd_f16_terms": 2})):
        net5.set_options(**opts)
        t = _time(lambda: net5(x5), 3, torch)
        o5 = net5(x5)[:4096].float().cpu().numpy()
        kern = net5.last_launch()["kernel"]
        net5.set_options(fwd_kernel=_lib.FWD_AUTO, fwd_f16_terms=3)
        if ref_out is None:
            ref_out = o5
        entry = {"ms": t * 1e3, "evals_per_s": B5 / t, "kernel": kern,
                 "algorithmic_fp32_tflops": pairs * (3 * 7 + 2 + 2 * 10) / t / 1e12,
                 "valu_f32_frac": pairs * 23 / t / 1e12 / PEAK_FP32_TFLOPS,
                 "max_rel_dev_vs_fp32_kernel": float(np.abs(o5 - ref_out).max() / np.abs(ref_out).max())}
        if "K1g" in key:
            entry["valu_f32_frac"] = 0.0                     # distances on the matrix cores: no algorithmic flops left on the VALU
            entry["mfma_f16"] = {"tflops": pairs * 256 / t / 1e12, "frac": pairs * 256 / t / 1e12 / PEAK_F16_MFMA_TFLOPS,
                                 "busy_frac": pairs / 1024 * 18 * 16.0 / (t * 2.4e9 * 1024),
                                 "what": "matrix-core utilisation asked for by BASELINE config 5: issued f16 MFMA flops (distances "
                                         "as a Gram expansion + Phi x W: 18 MFMAs of 16 cycles per 32 x 32 pairs) over the dense "
                                         "f16 peak, and MFMA cycles over SIMD time at 2.4 GHz"}
        if "K1h" in key:
            terms = 3 if "x3" in key else 1
            entry["mfma_f16"] = {"tflops": pairs * 2 * 16 * terms / t / 1e12,
                                 "frac": pairs * 2 * 16 * terms / t / 1e12 / PEAK_F16_MFMA_TFLOPS,
                                 "busy_frac": pairs / (16 * 32) * terms * 16.0 / (t * 2.4e9 * 1024),
                                 "what": "matrix-core utilisation asked for by BASELINE config 5: issued f16 MFMA flops over "
                                         "the dense f16 peak, and MFMA issue cycles (16 per 16x16x32 product) over SIMD time"}
        res5[key] = entry
    out["cfg5_imq_16384_centres"] = res5
    return out


if __name__ == "__main__":
    main()
