"""bench.py contract (one JSON line with roofline / cpu_baseline objects) and the N > 1 launch path, rehearsed on
ONE GPU: two ranks share cuda:0 over gloo (IRBFN_BENCH_SAME_DEVICE / IRBFN_DIST_BACKEND are rehearsal knobs; the
driver launches one rank per GPU over RCCL), the DIRECT form ``python bench.py --gpus 2`` (bench.py starts its own
ranks), and ONE rank over backend "nccl" (IRBFN_BENCH_FORCE_DIST keeps the group alive at world size 1) so that the
RCCL branch -- init with device_id, broadcast, all-reduce, max-over-ranks on CUDA tensors, barrier, destroy -- has
executed on the box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line(gpu):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-extras", "--cpu-sample", "2048"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert REQUIRED <= set(j) and j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2
    assert j["unit"] == "evals/s" and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"].startswith("f32")
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] in ("hbm", "mfma")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0
    # the product path, full precision: K1g (distances and Phi x W on the matrix cores) or K1h with (hi, lo) operand pairs
    assert rf["kernel"].startswith("rbf_fwd_f16gram<") or (rf["kernel"].startswith("rbf_fwd_f16mfma<") and "TERMS=3" in rf["kernel"])
    assert {"valu_f32", "mfma_f16"} <= set(rf["by_unit"]) and rf["by_unit"]["mfma_f16"]["frac"] < 1.0
    cb = j["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert cb["parity_rel_err_vs_f64"] < 1e-5


def _rank_free_env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.update(kw)
    return env


def test_two_ranks_under_torchrun_share_the_gpu_over_gloo(gpu):
    """The launcher form (what the task statement shows for N > 1), headline only."""
    env = _rank_free_env(IRBFN_BENCH_SAME_DEVICE="1", IRBFN_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "5",
                        "--warmup", "2", "--no-extras"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["process_group"] == {"backend": "gloo", "world_size": 2} and j["value"] > 0


def test_one_rank_over_rccl(gpu):
    """backend="nccl" (= RCCL) end to end on one GPU: a process group of ONE rank; every collective of the N > 1 path
    runs on CUDA tensors (parameter broadcast, max-over-ranks, barrier, the gradient all-reduce of the cfg-3 extra)."""
    env = _rank_free_env(IRBFN_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29534", "bench.py", "--gpus", "1", "--steps", "5",
                        "--warmup", "2", "--no-cpu-baseline", "--scaling-extras-only"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["process_group"] == {"backend": "nccl", "world_size": 1} and j["value"] > 0
    ex = j["extras"]
    assert ex["broadcast_params_cfg4"]["ms"] > 0 and ex["cfg3_fwd_vjp_allreduce_weak"]["evals_per_s"] > 0
    assert ex["cfg4_plan_tick_strong"]["batch_per_gpu"] == 262144


def test_direct_form_starts_its_own_ranks(gpu):
    """``python bench.py --gpus 2 ...`` exactly as the driver types it (no RANK in the environment): bench.py starts
    the two ranks as a child process tree before touching the GPU and relays rank 0's line and the return code."""
    env = _rank_free_env(IRBFN_BENCH_SAME_DEVICE="1", IRBFN_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)                      # ONE line (rank 0); the CPU baseline is an N = 1 item
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    assert j["cpu_baseline"] is None and j["value"] > 0
    # the N-rank numbers the metric names: strong-scaled trajectories/s, weak-scaled fwd + VJP + all-reduce, broadcast
    ex = j["extras"]
    tick, fb = ex["cfg4_plan_tick_strong"], ex["cfg3_fwd_vjp_allreduce_weak"]
    assert tick["global_batch"] == 262144 and tick["batch_per_gpu"] == 131072 and tick["traj_per_s"] > 0
    assert fb["global_batch"] == 2 * fb["batch_per_gpu"] and fb["evals_per_s"] > 0
    assert ex["broadcast_params_cfg4"]["bytes"] > 1_000_000 and ex["broadcast_params_cfg4"]["ms"] > 0
    assert "cfg5_imq_16384_centres" not in ex     # single-GPU kernel numbers are reported at N = 1 only
