"""bench.py contract (one JSON line with roofline / cpu_baseline objects) and the N > 1 launch path, rehearsed on
ONE GPU: two ranks share cuda:0 over gloo (IRBFN_BENCH_SAME_DEVICE / IRBFN_DIST_BACKEND are rehearsal knobs; the
driver launches one rank per GPU over RCCL)."""
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
    assert j["unit"] == "evals/s" and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32"
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and rf["bound"] in ("hbm", "mfma")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0
    cb = j["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert cb["parity_rel_err_vs_f64"] < 1e-5


def test_two_ranks_share_the_gpu_over_gloo(gpu):
    env = dict(os.environ, IRBFN_BENCH_SAME_DEVICE="1", IRBFN_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "5",
                        "--warmup", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)                      # ONE line (rank 0), no extras / CPU baseline at N > 1
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    assert j["cpu_baseline"] is None and "extras" not in j and j["value"] > 0
