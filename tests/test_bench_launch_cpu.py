"""The launch contract of bench.py, CPU side: ``python bench.py --gpus N`` with N > 1 and no RANK in the environment
starts its own ranks under ``torch.distributed.run`` as a child process tree (never an exec, and before any GPU call);
a process that IS a rank, or N = 1, runs in place."""
import argparse
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_RANK_VARS = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command(monkeypatch):
    b = _bench()
    for v in _RANK_VARS:
        monkeypatch.delenv(v, raising=False)
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    cmd = b.self_launch_cmd(argparse.Namespace(gpus=4), argv)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert b.self_launch_cmd(argparse.Namespace(gpus=1), ["--gpus", "1"]) is None          # N = 1 runs in place
    monkeypatch.setenv("RANK", "0")
    assert b.self_launch_cmd(argparse.Namespace(gpus=4), argv) is None                      # a rank never re-launches


def test_direct_form_takes_the_spawn_path_without_a_gpu():
    """No GPU here: each of the two ranks that bench.py starts stops at its 'needs a GPU' check, the parent relays the
    failure as its own return code.  Two such messages = two ranks were started by `python bench.py --gpus 2`."""
    env = {k: v for k, v in os.environ.items() if k not in _RANK_VARS}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():                 # on a GPU box the same command is covered by the -m gpu tests
        return
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr[-3000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
