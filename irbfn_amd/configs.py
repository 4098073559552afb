"""Synthetic BASELINE workloads (BASELINE.md section 3 / SURVEY.md section 8d).

All draws use ``numpy.random.default_rng(123)`` (123 = the reference's default seed,
``src/irbfn_mpc/arg_utils.py:106``).  The bounds of configs 2-5 are those of the reference
model card ``scripts/configs/dnmpc_1regions_newdata_oldintloss_nomirror_highk.yaml``.
"""
from __future__ import annotations

import numpy as np

# dyn_params of scripts/test_dynamics.ipynb cell 1 / irbfn_planner.py:50-66
DYN_PARAMS = [1.0, 1.0489, 0.04712, 0.15875, 0.17145, 5.0, 5.0, 0.074, 0.1, 3.2, 9.51, 0.4189, 7.0]

_LO7 = [0.0, 0.0, 0.0, -3.1, 0.0, -0.6, -3.0]
_HI7 = [7.0, 3.6, 3.6, 3.2, 7.0, 0.4, 2.5]
_LO3 = [1.0, -6.0, -1.3]
_HI3 = [30.0, 6.0, 1.3]

_SPECS = {
    # name: D, K, O, B, basis, lo, hi, delta, log_sig range, W std, centre margin
    1: dict(D=3, K=256, O=5, B=1024, basis="gaussian", lo=_LO3, hi=_HI3,
            delta=[15.0, 15.0, 100.0], ls=(0.0, 1.5), wstd=1.0, margin=0.0),
    2: dict(D=7, K=4096, O=10, B=65536, basis="gaussian", lo=_LO7, hi=_HI7,
            delta=[100.0] * 7, ls=(0.0, 2.0), wstd=1.0, margin=1.0),
    3: dict(D=7, K=4096, O=10, B=65536, basis="gaussian", lo=_LO7, hi=_HI7,
            delta=[100.0] * 7, ls=(0.0, 2.0), wstd=1.0, margin=1.0),
    4: dict(D=7, K=4096, O=100, B=262144, basis="gaussian", lo=_LO7, hi=_HI7,
            delta=[100.0] * 7, ls=(0.0, 2.0), wstd=0.3, margin=1.0),
    5: dict(D=7, K=16384, O=10, B=1048576, basis="inverse_multiquadric", lo=_LO7, hi=_HI7,
            delta=[100.0] * 7, ls=(0.0, 2.0), wstd=1.0, margin=1.0),
}


def model_card(idx: int) -> dict:
    """The YAML-equivalent model card (same fields the reference writes at
    scripts/train_nmpc.py:431-450) of BASELINE config ``idx`` (1-based)."""
    s = _SPECS[idx]
    D = s["D"]
    return {
        "in_features": D, "out_features": s["O"], "num_kernels": s["K"],
        "basis_func": s["basis"], "num_regions": 1,
        "lower_bounds": [[float(v)] for v in s["lo"]],
        "upper_bounds": [[float(v)] for v in s["hi"]],
        "dimension_ranges": [[0] * D], "activation_idx": list(range(D)),
        "delta": list(s["delta"]), "seed": 123,
    }


def synth_params(idx: int, dtype=np.float32) -> dict:
    s = _SPECS[idx]
    rng = np.random.default_rng(123)
    lo, hi = np.asarray(s["lo"]), np.asarray(s["hi"])
    D, K, O = s["D"], s["K"], s["O"]
    centers = rng.uniform(lo - s["margin"], hi + s["margin"], size=(1, K, D))
    log_sigs = rng.uniform(s["ls"][0], s["ls"][1], size=(1, K))
    kernel = rng.normal(0.0, s["wstd"], size=(K, O))
    bias = rng.normal(0.0, 0.1, size=(O,))
    return {"params": {
        "rbf_list": {"centers": centers.astype(dtype), "log_sigs": log_sigs.astype(dtype)},
        "linear": {"kernel": kernel.astype(dtype), "bias": bias.astype(dtype)}}}


def synth_queries(idx: int, B: int | None = None, dtype=np.float32, seed: int = 1123) -> np.ndarray:
    s = _SPECS[idx]
    B = s["B"] if B is None else B
    rng = np.random.default_rng(seed)
    return rng.uniform(np.asarray(s["lo"]), np.asarray(s["hi"]), size=(B, s["D"])).astype(dtype)


def synth_cotangent(idx: int, B: int | None = None, dtype=np.float32, seed: int = 2123) -> np.ndarray:
    s = _SPECS[idx]
    B = s["B"] if B is None else B
    return np.random.default_rng(seed).normal(0.0, 1.0, size=(B, s["O"])).astype(dtype)


def batch_size(idx: int) -> int:
    return _SPECS[idx]["B"]


def initial_state_from_query(x: np.ndarray) -> np.ndarray:
    """[0,0,0,v,0,angv,beta] from the 7-D query (scripts/train_nmpc.py:260-266)."""
    st = np.zeros((x.shape[0], 7), dtype=x.dtype)
    st[:, 3] = x[:, 0]
    st[:, 6] = x[:, 5]
    st[:, 5] = x[:, 6]
    return st
