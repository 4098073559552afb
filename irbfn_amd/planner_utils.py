"""Cubic-spiral path integrator with the reference's name (src/irbfn_mpc/planner_utils.py:62-77)."""
from __future__ import annotations

from . import _lib
from .dynamics import rollout_forward

N = 9   # planner_utils.py:8


def integrate_path_mult(params, n: int = N):
    """params [B, 5] = (k0, k1, k2, k3, s) -> all_states [B, n, 6] = [x, y, theta, kappa, dx, dy]."""
    if params.shape[1] != 5:
        raise ValueError("integrate_path_mult expects [B, 5] = (k0, k1, k2, k3, s)")
    return rollout_forward(_lib.ROLLOUT_SPIRAL, params, None, int(n))
