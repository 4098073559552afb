"""Checkpoint / model-card I/O in the reference's on-disk formats.

* model card: the YAML the reference writes at scripts/train_nmpc.py:431-450 and reloads at
  src/irbfn_mpc/irbfn_planner.py:46-79 (``yaml.safe_load`` only);
* parameters: ``flax.training.checkpoints`` legacy msgpack files (scripts/train_nmpc.py:426,497-502):
  a msgpack map ``{step, params, opt_state}`` whose ndarrays are ext type 1 =
  ``msgpack((shape, dtype_name, raw_bytes))``.  Decoded with plain ``msgpack`` -- nothing is unpickled.

``restore_checkpoint`` returns the parameter pytree ``WCRBFNet.apply`` takes; ``save_checkpoint``
writes a file the reference's ``restore_checkpoint`` can read back (same container, float32 leaves).
"""
from __future__ import annotations

import os
import re
from typing import Optional, Tuple

import numpy as np


def load_model_card(path: str) -> dict:
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f)


def save_model_card(path: str, card: dict) -> None:
    import yaml
    with open(path, "w") as f:
        yaml.dump(card, f, default_flow_style=False)


def _ext_hook(code, data):
    import msgpack
    if code == 1:      # flax.serialization._ndarray_to_bytes
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape).copy()
    return msgpack.ExtType(code, data)


def _default(obj):
    import msgpack
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        return msgpack.ExtType(1, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    raise TypeError(f"cannot serialise {type(obj)}")


def load_flax_msgpack(path: str) -> dict:
    import msgpack
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    """Highest-step ``checkpoint_<n>`` in a run directory (flax natural ordering)."""
    best, best_step = None, -1
    for name in os.listdir(ckpt_dir):
        m = re.fullmatch(re.escape(prefix) + r"(\d+)", name)
        if m and int(m.group(1)) > best_step:
            best, best_step = os.path.join(ckpt_dir, name), int(m.group(1))
    return best


def restore_checkpoint(ckpt: str) -> Tuple[dict, int]:
    """ckpt: a checkpoint file or a run directory.  Returns (params pytree, step)."""
    path = ckpt
    if os.path.isdir(ckpt):
        path = latest_checkpoint(ckpt)
        if path is None:
            raise FileNotFoundError(f"no checkpoint_<n> file in {ckpt}")
    tree = load_flax_msgpack(path)
    p = tree["params"]["params"] if "params" in tree["params"] else tree["params"]
    if "rbf_list" not in p or "linear" not in p or "centers" not in p["rbf_list"]:
        raise ValueError(f"{path} does not hold a WCRBFNet / DeeperWCRBFNet parameter tree (found {sorted(p)}): "
                         "MLP / Cluster variants are out of scope")
    params = {"params": {
        "rbf_list": {"centers": np.asarray(p["rbf_list"]["centers"]), "log_sigs": np.asarray(p["rbf_list"]["log_sigs"])}}}
    # WCRBFNet holds `linear`; DeeperWCRBFNet adds `linear_pre1`, `linear_pre2` (model.py:254-256)
    for name in ("linear_pre1", "linear_pre2", "linear"):
        if name in p:
            params["params"][name] = {"kernel": np.asarray(p[name]["kernel"]), "bias": np.asarray(p[name]["bias"])}
    return params, int(tree.get("step", 0))


def save_checkpoint(ckpt_dir: str, params: dict, step: int, prefix: str = "checkpoint_") -> str:
    import msgpack
    p = params["params"] if "params" in params else params

    def host(a):
        return np.asarray(a.detach().cpu() if hasattr(a, "detach") else a)
    inner = {"rbf_list": {"centers": host(p["rbf_list"]["centers"]), "log_sigs": host(p["rbf_list"]["log_sigs"])}}
    for name in ("linear_pre1", "linear_pre2", "linear"):
        if name in p:
            inner[name] = {"kernel": host(p[name]["kernel"]), "bias": host(p[name]["bias"])}
    tree = {"step": int(step), "params": {"params": inner}, "opt_state": {}}
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    with open(path, "wb") as f:
        f.write(msgpack.packb(tree, default=_default, use_bin_type=True, strict_types=False))
    return path
