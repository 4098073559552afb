"""Checkpoint / model-card I/O in the reference's on-disk formats.

* model card: the YAML the reference writes at scripts/train_nmpc.py:431-450 and reloads at
  src/irbfn_mpc/irbfn_planner.py:46-79 (``yaml.safe_load`` only);
* parameters: ``flax.training.checkpoints`` legacy msgpack files (scripts/train_nmpc.py:426,497-502):
  a msgpack map ``{step, params, opt_state}`` whose ndarrays are ext type 1 =
  ``msgpack((shape, dtype_name, raw_bytes))``.  Decoded with plain ``msgpack`` -- nothing is unpickled.

``restore_checkpoint`` returns the parameter pytree ``WCRBFNet.apply`` takes.  ``save_checkpoint`` writes the
tree the reference's ``checkpoints.restore_checkpoint(ckpt_dir, target=state)`` expects (irbfn_planner.py:88,
eval_irbfn_dnmpc.py:54): ``step`` as a 0-d integer array and ``opt_state`` with the structure of
``optax.chain(clip_by_global_norm, adam)`` (scripts/train_nmpc.py:231-233) --
``{'0': {}, '1': {'0': {'count', 'mu': {'params': ...}, 'nu': {'params': ...}}, '1': {}}}`` -- flax restores a
tuple state from a dict with one entry per element and refuses any other length.  The structure is the one
decoded from the reference's own files (the older runs hold the bare-adam form ``{'0': {count, mu, nu}, '1': {}}``;
``restore_opt_state`` reads both).  flax / optax are not importable here: the round trip through the reference's
loader itself is **parity unpinned**; tests/test_checkpoint_cpu.py compares key structures with a reference file.
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
        a = np.asarray(obj, order="C")              # (ascontiguousarray would turn a 0-d array into shape (1,))
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
        raise ValueError(f"{path} does not hold a WCRBFNet / DeeperWCRBFNet / ClusterWCRBFNet parameter tree "
                         f"(found {sorted(p)}): the MLP baseline is out of scope")
    params = {"params": {
        "rbf_list": {"centers": np.asarray(p["rbf_list"]["centers"]), "log_sigs": np.asarray(p["rbf_list"]["log_sigs"])}}}
    # WCRBFNet holds `linear`; DeeperWCRBFNet adds `linear_pre1`, `linear_pre2` (model.py:254-256); ClusterWCRBFNet
    # adds the gate's Dense `cluster` (model.py:341-414)
    for name in _DENSE_GROUPS:
        if name in p:
            params["params"][name] = {"kernel": np.asarray(p[name]["kernel"]), "bias": np.asarray(p[name]["bias"])}
    return params, int(tree.get("step", 0))


_DENSE_GROUPS = ("linear_pre1", "linear_pre2", "linear", "cluster")


def _host_tree(params: dict) -> dict:
    p = params["params"] if "params" in params else params
    unknown = sorted(set(p) - {"rbf_list", *_DENSE_GROUPS})
    if unknown:          # never drop a parameter group silently: a checkpoint that cannot be resumed is worse than none
        raise ValueError(f"parameter groups {unknown} are not part of any supported model (rbf_list + {_DENSE_GROUPS})")

    def host(a):
        return np.asarray(a.detach().cpu() if hasattr(a, "detach") else a)
    inner = {"rbf_list": {"centers": host(p["rbf_list"]["centers"]), "log_sigs": host(p["rbf_list"]["log_sigs"])}}
    for name in _DENSE_GROUPS:
        if name in p:
            inner[name] = {"kernel": host(p[name]["kernel"]), "bias": host(p[name]["bias"])}
    return inner


def restore_opt_state(ckpt: str):
    """Adam moments of a checkpoint: (mu pytree, nu pytree, count) or None.  Reads the chain(clip, adam) form the
    reference writes today and the bare-adam form of its older runs."""
    path = latest_checkpoint(ckpt) if os.path.isdir(ckpt) else ckpt
    opt = load_flax_msgpack(path).get("opt_state") or {}
    for cand in (opt.get("1", {}).get("0") if isinstance(opt.get("1"), dict) else None, opt.get("0")):
        if isinstance(cand, dict) and {"count", "mu", "nu"} <= set(cand):
            return {"params": cand["mu"]["params"]}, {"params": cand["nu"]["params"]}, int(cand["count"])
    return None


def save_checkpoint(ckpt_dir: str, params: dict, step: int, prefix: str = "checkpoint_", opt_state=None) -> str:
    """opt_state: None (fresh optimiser: zero moments, count 0) or (mu pytree, nu pytree, count) -- e.g.
    ``train.TrainState.opt_state()``."""
    import msgpack
    inner = _host_tree(params)
    if opt_state is None:
        zeros = {g: {n: np.zeros_like(a) for n, a in d.items()} for g, d in inner.items()}
        mu, nu, count = zeros, {g: {n: a.copy() for n, a in d.items()} for g, d in zeros.items()}, 0
    else:
        mu, nu, count = _host_tree(opt_state[0]), _host_tree(opt_state[1]), int(opt_state[2])
    adam = {"count": np.asarray(count, np.int32), "mu": {"params": mu}, "nu": {"params": nu}}
    tree = {"step": np.asarray(int(step), np.int64), "params": {"params": inner},
            "opt_state": {"0": {}, "1": {"0": adam, "1": {}}}}
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{step}")
    with open(path, "wb") as f:
        f.write(msgpack.packb(tree, default=_default, use_bin_type=True, strict_types=False))
    return path
