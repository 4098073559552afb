"""NMPC look-up tables: the ``.npz`` data format either side of the hot path and
the host-side preparation the reference trainers do before the first step.

Format (scripts/train_nmpc.py:48-58, scripts/train_nmpc_frenet.py:48-66): ``inputs [N, D]``,
``outputs [N, T, 2]`` = (accel, steer-velocity) per horizon step, optional ``constraints``; a row whose
NMPC solve failed holds ``-999`` in ``outputs``.

    Cartesian columns  [v_c, x_g, y_g, t_g, v_g, beta, angvz]                 (train_nmpc.py:50-56)
    Frenet columns     [ey, delta, vx_car, vy_car, vx_goal, wz, epsi, curv]   (train_nmpc_frenet.py:58-66)

Everything here is one-time NumPy work (as in the reference); the per-epoch batch gather runs on the
device (``DeviceTable``), replacing the host fancy-indexing of train_nmpc.py:466-467.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

INFEASIBLE = -999
CARTESIAN, FRENET = "cartesian", "frenet"
# columns whose sign flips under the left/right mirror symmetry
_MIRROR_COLS = {CARTESIAN: (2, 3),       # y_g, t_g          train_nmpc.py:66-67
                FRENET: (0, 6)}          # ey, epsi          train_nmpc_frenet.py:91,97
# `delta` of the smooth region gate the trainers write into the model card
GATE_DELTA = {CARTESIAN: [100.0] * 7,                                          # train_nmpc.py:219
              FRENET: [15.0, 10.0, 100.0, 100.0, 100.0, 10.0, 10.0, 10.0]}     # train_nmpc_frenet.py:246


def remove_infeasible(inputs: np.ndarray, outputs: np.ndarray, *extra: np.ndarray):
    """train_nmpc_frenet.py:51-56 -- keep the rows that hold at least one entry different from -999
    (``np.unique(np.where(outputs != -999)[0])``: a row is dropped only if EVERY entry is -999)."""
    valid = np.unique(np.where(outputs != INFEASIBLE)[0])
    return (inputs[valid], outputs[valid]) + tuple(e[valid] for e in extra) + (valid,)


def load_table(npz_path: str, kind: str = CARTESIAN, drop_infeasible: Optional[bool] = None):
    """-> (inputs [N,D], outputs [N,T,2]).  The Frenet trainer filters infeasible rows
    (train_nmpc_frenet.py:51-54), the Cartesian one does not (train_nmpc.py:48-49)."""
    if kind not in _MIRROR_COLS:
        raise ValueError(f"kind must be '{CARTESIAN}' or '{FRENET}'")
    with np.load(npz_path) as data:           # allow_pickle stays False
        inputs, outputs = data["inputs"], data["outputs"]
    if outputs.ndim != 3 or outputs.shape[2] != 2 or inputs.ndim != 2 or inputs.shape[0] != outputs.shape[0]:
        raise ValueError(f"{npz_path}: expected inputs [N,D] and outputs [N,T,2], got {inputs.shape} / {outputs.shape}")
    if drop_infeasible is None:
        drop_infeasible = kind == FRENET
    if drop_infeasible:
        inputs, outputs, _ = remove_infeasible(inputs, outputs)
    return inputs, outputs


def mirror(inputs: np.ndarray, outputs: np.ndarray, kind: str = CARTESIAN):
    """``--mirror_data``: append the left/right mirrored copy of every row (train_nmpc.py:61-72,
    train_nmpc_frenet.py:89-101): mirrored input columns and the steer-velocity outputs change sign."""
    m_in = inputs.copy()
    for c in _MIRROR_COLS[kind]:
        m_in[:, c] = -m_in[:, c]
    m_out = outputs.copy()
    m_out[:, :, 1] = -m_out[:, :, 1]
    return np.concatenate((inputs, m_in), axis=0), np.concatenate((outputs, m_out), axis=0)


def flatten_outputs(outputs: np.ndarray, only_onestep: bool = False) -> np.ndarray:
    """[N,T,2] -> [N,2T] = [a_0..a_{T-1}, sv_0..sv_{T-1}] (train_nmpc.py:172); ``only_onestep`` keeps
    columns [0, 5] -- hard-coded for T = 5 in the reference (train_nmpc.py:173-174)."""
    flat = np.hstack([outputs[:, :, 0], outputs[:, :, 1]])
    if only_onestep:
        flat = flat[:, [0, 5]]
    return flat


def generate_bounds(inputs: np.ndarray, num_splits: Sequence[int]):
    """Region bounds from the table's own grid values (train_nmpc.py:93-161,
    train_nmpc_frenet.py:116-197): per column the sorted unique values, cut into ``num_splits[d]``
    index-equidistant ranges (``np.linspace(..., dtype=int)`` truncates); ``dimension_ranges`` is the
    'ij' mesh of range indices (train_nmpc.py:178-184).  -> (lower_bounds, upper_bounds,
    dimension_ranges, num_regions)."""
    if len(num_splits) != inputs.shape[1]:
        raise ValueError("one split count per input column")
    lower: List[list] = []
    upper: List[list] = []
    for d, n in enumerate(num_splits):
        vals = np.sort(np.unique(inputs[:, d]))
        ind = np.linspace(start=0, stop=len(vals) - 1, num=int(n) + 1, endpoint=True, dtype=int)
        lower.append(list(vals[ind[:-1]]))
        upper.append(list(vals[ind[1:]]))
    ranges = [np.arange(len(b)) for b in lower]
    dimension_ranges = np.stack(np.meshgrid(*ranges, indexing="ij"), axis=-1).reshape(-1, len(ranges)).tolist()
    return lower, upper, dimension_ranges, int(np.prod([int(n) for n in num_splits]))


def model_card(inputs: np.ndarray, flat_outputs: np.ndarray, num_splits: Sequence[int], num_kernels: int,
               basis_func: str, kind: str = CARTESIAN) -> dict:
    """The WCRBFNet constructor fields the trainers derive from a table (train_nmpc.py:188-238,
    440-450): what ``WCRBFNet.from_config`` / ``checkpoint.save_model_card`` take."""
    lower, upper, dimension_ranges, num_regions = generate_bounds(inputs, num_splits)
    D = inputs.shape[1]

    def plain(v):
        return [[float(x) for x in row] for row in v]
    return {"in_features": int(D), "out_features": int(flat_outputs.shape[1]), "num_kernels": int(num_kernels),
            "basis_func": basis_func, "num_regions": num_regions, "lower_bounds": plain(lower),
            "upper_bounds": plain(upper), "dimension_ranges": dimension_ranges,
            "activation_idx": list(range(D)), "delta": list(GATE_DELTA[kind][:D])}


def prepare(npz_path: str, kind: str = CARTESIAN, mirror_data: bool = False, only_onestep: bool = False):
    """load -> (filter) -> (mirror) -> flatten: (flattened_input [N',D], flattened_output [N',O])."""
    inputs, outputs = load_table(npz_path, kind)
    if mirror_data:
        inputs, outputs = mirror(inputs, outputs, kind)
    return inputs, flatten_outputs(outputs, only_onestep)


class DeviceTable:
    """Training table resident in HBM; an epoch is a device-side permutation + row gather
    (train_nmpc.py:456-467 does both on the host).  288 GB of HBM holds any table the reference names
    (the largest, 8x19x19x64x8x6x12 rows x 17 floats, is 7 GB)."""

    def __init__(self, flat_inputs, flat_outputs, device=None, seed: int = 0):
        from . import _lib
        torch = _lib.require_gpu()
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.x = torch.as_tensor(np.ascontiguousarray(flat_inputs, dtype=np.float32)).to(dev)
        self.y = torch.as_tensor(np.ascontiguousarray(flat_outputs, dtype=np.float32)).to(dev)
        if self.x.shape[0] != self.y.shape[0]:
            raise ValueError("inputs and outputs differ in row count")
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self._torch = torch

    def __len__(self):
        return self.x.shape[0]

    def epoch(self, batch_size: int):
        """Yields (batch_x, batch_y) for ``len // batch_size`` full batches of a fresh permutation (the
        remainder is dropped, train_nmpc.py:459-464).  The permutation stream is torch's, not
        jax.random's: batch COMPOSITION is not comparable with the reference, only its distribution."""
        torch = self._torch
        n = (len(self) // batch_size) * batch_size
        perm = torch.randperm(len(self), device=self.x.device, generator=self.gen)[:n].view(-1, batch_size)
        for rows in perm:
            yield self.x.index_select(0, rows), self.y.index_select(0, rows)
