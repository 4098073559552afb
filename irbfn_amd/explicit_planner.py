"""Explicit-MPC table look-up: the baseline the learned network replaces
(src/irbfn_mpc/explicit_planner.py).  The table lives in HBM; both look-up forms of the reference are
batched kernels:

* ``grid_lookup``   -- ``ExplicitPlanner.plan``: per axis ``min(shape-1, searchsorted(keys, v, "right"))``
  then ``outputs[i0, ..., i6]`` (explicit_planner.py:165-175);
* ``nearest``       -- ``ExplicitFrenetPlanner.plan``: ``scipy.spatial.KDTree(inputs).query(lookup)``
  (explicit_planner.py:219, :383) as an exact brute-force scan of the float32 table.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import _ptr, _stream_ptr


class ExplicitTable:
    def __init__(self, inputs: np.ndarray, outputs: np.ndarray):
        """inputs [N,D], outputs [N,T,2] (the ``*_sorted.npz`` tables: rows in row-major grid order)."""
        torch = _lib.require_gpu()
        inputs = np.asarray(inputs)
        N, D = inputs.shape
        self.N, self.D = N, D
        flat = np.ascontiguousarray(np.asarray(outputs).reshape(N, -1), dtype=np.float32)
        self.OW = flat.shape[1]
        # explicit_planner.py:38-41: the unique values of every input column are the grid axes
        self.input_keys = [np.unique(inputs[:, d]) for d in range(D)]
        self.shape = [len(k) for k in self.input_keys]
        self.is_grid = int(np.prod(self.shape)) == N
        self._offsets = (C.c_int32 * (D + 1))(*np.concatenate(([0], np.cumsum(self.shape))).astype(np.int32))
        self._shape = (C.c_int32 * D)(*self.shape)
        self.keys = torch.from_numpy(np.concatenate(self.input_keys).astype(np.float64)).cuda()
        self.inputs = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).cuda()
        self.table = torch.from_numpy(flat).cuda()
        self._ws = None
        self._torch = torch

    def _x(self, lookup, dtype):
        torch = self._torch
        t = lookup if isinstance(lookup, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(lookup))
        t = t.to(device=self.table.device, dtype=dtype).contiguous()
        if t.dim() != 2 or t.shape[1] != self.D:
            raise ValueError(f"lookup must be [B, {self.D}]")
        return t

    def grid_lookup(self, lookup):
        """-> (flat row index [B] int64, outputs [B, OW]); needs a full grid table."""
        if not self.is_grid:
            raise ValueError("table rows do not form the full grid of their axis values")
        torch, lib = self._torch, _lib.load()
        x = self._x(lookup, torch.float64)
        B = x.shape[0]
        idx = torch.empty((B,), dtype=torch.int64, device=x.device)
        out = torch.empty((B, self.OW), dtype=torch.float32, device=x.device)
        st = lib.irbfn_lut_grid_lookup(_ptr(self.keys), self._offsets, self._shape, _ptr(self.table), _ptr(x), _ptr(idx),
                                       _ptr(out), B, self.D, self.OW, _stream_ptr(torch))
        _lib.check(st, "irbfn_lut_grid_lookup")
        return idx, out

    def nearest(self, lookup):
        """-> (row index [B] int64, distance [B], outputs [B, OW])."""
        torch, lib = self._torch, _lib.load()
        x = self._x(lookup, torch.float32)
        B = x.shape[0]
        need = lib.irbfn_lut_nearest_workspace_bytes(self.N, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=x.device)
        idx = torch.empty((B,), dtype=torch.int64, device=x.device)
        dist = torch.empty((B,), dtype=torch.float32, device=x.device)
        out = torch.empty((B, self.OW), dtype=torch.float32, device=x.device)
        st = lib.irbfn_lut_nearest(_ptr(self.inputs), _ptr(self.table), _ptr(x), _ptr(idx), _ptr(dist), _ptr(out), self.N,
                                   B, self.D, self.OW, _ptr(self._ws), need, _stream_ptr(torch))
        _lib.check(st, "irbfn_lut_nearest")
        return idx, dist, out
