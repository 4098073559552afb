"""``WCRBFNet`` -- host-side mirror of the reference's Flax module (src/irbfn_mpc/model.py:98-198).

Same constructor fields as the reference / its YAML model card, same call surface
``WCRBFNet(**cfg).apply(params, x)`` with the same parameter pytree
``{"params": {"rbf_list": {"centers"[R,K,D], "log_sigs"[R,K]}, "linear": {"kernel"[K,O], "bias"[O]}}}``
(checkpoint layout).  The arithmetic runs in the HIP kernels behind ``libirbfn_hip.so``; torch is
used only for device memory and streams.  No CPU path: without the library or a GPU, calls raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional, Sequence

import numpy as np

from . import _lib
from .flax_rbf import basis_name

_CFG_FIELDS = ("in_features", "out_features", "num_kernels", "basis_func", "num_regions", "lower_bounds",
               "upper_bounds", "dimension_ranges", "activation_idx", "delta")


def _inner(params: dict) -> dict:
    return params["params"] if "params" in params else params


def _digest(arr: np.ndarray) -> bytes:
    """Content digest of a host array (xxh3 at memory speed when xxhash is there, blake2b otherwise)."""
    buf = np.ascontiguousarray(arr)
    try:
        import xxhash
        return xxhash.xxh3_128_digest(memoryview(buf).cast("B"))
    except ImportError:
        import hashlib
        return hashlib.blake2b(memoryview(buf).cast("B"), digest_size=16).digest()


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _stream_ptr(torch) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_device_f32(a, torch, device=None):
    """numpy / torch (any float dtype, any device) -> contiguous float32 cuda tensor (no copy if already so)."""
    if isinstance(a, torch.Tensor):
        t = a
    else:
        arr = np.ascontiguousarray(np.asarray(a))
        t = torch.from_numpy(arr) if arr.flags.writeable else torch.tensor(arr)    # read-only arrays: torch wants a copy
    if t.dtype != torch.float32:
        t = t.to(torch.float32)     # float64 checkpoints are cast on load
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    if t.device != dev:
        t = t.to(dev)
    return t.contiguous()


def like_input(out_t, ref, torch):
    """Give the result the flavour of the caller's input: numpy in -> numpy out, torch-cpu -> torch-cpu."""
    if isinstance(ref, torch.Tensor):
        return out_t if ref.is_cuda else out_t.cpu()
    return out_t.cpu().numpy()


class WCRBFNet:
    """Interpolating RBF network (smooth region gate, R vmapped RBF layers, Dense).

    Fields as in src/irbfn_mpc/model.py:112-125.  ``centers`` / ``fixed_centers`` / ``fixed_width``
    select upstream-only layer classes whose source is not in the reference snapshot
    (model.py:131-140); they all share the forward arithmetic of ``RBFLayer`` and differ in which
    parameters train, so they are accepted and ignored for ``apply``.
    """

    def __init__(self, in_features: int, out_features: int, num_kernels: int, basis_func: Any,
                 num_regions: int, lower_bounds: Sequence[Sequence[float]],
                 upper_bounds: Sequence[Sequence[float]], dimension_ranges: Sequence[Sequence[int]],
                 activation_idx: Sequence[int], delta: Sequence[float], centers=None,
                 fixed_centers: bool = False, fixed_width: bool = False, use_float64: bool = False, **_unused):
        self.in_features = int(in_features)
        self.out_features = int(out_features)
        self.num_kernels = int(num_kernels)
        self.basis_func = basis_name(basis_func)
        self.num_regions = int(num_regions)
        self.lower_bounds = [list(map(float, r)) for r in lower_bounds]
        self.upper_bounds = [list(map(float, r)) for r in upper_bounds]
        self.dimension_ranges = [list(map(int, r)) for r in dimension_ranges]
        self.activation_idx = list(activation_idx)
        self.delta = list(map(float, delta))
        self.fixed_centers, self.fixed_width = bool(fixed_centers), bool(fixed_width)
        # float64 mode = the reference's --use_float64 (scripts/train_nmpc.py:41-42): apply / vjp evaluate in float64 on the
        # GPU (irbfn_f64_*: plain f64 kernels, not the tuned float32 path) and return float64
        self.use_float64 = bool(use_float64)
        self._f64 = {}
        self._warned_f64 = False
        self.num_split_dimensions = len(self.activation_idx)          # model.py:128
        ns = self.num_split_dimensions
        if ns > self.in_features:
            raise ValueError("len(activation_idx) exceeds in_features")
        if len(self.lower_bounds) < ns or len(self.upper_bounds) < ns or len(self.delta) < ns:
            raise ValueError("lower_bounds / upper_bounds / delta must cover every split dimension")
        for i, r in enumerate(self.dimension_ranges):
            if len(r) < ns:
                raise ValueError(f"dimension_ranges[{i}] shorter than the number of split dimensions")
            for d in range(ns):
                if not (0 <= r[d] < len(self.lower_bounds[d]) and r[d] < len(self.upper_bounds[d])):
                    raise IndexError(f"dimension_ranges[{i}][{d}]={r[d]} indexes outside the bounds of dim {d}")
        self._handles: Dict[int, C.c_void_p] = {}
        self._bound_fp: Dict[int, tuple] = {}
        self._keepalive: Dict[int, tuple] = {}
        self._vjp_ws: Dict[int, Any] = {}

    # ------------------------------------------------------------------ construction helpers
    @classmethod
    def from_config(cls, cfg, use_float64: bool = False) -> "WCRBFNet":
        """cfg: dict, argparse.Namespace or path of a YAML model card (the file the reference writes at
        scripts/train_nmpc.py:431-450 and reloads at src/irbfn_mpc/irbfn_planner.py:46-79).  The card does not record
        --use_float64 (a process-wide jax flag in the reference): pass ``use_float64=True`` for that mode."""
        if isinstance(cfg, str):
            import yaml
            with open(cfg, "r") as f:
                cfg = yaml.safe_load(f)
        elif not isinstance(cfg, dict):
            cfg = vars(cfg)
        return cls(**{k: cfg[k] for k in _CFG_FIELDS}, use_float64=use_float64 or bool(cfg.get("use_float64", False)))

    def config(self) -> dict:
        return {k: getattr(self, k) for k in _CFG_FIELDS}

    def init(self, seed: int = 0, dtype=np.float32) -> dict:
        """Fresh parameter pytree with the reference initialisers: centers ~ N(0,1), log_sigs = 0
        (flax_rbf.py:246-256), Dense kernel lecun_normal, bias 0 (flax.linen.Dense defaults)."""
        rng = np.random.default_rng(seed)
        R, K, D, O = self.num_regions, self.num_kernels, self.in_features, self.out_features
        std = 1.0 / np.sqrt(K) / 0.87962566103423978   # truncated-normal correction of lecun_normal
        kern = np.clip(rng.normal(size=(K, O)), -2, 2) * std
        return {"params": {
            "rbf_list": {"centers": rng.normal(size=(R, K, D)).astype(dtype),
                         "log_sigs": np.zeros((R, K), dtype)},
            "linear": {"kernel": kern.astype(dtype), "bias": np.zeros((O,), dtype)}}}

    # ------------------------------------------------------------------ descriptor management
    def _gate_tables(self):
        ns = self.num_split_dimensions
        mr = max([1] + [max(len(self.lower_bounds[d]), len(self.upper_bounds[d])) for d in range(ns)])
        lo = np.zeros((max(ns, 1), mr), np.float32)
        hi = np.zeros((max(ns, 1), mr), np.float32)
        for d in range(ns):
            lo[d, :len(self.lower_bounds[d])] = self.lower_bounds[d]
            hi[d, :len(self.upper_bounds[d])] = self.upper_bounds[d]
        delta = np.asarray(self.delta[:ns] if ns else [0.0], np.float32)
        nr = min(len(self.dimension_ranges), self.num_regions)   # .at[:, i].set past R is dropped
        dr = np.asarray([r[:ns] for r in self.dimension_ranges[:nr]], np.int32).reshape(nr, ns)
        return ns, mr, lo, hi, delta, np.ascontiguousarray(dr), nr

    def _handle(self, torch) -> C.c_void_p:
        dev = torch.cuda.current_device()
        h = self._handles.get(dev)
        if h is None:
            lib = _lib.load()
            ns, mr, lo, hi, delta, dr, nr = self._gate_tables()
            h = C.c_void_p()
            st = lib.irbfn_net_create(
                C.byref(h), self.in_features, self.num_regions, self.num_kernels, self.out_features,
                _lib.BASIS_ENUM[self.basis_func], ns, mr, lo.ctypes.data_as(C.c_void_p),
                hi.ctypes.data_as(C.c_void_p), delta.ctypes.data_as(C.c_void_p),
                dr.ctypes.data_as(C.c_void_p), nr)
            _lib.check(st, "irbfn_net_create")
            self._handles[dev] = h
        return h

    def set_options(self, **opts) -> "WCRBFNet":
        """Per-descriptor kernel selection / launch geometry (``irbfn_net_set_option``; names = keys of
        ``_lib.OPTIONS``, e.g. ``fwd_kernel=_lib.FWD_K1``).  For A/B measurements and the reduced-precision
        report of BASELINE config 5; the defaults are the product path."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        h = self._handle(torch)
        for k, v in opts.items():
            _lib.check(lib.irbfn_net_set_option(h, _lib.OPTIONS[k], int(v)), f"irbfn_net_set_option({k}={v})")
        return self

    def __del__(self):
        try:
            lib = _lib.load()
            for h in self._handles.values():
                lib.irbfn_net_destroy(h)
        except Exception:
            pass
        self._handles = {}

    # ------------------------------------------------------------------ parameters
    def _check_shapes(self, p: dict):
        R, K, D, O = self.num_regions, self.num_kernels, self.in_features, self.out_features
        want = {"centers": (R, K, D), "log_sigs": (R, K), "kernel": (K, O), "bias": (O,)}
        got = {"centers": tuple(p["rbf_list"]["centers"].shape), "log_sigs": tuple(p["rbf_list"]["log_sigs"].shape),
               "kernel": tuple(p["linear"]["kernel"].shape), "bias": tuple(p["linear"]["bias"].shape)}
        for k in want:
            if want[k] != got[k]:
                raise ValueError(f"params {k} has shape {got[k]}, the model card implies {want[k]}")

    @staticmethod
    def _fingerprint(leaves, torch) -> tuple:
        """What ``bind`` remembers of the leaves it uploaded, to skip an identical upload.  A torch tensor is
        remembered as (the tensor OBJECT itself, data_ptr, _version): the strong reference keeps its id / storage from
        being recycled by a later, different tensor while it is cached, and any in-place write bumps ``_version``.  A
        NumPy array is remembered by CONTENT (a digest of its bytes, 0.3 MB at config 2), never by identity: ids and
        buffers of dropped temporaries are reused by CPython / the allocator -- exactly what
        ``net.apply(jax.tree.map(np.asarray, p), x)`` inside a ``pure_callback`` produces every step -- and a frozen
        array can be thawed, changed and frozen again."""
        fp = []
        for a in leaves:
            if isinstance(a, torch.Tensor):
                fp.append(("t", a, a.data_ptr(), a._version))
            else:
                arr = np.asarray(a)
                fp.append(("n", _digest(arr), arr.shape, str(arr.dtype)))
        return tuple(fp)

    @staticmethod
    def _same_fingerprint(old, new) -> bool:
        if old is None or len(old) != len(new):
            return False
        for o, n in zip(old, new):
            if o[0] != n[0]:
                return False
            if o[0] == "t":
                if o[1] is not n[1] or o[2:] != n[2:]:
                    return False
            elif o[1:] != n[1:]:
                return False
        return True

    def bind(self, params: dict) -> "WCRBFNet":
        """Uploads / re-packs the parameter pytree for the current device (irbfn_net_set_params)."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        self._check_shapes(p)
        leaves = [p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], p["linear"]["kernel"], p["linear"]["bias"]]
        dev = torch.cuda.current_device()
        fp = self._fingerprint(leaves, torch)
        if self._same_fingerprint(self._bound_fp.get(dev), fp):
            return self
        h = self._handle(torch)
        t = [to_device_f32(a, torch) for a in leaves]
        st = lib.irbfn_net_set_params(h, _ptr(t[0]), _ptr(t[1]), _ptr(t[2]), _ptr(t[3]), _stream_ptr(torch))
        _lib.check(st, "irbfn_net_set_params")
        self._keepalive[dev] = tuple(t)     # until the pack kernel has run
        self._bound_fp[dev] = fp
        return self

    # ------------------------------------------------------------------ forward
    def __call__(self, x):
        """Forward with the currently bound parameters: x[B,D] -> out[B,O]  (model.py:169-198)."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        if torch.cuda.current_device() not in self._bound_fp:
            raise ValueError("WCRBFNet: no parameters bound on this device; call apply(params, x) or bind(params)")
        shape = tuple(x.shape)
        if len(shape) != 2 or shape[1] != self.in_features:
            raise ValueError(f"x must have shape (B, {self.in_features}), got {shape}")
        xd = to_device_f32(x, torch)
        B = shape[0]
        out = torch.empty((B, self.out_features), dtype=torch.float32, device=xd.device)
        if B:
            st = lib.irbfn_net_forward(self._handle(torch), _ptr(xd), _ptr(out), B, _stream_ptr(torch))
            _lib.check(st, "irbfn_net_forward")
        return like_input(out, x, torch)

    def apply(self, params: dict, x):
        """Drop-in for ``WCRBFNet.apply(params, x)`` / ``state.apply_fn(state.params, x)``
        (src/irbfn_mpc/irbfn_planner.py:31).  ``use_float64=True``: evaluated in float64 (``apply64``)."""
        if self.use_float64:
            return self.apply64(params, x)
        self._warn_if_float64(params)
        self.bind(params)
        return self(x)

    # ------------------------------------------------------------------ float64 mode
    def _warn_if_float64(self, params: dict):
        """A --use_float64 checkpoint (float64 centers / log_sigs) handed to the float32 path: say so once."""
        if self._warned_f64:
            return
        p = _inner(params)
        dts = {str(getattr(p[g][n], "dtype", "")) for g, n in (("rbf_list", "centers"), ("rbf_list", "log_sigs"),
                                                                 ("linear", "kernel"), ("linear", "bias"))}
        if any("float64" in d for d in dts):
            import warnings
            warnings.warn("WCRBFNet: float64 parameter leaves are evaluated in float32 (within 1e-5 of a float64 run); construct "
                          "the net with use_float64=True for the float64 mode of the reference (--use_float64)", stacklevel=3)
            self._warned_f64 = True

    def _f64_card(self, torch):
        dev = torch.cuda.current_device()
        ent = self._f64.get(dev)
        if ent is None:
            ns = self.num_split_dimensions
            mr = max([1] + [max(len(self.lower_bounds[d]), len(self.upper_bounds[d])) for d in range(ns)])
            lo = np.zeros((max(ns, 1), mr), np.float64)
            hi = np.zeros((max(ns, 1), mr), np.float64)
            for d in range(ns):
                lo[d, :len(self.lower_bounds[d])] = self.lower_bounds[d]
                hi[d, :len(self.upper_bounds[d])] = self.upper_bounds[d]
            nr = min(len(self.dimension_ranges), self.num_regions)
            dr = np.asarray([r[:ns] for r in self.dimension_ranges[:nr]], np.int32).reshape(nr, max(ns, 0))
            tens = [torch.from_numpy(a).cuda() for a in (lo, hi, np.asarray(self.delta[:ns] if ns else [0.0], np.float64),
                                                         np.ascontiguousarray(dr if dr.size else np.zeros((1, 1), np.int32)))]
            card = _lib.F64Card(self.in_features, self.num_regions, self.num_kernels, self.out_features,
                                _lib.BASIS_ENUM[self.basis_func], ns, mr, nr, tens[0].data_ptr(), tens[1].data_ptr(),
                                tens[2].data_ptr(), tens[3].data_ptr())
            ent = (card, tens, {})
            self._f64[dev] = ent
        return ent

    @staticmethod
    def _dev_f64(a, torch):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
        return t.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float64).contiguous()

    def _f64_ws(self, torch, ent, nbytes):
        ws = ent[2].get("ws")
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty((max(int(nbytes), 8),), dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
            ent[2]["ws"] = ws
        return ws

    def apply64(self, params: dict, x):
        """``apply`` in float64 (the reference under --use_float64, scripts/train_nmpc.py:41-42) -> float64 [B,O]."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        self._check_shapes(p)
        ent = self._f64_card(torch)
        card = ent[0]
        c, l, k, b = (self._dev_f64(a, torch) for a in (p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], p["linear"]["kernel"],
                                                        p["linear"]["bias"]))
        xd = self._dev_f64(x, torch)
        if xd.dim() != 2 or xd.shape[1] != self.in_features:
            raise ValueError(f"x must have shape (B, {self.in_features}), got {tuple(xd.shape)}")
        B = xd.shape[0]
        out = torch.empty((B, self.out_features), dtype=torch.float64, device=xd.device)
        nbytes = int(lib.irbfn_f64_workspace_bytes(C.byref(card), B, 0))
        ws = self._f64_ws(torch, ent, nbytes)
        st = lib.irbfn_f64_forward(C.byref(card), _ptr(c), _ptr(l), _ptr(k), _ptr(b), _ptr(xd), _ptr(out), B, _ptr(ws), nbytes,
                                   _stream_ptr(torch))
        _lib.check(st, "irbfn_f64_forward")
        return like_input(out, x, torch)

    def vjp64(self, params: dict, x, gout) -> dict:
        """Parameter VJP in float64 (``jax.value_and_grad`` of the reference under --use_float64) -> float64 gradient pytree."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        self._check_shapes(p)
        ent = self._f64_card(torch)
        card = ent[0]
        c, l, k = (self._dev_f64(a, torch) for a in (p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], p["linear"]["kernel"]))
        xd, gd = self._dev_f64(x, torch), self._dev_f64(gout, torch)
        B = xd.shape[0]
        if tuple(gd.shape) != (B, self.out_features):
            raise ValueError(f"gout must have shape ({B}, {self.out_features}), got {tuple(gd.shape)}")
        R, K, D, O = self.num_regions, self.num_kernels, self.in_features, self.out_features
        gc = torch.empty((R, K, D), dtype=torch.float64, device=xd.device)
        gl = torch.empty((R, K), dtype=torch.float64, device=xd.device)
        gk = torch.empty((K, O), dtype=torch.float64, device=xd.device)
        gb = torch.empty((O,), dtype=torch.float64, device=xd.device)
        nbytes = int(lib.irbfn_f64_workspace_bytes(C.byref(card), B, 1))
        ws = self._f64_ws(torch, ent, nbytes)
        st = lib.irbfn_f64_vjp(C.byref(card), _ptr(c), _ptr(l), _ptr(k), _ptr(xd), _ptr(gd), _ptr(gc), _ptr(gl), _ptr(gk), _ptr(gb),
                               B, _ptr(ws), nbytes, _stream_ptr(torch))
        _lib.check(st, "irbfn_f64_vjp")
        conv = lambda t: like_input(t, x, torch)
        return {"params": {"rbf_list": {"centers": conv(gc), "log_sigs": conv(gl)}, "linear": {"kernel": conv(gk), "bias": conv(gb)}}}

    def gate(self, x):
        """``_region_activation`` (model.py:42-95): x[B,D] -> gamma[B,R]."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        xd = to_device_f32(x, torch)
        B = xd.shape[0]
        g = torch.empty((B, self.num_regions), dtype=torch.float32, device=xd.device)
        if B:
            _lib.check(lib.irbfn_net_gate(self._handle(torch), _ptr(xd), _ptr(g), B, _stream_ptr(torch)),
                       "irbfn_net_gate")
        return like_input(g, x, torch)

    # ------------------------------------------------------------------ backward
    def vjp(self, params: dict, x, gout, out: Optional[dict] = None) -> dict:
        """Parameter VJP: cotangent gout[B,O] -> gradient pytree (same structure as ``params``).
        Replaces ``jax.value_and_grad(loss_fn)(params)`` restricted to the network
        (scripts/train_nmpc.py:297-298).  Gradients w.r.t. x are never taken by the reference."""
        if self.use_float64 and out is None:
            return self.vjp64(params, x, gout)
        torch = _lib.require_gpu()
        lib = _lib.load()
        self._warn_if_float64(params)
        self.bind(params)
        xd, gd = to_device_f32(x, torch), to_device_f32(gout, torch)
        B = xd.shape[0]
        if tuple(gd.shape) != (B, self.out_features):
            raise ValueError(f"gout must have shape ({B}, {self.out_features}), got {tuple(gd.shape)}")
        R, K, D, O = self.num_regions, self.num_kernels, self.in_features, self.out_features
        dev = xd.device
        if out is not None:      # caller-provided gradient leaves (e.g. views of one flat buffer)
            o = _inner(out)
            gc, gl, gk, gb = o["rbf_list"]["centers"], o["rbf_list"]["log_sigs"], o["linear"]["kernel"], o["linear"]["bias"]
            for t, shp in ((gc, (R, K, D)), (gl, (R, K)), (gk, (K, O)), (gb, (O,))):
                if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                    raise ValueError("vjp out= leaves must be contiguous float32 cuda tensors of the parameter shapes")
        else:
            gc = torch.empty((R, K, D), dtype=torch.float32, device=dev)
            gl = torch.empty((R, K), dtype=torch.float32, device=dev)
            gk = torch.empty((K, O), dtype=torch.float32, device=dev)
            gb = torch.empty((O,), dtype=torch.float32, device=dev)
        h = self._handle(torch)
        nbytes = int(lib.irbfn_net_vjp_workspace_bytes(h, B))
        ws = self._vjp_ws.get(dev.index)
        if ws is None or ws.numel() < nbytes:        # grown on demand, reused across steps
            ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=dev)
            self._vjp_ws[dev.index] = ws
        st = lib.irbfn_net_vjp(h, _ptr(xd), _ptr(gd), _ptr(gc), _ptr(gl), _ptr(gk), _ptr(gb), B, _ptr(ws), nbytes,
                               _stream_ptr(torch))
        _lib.check(st, "irbfn_net_vjp")
        conv = (lambda t: t) if out is not None else (lambda t: like_input(t, x, torch))
        return {"params": {"rbf_list": {"centers": conv(gc), "log_sigs": conv(gl)},
                           "linear": {"kernel": conv(gk), "bias": conv(gb)}}}

    def last_launch(self) -> dict:
        torch = _lib.require_gpu()
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        g, b = C.c_int(0), C.c_int(0)
        lib.irbfn_net_last_launch(self._handle(torch), buf, 128, C.byref(g), C.byref(b))
        return {"kernel": buf.value.decode(), "grid": g.value, "block": b.value}


class DeeperWCRBFNet:
    """``DeeperWCRBFNet`` of the reference (src/irbfn_mpc/model.py:201-289): the RBF stage followed by
    Dense(64) -> relu -> Dense(64) -> relu -> Dense(out_features).  Parameter pytree (checkpoint layout):
    ``{"rbf_list": {centers, log_sigs}, "linear_pre1": {kernel[K,64], bias}, "linear_pre2":
    {kernel[64,64], bias}, "linear": {kernel[64,O], bias}}``.  The RBF stage + linear_pre1 run in the fused RBF
    kernel (64-wide Dense), the rest in ``irbfn_mlp_head_forward``; ``vjp`` = ``irbfn_mlp_head_vjp`` followed by the
    RBF-stage VJP seeded with the cotangent of linear_pre1's output."""

    HIDDEN = 64          # model.py:254-255

    def __init__(self, in_features, out_features, num_kernels, basis_func, num_regions, lower_bounds,
                 upper_bounds, dimension_ranges, activation_idx, delta, **_unused):
        self.out_features = int(out_features)
        self.in_features = int(in_features)
        self.stage = WCRBFNet(in_features=in_features, out_features=self.HIDDEN, num_kernels=num_kernels,
                              basis_func=basis_func, num_regions=num_regions, lower_bounds=lower_bounds,
                              upper_bounds=upper_bounds, dimension_ranges=dimension_ranges,
                              activation_idx=activation_idx, delta=delta)
        self._fp = None
        self._head = None

    @classmethod
    def from_config(cls, cfg) -> "DeeperWCRBFNet":
        if isinstance(cfg, str):
            import yaml
            with open(cfg, "r") as f:
                cfg = yaml.safe_load(f)
        elif not isinstance(cfg, dict):
            cfg = vars(cfg)
        return cls(**{k: cfg[k] for k in _CFG_FIELDS})

    def apply(self, params: dict, x):
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        H, O = self.HIDDEN, self.out_features
        for name, shp in (("linear_pre2", (H, H)), ("linear", (H, O))):
            if tuple(p[name]["kernel"].shape) != shp:
                raise ValueError(f"params {name}.kernel has shape {tuple(p[name]['kernel'].shape)}, expected {shp}")
        stage_params = {"rbf_list": p["rbf_list"], "linear": p["linear_pre1"]}
        xd = to_device_f32(x, torch)
        h1 = self.stage.apply(stage_params, xd)                     # linear_pre1(rbf_out)   model.py:283
        head = [to_device_f32(a, torch) for a in (p["linear_pre2"]["kernel"], p["linear_pre2"]["bias"],
                                                  p["linear"]["kernel"], p["linear"]["bias"])]
        B = xd.shape[0]
        out = torch.empty((B, O), dtype=torch.float32, device=xd.device)
        st = lib.irbfn_mlp_head_forward(_ptr(h1), _ptr(head[0]), _ptr(head[1]), _ptr(head[2]), _ptr(head[3]), _ptr(out),
                                        B, H, H, O, _stream_ptr(torch))
        _lib.check(st, "irbfn_mlp_head_forward")
        return like_input(out, x, torch)


    def _vjp_impl(self, params: dict, x, gout):
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        H, O = self.HIDDEN, self.out_features
        stage_params = {"rbf_list": p["rbf_list"], "linear": p["linear_pre1"]}
        xd, gd = to_device_f32(x, torch), to_device_f32(gout, torch)
        B = xd.shape[0]
        if tuple(gd.shape) != (B, O):
            raise ValueError(f"gout must have shape ({B}, {O})")
        h1 = self.stage.apply(stage_params, xd)
        w2, b2, w3 = (to_device_f32(a, torch) for a in (p["linear_pre2"]["kernel"], p["linear_pre2"]["bias"], p["linear"]["kernel"]))
        dev = xd.device
        gh1 = torch.empty((B, H), dtype=torch.float32, device=dev)
        gw2, gb2 = torch.empty((H, H), dtype=torch.float32, device=dev), torch.empty((H,), dtype=torch.float32, device=dev)
        gw3, gb3 = torch.empty((H, O), dtype=torch.float32, device=dev), torch.empty((O,), dtype=torch.float32, device=dev)
        nbytes = int(lib.irbfn_mlp_head_vjp_workspace_bytes(H, H, O))
        if nbytes < 0:
            _lib.check(nbytes, "irbfn_mlp_head_vjp_workspace_bytes")
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        st = lib.irbfn_mlp_head_vjp(_ptr(h1), _ptr(w2), _ptr(b2), _ptr(w3), _ptr(gd), _ptr(gh1), _ptr(gw2), _ptr(gb2),
                                    _ptr(gw3), _ptr(gb3), B, H, H, O, _ptr(ws), nbytes, _stream_ptr(torch))
        _lib.check(st, "irbfn_mlp_head_vjp")
        gs = self.stage.vjp(stage_params, xd, gh1)["params"]
        conv = lambda t: like_input(t, x, torch)
        return {"params": {"rbf_list": {k: conv(v) for k, v in gs["rbf_list"].items()},
                           "linear_pre1": {k: conv(v) for k, v in gs["linear"].items()},
                           "linear_pre2": {"kernel": conv(gw2), "bias": conv(gb2)},
                           "linear": {"kernel": conv(gw3), "bias": conv(gb3)}}}

    def vjp(self, params: dict, x, gout) -> dict:
        """Parameter VJP of the whole model: cotangent gout[B,O] -> gradient pytree with the structure of ``params``
        (what ``jax.value_and_grad`` returns for a DeeperWCRBFNet, scripts/train_nmpc_frenet.py:388-389,416-417)."""
        return self._vjp_impl(params, x, gout)


class ClusterWCRBFNet:
    """``ClusterWCRBFNet`` of the reference (src/irbfn_mpc/model.py:341-414): R RBF layers mixed by a learned
    softmax gate ``softmax(Dense(R)(x))`` instead of the tanh indicator.  Parameter pytree: ``{"rbf_list":
    {centers[R,K,D], log_sigs[R,K]}, "linear": {kernel[K,O], bias[O]}, "cluster": {kernel[D,R], bias[R]}}``.
    ``apply`` returns ``(out, logits)`` like the reference module; ``vjp`` is the parameter VJP of both outputs (what
    ``train_step_fullint_withcluster`` differentiates, scripts/train_nmpc_frenet.py:424-453).  No trained checkpoint
    of this variant survives in the reference (.MISSING_LARGE_BLOBS) -> parity against the oracle restatement only."""

    def __init__(self, in_features, out_features, num_kernels, basis_func, num_regions, **_unused):
        self.in_features, self.out_features = int(in_features), int(out_features)
        self.num_regions = int(num_regions)
        self._gate_ws = {}
        # descriptor with R regions and no gate tables: the region weights come from the softmax gate
        self.stage = WCRBFNet(in_features=in_features, out_features=out_features, num_kernels=num_kernels,
                              basis_func=basis_func, num_regions=num_regions, lower_bounds=[], upper_bounds=[],
                              dimension_ranges=[], activation_idx=[], delta=[])

    def apply(self, params: dict, x):
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        D, R, O = self.in_features, self.num_regions, self.out_features
        if tuple(p["cluster"]["kernel"].shape) != (D, R) or tuple(p["cluster"]["bias"].shape) != (R,):
            raise ValueError(f"params cluster.kernel / bias must be [{D},{R}] / [{R}]")
        self.stage.bind({"rbf_list": p["rbf_list"], "linear": p["linear"]})
        xd = to_device_f32(x, torch)
        B = xd.shape[0]
        if tuple(xd.shape) != (B, D):
            raise ValueError(f"x must be [B, {D}]")
        wc, bc = to_device_f32(p["cluster"]["kernel"], torch), to_device_f32(p["cluster"]["bias"], torch)
        logits = torch.empty((B, R), dtype=torch.float32, device=xd.device)
        gamma = torch.empty((B, R), dtype=torch.float32, device=xd.device)
        out = torch.empty((B, O), dtype=torch.float32, device=xd.device)
        st = lib.irbfn_cluster_gate(_ptr(xd), _ptr(wc), _ptr(bc), _ptr(logits), _ptr(gamma), B, D, R, _stream_ptr(torch))
        _lib.check(st, "irbfn_cluster_gate")
        st = lib.irbfn_net_forward_gamma(self.stage._handle(torch), _ptr(xd), _ptr(gamma), _ptr(out), B, _stream_ptr(torch))
        _lib.check(st, "irbfn_net_forward_gamma")
        return like_input(out, x, torch), like_input(logits, x, torch)

    def vjp(self, params: dict, x, gout, glogits=None, out: Optional[dict] = None) -> dict:
        """Parameter VJP of ``apply``: cotangents gout[B,O] of ``out`` and (optionally) glogits[B,R] of ``logits`` ->
        gradient pytree with the structure of ``params`` (rbf_list / linear / cluster).  The gate is recomputed (one
        [B,D]x[D,R] pass), K2 runs with the softmax weights, the cotangent of those weights comes from
        ``irbfn_net_vjp_gamma`` and goes back through softmax + Dense in ``irbfn_cluster_gate_vjp``."""
        torch = _lib.require_gpu()
        lib = _lib.load()
        p = _inner(params)
        D, R, O, K = self.in_features, self.num_regions, self.out_features, self.stage.num_kernels
        if O > 16:
            raise ValueError("ClusterWCRBFNet.vjp supports out_features <= 16 (the reference trains it with 10)")
        self.stage.bind({"rbf_list": p["rbf_list"], "linear": p["linear"]})
        xd, gd = to_device_f32(x, torch), to_device_f32(gout, torch)
        B = xd.shape[0]
        if tuple(xd.shape) != (B, D) or tuple(gd.shape) != (B, O):
            raise ValueError(f"x must be [B, {D}] and gout [B, {O}]")
        gl_in = None
        if glogits is not None:
            gl_in = to_device_f32(glogits, torch)
            if tuple(gl_in.shape) != (B, R):
                raise ValueError(f"glogits must be [B, {R}]")
        dev = xd.device
        shapes = {"centers": (R, K, D), "log_sigs": (R, K), "kernel": (K, O), "bias": (O,), "ckernel": (D, R), "cbias": (R,)}
        if out is not None:
            o = _inner(out)
            leaves = {"centers": o["rbf_list"]["centers"], "log_sigs": o["rbf_list"]["log_sigs"], "kernel": o["linear"]["kernel"],
                      "bias": o["linear"]["bias"], "ckernel": o["cluster"]["kernel"], "cbias": o["cluster"]["bias"]}
            for n, t in leaves.items():
                if tuple(t.shape) != shapes[n] or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                    raise ValueError("vjp out= leaves must be contiguous float32 cuda tensors of the parameter shapes")
        else:
            leaves = {n: torch.empty(shp, dtype=torch.float32, device=dev) for n, shp in shapes.items()}
        wc, bc = to_device_f32(p["cluster"]["kernel"], torch), to_device_f32(p["cluster"]["bias"], torch)
        logits = torch.empty((B, R), dtype=torch.float32, device=dev)
        gamma = torch.empty((B, R), dtype=torch.float32, device=dev)
        dgamma = torch.empty((B, R), dtype=torch.float32, device=dev)
        stream = _stream_ptr(torch)
        _lib.check(lib.irbfn_cluster_gate(_ptr(xd), _ptr(wc), _ptr(bc), _ptr(logits), _ptr(gamma), B, D, R, stream),
                   "irbfn_cluster_gate")
        h = self.stage._handle(torch)
        nbytes = int(lib.irbfn_net_vjp_workspace_bytes(h, B))
        ws = self.stage._vjp_ws.get(dev.index)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty((max(nbytes, 4),), dtype=torch.uint8, device=dev)
            self.stage._vjp_ws[dev.index] = ws
        st = lib.irbfn_net_vjp_gamma(h, _ptr(xd), _ptr(gamma), _ptr(gd), _ptr(leaves["centers"]), _ptr(leaves["log_sigs"]),
                                     _ptr(leaves["kernel"]), _ptr(leaves["bias"]), _ptr(dgamma), B, _ptr(ws), nbytes, stream)
        _lib.check(st, "irbfn_net_vjp_gamma")
        gbytes = int(lib.irbfn_cluster_gate_vjp_workspace_bytes(D, R))
        gws = self._gate_ws.get(dev.index)
        if gws is None or gws.numel() < gbytes:
            gws = torch.empty((max(gbytes, 4),), dtype=torch.uint8, device=dev)
            self._gate_ws[dev.index] = gws
        st = lib.irbfn_cluster_gate_vjp(_ptr(xd), _ptr(gamma), _ptr(dgamma), _ptr(gl_in) if gl_in is not None else None,
                                        _ptr(logits), _ptr(leaves["ckernel"]), _ptr(leaves["cbias"]), B, D, R, _ptr(gws), gbytes,
                                        stream)
        _lib.check(st, "irbfn_cluster_gate_vjp")
        conv = (lambda t: t) if out is not None else (lambda t: like_input(t, x, torch))
        return {"params": {"rbf_list": {"centers": conv(leaves["centers"]), "log_sigs": conv(leaves["log_sigs"])},
                           "linear": {"kernel": conv(leaves["kernel"]), "bias": conv(leaves["bias"])},
                           "cluster": {"kernel": conv(leaves["ckernel"]), "bias": conv(leaves["cbias"])}}}


class _State:
    """Minimal stand-in of flax's TrainState for ``pred_step``: ``apply_fn`` + ``params``."""

    def __init__(self, apply_fn, params):
        self.apply_fn, self.params = apply_fn, params


def make_state(net: WCRBFNet, params: dict) -> _State:
    return _State(net.apply, params)


def pred_step(state, x):
    """``pred_step(state, x)`` of src/irbfn_mpc/irbfn_planner.py:29-32."""
    return state.apply_fn(state.params, x)
