"""ctypes wrapper of the C restatement (oracle/irbfn_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Used by tests/ (cross-check against the NumPy restatement) and by bench.py's ``cpu_baseline`` leg
(an OpenMP CPU timing of the same workload).  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# IRBFN_ORACLE_LIB: load another build of the same sources (the sanitizer build, `make -C oracle asan`)
LIB = os.environ.get("IRBFN_ORACLE_LIB") or os.path.join(HERE, "_build", "libirbfn_oracle.so")
BASIS_ENUM = {"gaussian": 0, "gaussian_wide": 1, "gaussian_wider": 2, "inverse_quadratic": 3, "linear": 4,
              "quadratic": 5, "multiquadric": 6, "inverse_multiquadric": 7, "spline": 8, "poisson_one": 9,
              "poisson_two": 10, "matern32": 11, "matern52": 12}
_lib = None


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("irbfn_oracle.c", "irbfn_oracle_impl.h", "Makefile")]
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(map(os.path.getmtime, src)):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return load().oracle_num_threads()


def set_num_threads(n: int):
    load().oracle_set_num_threads(C.c_int(n))


def gate_tables(cfg, dtype):
    ns = len(cfg["activation_idx"])
    mr = max([1] + [max(len(cfg["lower_bounds"][d]), len(cfg["upper_bounds"][d])) for d in range(ns)])
    lo = np.zeros((max(ns, 1), mr), dtype)
    hi = np.zeros((max(ns, 1), mr), dtype)
    for d in range(ns):
        lo[d, :len(cfg["lower_bounds"][d])] = cfg["lower_bounds"][d]
        hi[d, :len(cfg["upper_bounds"][d])] = cfg["upper_bounds"][d]
    nr = min(len(cfg["dimension_ranges"]), cfg["num_regions"])
    dr = np.ascontiguousarray(np.asarray([r[:ns] for r in cfg["dimension_ranges"][:nr]], np.int32).reshape(nr, ns))
    return ns, mr, lo, hi, np.asarray(cfg["delta"][:ns] if ns else [0.0], dtype), dr, nr


def wcrbf_forward(cfg, params, x, dtype=np.float32):
    lib = load()
    p = params["params"] if "params" in params else params
    c = np.ascontiguousarray(p["rbf_list"]["centers"], dtype)
    ls = np.ascontiguousarray(p["rbf_list"]["log_sigs"], dtype)
    W = np.ascontiguousarray(p["linear"]["kernel"], dtype)
    b = np.ascontiguousarray(p["linear"]["bias"], dtype)
    x = np.ascontiguousarray(x, dtype)
    ns, mr, lo, hi, delta, dr, nr = gate_tables(cfg, dtype)
    R, K, D = c.shape
    O = W.shape[1]
    out = np.empty((x.shape[0], O), dtype)
    fn = lib.oracle_wcrbf_forward_f32 if dtype == np.float32 else lib.oracle_wcrbf_forward_f64
    fn(_p(x), _p(c), _p(ls), _p(W), _p(b), _p(lo), _p(hi), _p(delta), _p(dr), C.c_int(nr), C.c_int(mr),
       C.c_int(ns), C.c_int(BASIS_ENUM[cfg["basis_func"]]), C.c_long(x.shape[0]), C.c_int(D), C.c_int(R),
       C.c_int(K), C.c_int(O), _p(out))
    return out


def wcrbf_vjp(cfg, params, x, gout, dtype=np.float64):
    """Parameter VJP for full-size batches (C twin of irbfn_oracle.wcrbfnet_vjp): -> gradient pytree."""
    lib = load()
    p = params["params"] if "params" in params else params
    c = np.ascontiguousarray(p["rbf_list"]["centers"], dtype)
    ls = np.ascontiguousarray(p["rbf_list"]["log_sigs"], dtype)
    W = np.ascontiguousarray(p["linear"]["kernel"], dtype)
    x = np.ascontiguousarray(x, dtype)
    g = np.ascontiguousarray(gout, dtype)
    ns, mr, lo, hi, delta, dr, nr = gate_tables(cfg, dtype)
    R, K, D = c.shape
    O = W.shape[1]
    gc, gl, gW, gb = np.empty_like(c), np.empty_like(ls), np.empty_like(W), np.empty((O,), dtype)
    fn = lib.oracle_wcrbf_vjp_f32 if dtype == np.float32 else lib.oracle_wcrbf_vjp_f64
    fn.restype = C.c_int
    rc = fn(_p(x), _p(g), _p(c), _p(ls), _p(W), _p(lo), _p(hi), _p(delta), _p(dr), C.c_int(nr), C.c_int(mr), C.c_int(ns),
            C.c_int(BASIS_ENUM[cfg["basis_func"]]), C.c_long(x.shape[0]), C.c_int(D), C.c_int(R), C.c_int(K), C.c_int(O),
            _p(gc), _p(gl), _p(gW), _p(gb))
    if rc != 0:
        raise ValueError(f"hand VJP not defined for basis {cfg['basis_func']}")
    return {"params": {"rbf_list": {"centers": gc, "log_sigs": gl}, "linear": {"kernel": gW, "bias": gb}}}


def _roll(name, dtype, xu, p, T, S, extra=()):
    lib = load()
    xu = np.ascontiguousarray(xu, dtype)
    out = np.empty((xu.shape[0], T, S), dtype)
    fn = getattr(lib, f"{name}_{'f32' if dtype == np.float32 else 'f64'}")
    args = [_p(xu)]
    if p is not None:
        pp = np.ascontiguousarray(p, dtype)
        args.append(_p(pp))
    args += [C.c_long(xu.shape[0]), C.c_int(T), *extra, _p(out)]
    fn(*args)
    return out


def integrate_st_mult(xu, p, T, dtype=np.float32, kinematic_only=False):
    return _roll("oracle_integrate_st_mult", dtype, xu, p, T, 7, (C.c_int(int(kinematic_only)),))


def integrate_frenet_mult(xu, p, T, dtype=np.float32):
    return _roll("oracle_integrate_frenet_mult", dtype, xu, p, T, 8)


def rollout_fullint(v0, u, T, dtype=np.float32):
    lib = load()
    v0 = np.ascontiguousarray(v0, dtype)
    u = np.ascontiguousarray(u, dtype)
    out = np.empty((u.shape[0], T, 5), dtype)
    fn = lib.oracle_rollout_fullint_f32 if dtype == np.float32 else lib.oracle_rollout_fullint_f64
    fn(_p(v0), _p(u), C.c_long(u.shape[0]), C.c_int(T), _p(out))
    return out


def integrate_path_mult(params, N=9, dtype=np.float32):
    return _roll("oracle_integrate_path_mult", dtype, params, None, N, 6)
