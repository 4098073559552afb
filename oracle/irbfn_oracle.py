"""CPU oracle for the IRBFN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file restates, line by line, the arithmetic of the reference's hot path
(hzheng40/irbfn @ 2024_10_08) in NumPy (and, through the same code, in
torch-CPU so that ``torch.autograd`` can stand in for ``jax.grad``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product path (``irbfn_amd``) never does.

Pinning status
--------------
* Roll-out B (``dynamic_st_onestep`` / ``integrate_st_mult``): PINNED by KAT-1
  (``scripts/test_dynamics.ipynb`` cell 4, float32, bit-exact) and by the
  CommonRoad derivative vectors KAT-2/KAT-3
  (``deprecated/f1tenth_gym/tests/test_dynamics.py:55-96``).
* RBF stage (``flax_rbf``): the arithmetic lives in the un-pinned third-party
  dependency ``flax_rbf @ git+https://github.com/hzheng40/flax_rbf``
  (``pyproject.toml:18``); the only readable source is the older vendored
  snapshot ``deprecated/f1tenth_gym/examples/flax_rbf/flax_rbf/flax_rbf.py``.
  The reference holds no test / recorded output of any ``RBFLayer`` or
  ``WCRBFNet`` call, and JAX/Flax are not importable in the build container
  (ordinary ``ModuleNotFoundError``), so for the RBF stage, the gate, the
  Frenet roll-out, the inline bicycle and the spiral integrator:
  **parity unpinned** -- parity is by source restatement only.

All citations are ``path:line`` relative to the reference root.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Sequence

import numpy as np

G_ACC = 9.81  # src/irbfn_mpc/dynamics.py:6


# --------------------------------------------------------------------------
# tiny array-namespace shim: the same restatement runs on numpy and on torch
# --------------------------------------------------------------------------
class _NP:
    name = "numpy"
    sin, cos, tan, tanh, exp, sqrt, log = np.sin, np.cos, np.tan, np.tanh, np.exp, np.sqrt, np.log

    @staticmethod
    def clip(a, lo, hi):
        return np.clip(a, lo, hi)

    @staticmethod
    def where(c, a, b):
        return np.where(c, a, b)

    @staticmethod
    def stack(xs, axis):
        return np.stack(xs, axis=axis)

    @staticmethod
    def zeros_like(a):
        return np.zeros_like(a)

    @staticmethod
    def ones_like(a):
        return np.ones_like(a)

    @staticmethod
    def const(v, like):
        return np.asarray(v, dtype=like.dtype)

    @staticmethod
    def sum(a, axis):
        return a.sum(axis=axis)


def _torch_ns():
    import torch

    class _TH:
        name = "torch"
        sin, cos, tan, tanh, exp, sqrt, log = (
            torch.sin, torch.cos, torch.tan, torch.tanh, torch.exp, torch.sqrt, torch.log)

        @staticmethod
        def clip(a, lo, hi):
            return torch.clip(a, min=lo, max=hi)

        @staticmethod
        def where(c, a, b):
            return torch.where(c, a, b)

        @staticmethod
        def stack(xs, axis):
            return torch.stack(xs, dim=axis)

        @staticmethod
        def zeros_like(a):
            return torch.zeros_like(a)

        @staticmethod
        def ones_like(a):
            return torch.ones_like(a)

        @staticmethod
        def const(v, like):
            return torch.as_tensor(v, dtype=like.dtype)

        @staticmethod
        def sum(a, axis):
            return a.sum(dim=axis)

    return _TH


def _ns(a):
    return _NP if isinstance(a, np.ndarray) else _torch_ns()


# --------------------------------------------------------------------------
# a-2  basis functions -- flax_rbf.py:34-111 (vendored snapshot)
# --------------------------------------------------------------------------
def _b_gaussian(a, xp):             # flax_rbf.py:35-37
    return xp.exp(-1 * a ** 2)


def _b_gaussian_wide(a, xp):        # flax_rbf.py:40-42
    return xp.exp(-0.1 * a ** 2)


def _b_gaussian_wider(a, xp):       # flax_rbf.py:45-47
    return xp.exp(-0.01 * a ** 2)


def _b_inverse_quadratic(a, xp):    # flax_rbf.py:50-52
    return xp.ones_like(a) / (xp.ones_like(a) + a ** 2)


def _b_linear(a, xp):               # flax_rbf.py:55-57
    return a


def _b_quadratic(a, xp):            # flax_rbf.py:61-63
    return a ** 2


def _b_multiquadric(a, xp):         # flax_rbf.py:67-69
    return (xp.ones_like(a) + a ** 2) ** 0.5


def _b_inverse_multiquadric(a, xp):  # flax_rbf.py:73-75
    return xp.ones_like(a) / (xp.ones_like(a) + a ** 2) ** 0.5


def _b_spline(a, xp):               # flax_rbf.py:79-81
    return a ** 2 * xp.log(a + xp.ones_like(a))


def _b_poisson_one(a, xp):          # flax_rbf.py:85-87
    return (a - xp.ones_like(a)) * xp.exp(-a)


def _b_poisson_two(a, xp):          # flax_rbf.py:91-97
    return ((a - 2 * xp.ones_like(a)) / 2 * xp.ones_like(a)) * a * xp.exp(-a)


def _b_matern32(a, xp):             # flax_rbf.py:101-103
    return (xp.ones_like(a) + 3 ** 0.5 * a) * xp.exp(-(3 ** 0.5) * a)


def _b_matern52(a, xp):             # flax_rbf.py:107-111
    return (xp.ones_like(a) + 5 ** 0.5 * a + (5 / 3) * a ** 2) * xp.exp(-(5 ** 0.5) * a)


BASIS: Dict[str, Callable] = {
    "gaussian": _b_gaussian,
    "gaussian_wide": _b_gaussian_wide,
    "gaussian_wider": _b_gaussian_wider,
    "inverse_quadratic": _b_inverse_quadratic,
    "linear": _b_linear,
    "quadratic": _b_quadratic,
    "multiquadric": _b_multiquadric,
    "inverse_multiquadric": _b_inverse_multiquadric,
    "spline": _b_spline,
    "poisson_one": _b_poisson_one,
    "poisson_two": _b_poisson_two,
    "matern32": _b_matern32,
    "matern52": _b_matern52,
}


def basis(name: str, alpha):
    return BASIS[name](alpha, _ns(alpha))


# --------------------------------------------------------------------------
# a-3  smooth region gate -- src/irbfn_mpc/model.py:42-95
# --------------------------------------------------------------------------
def region_activation(x, num_regions: int, num_split_dimensions: int,
                      lower_bounds, upper_bounds, delta, dimension_ranges):
    """gamma[B, R].  Indexes x[:, d] for d = 0..nsplit-1 (model.py:74-81), regions
    beyond len(dimension_ranges) stay 0 (model.py:70, 88-93)."""
    xp = _ns(x)
    all_gammas = []
    for d in range(num_split_dimensions):           # model.py:74
        lo = xp.const(np.asarray(lower_bounds[d], dtype=np.float64), x)
        hi = xp.const(np.asarray(upper_bounds[d], dtype=np.float64), x)
        lower_diffs = x[:, d][:, None] - lo[None, :]      # model.py:75-77
        upper_diffs = hi[None, :] - x[:, d][:, None]      # model.py:78-81
        dl = xp.const(delta[d], x)
        gamma = ((xp.tanh(dl * lower_diffs) + 1) / 2) * ((xp.tanh(dl * upper_diffs) + 1) / 2)  # :83-85
        all_gammas.append(gamma)
    cols = []
    for i in range(num_regions):
        if i < len(dimension_ranges):               # model.py:88-93
            cur = all_gammas[0][:, dimension_ranges[i][0]]
            for j in range(1, num_split_dimensions):
                cur = cur * all_gammas[j][:, dimension_ranges[i][j]]
        else:                                       # model.py:70 (zeros)
            cur = xp.zeros_like(x[:, 0])
        cols.append(cur)
    return xp.stack(cols, 1)


# --------------------------------------------------------------------------
# a-1  RBF layer -- flax_rbf.py:258-285, vmapped by model.py:143-159
# --------------------------------------------------------------------------
def rbf_layer(x, centers, log_sigs, basis_name: str):
    """x[B,D], centers[R,K,D], log_sigs[R,K] -> phi[B,R,K].
    d = sqrt(sum((x - c)^2)) / exp(log_sig)   (flax_rbf.py:280) ; phi = basis(d) (:283)."""
    xp = _ns(x)
    diff = x[:, None, None, :] - centers[None, :, :, :]          # flax_rbf.py:275-280
    d = (xp.sum(diff ** 2, -1) ** 0.5) / xp.exp(log_sigs)[None, :, :]
    return BASIS[basis_name](d, xp)


# --------------------------------------------------------------------------
# a-4  WCRBFNet.__call__ -- src/irbfn_mpc/model.py:169-198
# --------------------------------------------------------------------------
def wcrbfnet_apply(cfg: dict, params: dict, x, return_aux: bool = False, chunk: int = 0):
    """cfg: the YAML model card (in_features, out_features, num_kernels, basis_func,
    num_regions, lower_bounds, upper_bounds, dimension_ranges, activation_idx, delta).
    params: {"params": {"rbf_list": {"centers","log_sigs"}, "linear": {"kernel","bias"}}}.
    """
    p = params["params"] if "params" in params else params
    centers, log_sigs = p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"]
    kernel, bias = p["linear"]["kernel"], p["linear"]["bias"]
    xp = _ns(x)
    nsplit = len(cfg["activation_idx"])              # model.py:128

    def _one(xc):
        gamma = region_activation(xc, cfg["num_regions"], nsplit, cfg["lower_bounds"],
                                  cfg["upper_bounds"], cfg["delta"], cfg["dimension_ranges"])  # :178-186
        all_x = rbf_layer(xc, centers, log_sigs, cfg["basis_func"])       # model.py:190
        rbf_out = xp.sum(gamma[:, :, None] * all_x, 1)                    # model.py:187,193
        out = rbf_out @ kernel + bias                                     # model.py:196
        return out, rbf_out, gamma

    if chunk and x.shape[0] > chunk and xp is _NP:
        outs = [_one(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)]
        out, h, gam = (np.concatenate([o[j] for o in outs], 0) for j in range(3))
    else:
        out, h, gam = _one(x)
    return (out, h, gam) if return_aux else out


def deeper_wcrbfnet_apply(cfg: dict, params: dict, x):
    """DeeperWCRBFNet.__call__ -- src/irbfn_mpc/model.py:259-289 (Dense(64), Dense(64), Dense(O) with
    relu between, model.py:254-256, 283-287)."""
    p = params["params"] if "params" in params else params
    xp = _ns(x)
    stage = {"rbf_list": p["rbf_list"], "linear": p["linear_pre1"]}
    stage_cfg = dict(cfg, out_features=p["linear_pre1"]["kernel"].shape[1])
    out_pre1 = wcrbfnet_apply(stage_cfg, stage, x)                               # :283
    relu = (lambda t: np.maximum(t, 0)) if xp is _NP else (lambda t: t.clamp(min=0))
    out_pre2 = relu(out_pre1) @ p["linear_pre2"]["kernel"] + p["linear_pre2"]["bias"]   # :284
    return relu(out_pre2) @ p["linear"]["kernel"] + p["linear"]["bias"]                 # :285


def cluster_wcrbfnet_apply(cfg: dict, params: dict, x):
    """ClusterWCRBFNet.__call__ -- src/irbfn_mpc/model.py:393-414: (out, logits).  numpy (float64) or torch tensors
    (autograd: the oracle of the cluster VJP)."""
    p = params["params"] if "params" in params else params
    xp = _ns(x)
    if xp is _NP:
        x = np.asarray(x, np.float64)
        p = {g: {n: np.asarray(v, np.float64) for n, v in d.items()} for g, d in p.items()}
    all_x = rbf_layer(x, p["rbf_list"]["centers"], p["rbf_list"]["log_sigs"], cfg["basis_func"])   # [B,R,K]  model.py:400
    logits = x @ p["cluster"]["kernel"] + p["cluster"]["bias"]      # nn.Dense(num_regions)      :403
    if xp is _NP:
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        cluster_ind = e / e.sum(axis=1, keepdims=True)              # nn.softmax                 :404
    else:
        import torch
        cluster_ind = torch.softmax(logits, dim=1)
    rbf_out = (cluster_ind[:, :, None] * all_x).sum(1)              # :405-409
    out = rbf_out @ p["linear"]["kernel"] + p["linear"]["bias"]     # :412
    return out, logits


def train_fullint_withcluster_loss(cfg, params, x, y, cluster_ids, dyn_params):
    """scripts/train_nmpc_frenet.py:424-453 -- the Frenet full-integration loss of a ClusterWCRBFNet plus
    optax.softmax_cross_entropy(logits, cluster_ids).mean() (= -sum(labels * log_softmax(logits), -1).mean(); optax is
    un-pinned, published definition).  torch tensors (autograd) or numpy."""
    xp = _ns(x)
    init = x[:, [0, 0, 1, 2, 3, 5, 6, 7]]                                      # :428
    y_pred, logits = cluster_wcrbfnet_apply(cfg, params, x)                    # :430
    if xp is _NP:
        m = logits.max(axis=1, keepdims=True)
        logp = logits - m - np.log(np.exp(logits - m).sum(axis=1, keepdims=True))
        x_pred_u, x_u = np.hstack((init, y_pred)), np.hstack((init, y))
        absf = np.abs
    else:
        import torch
        logp = torch.log_softmax(logits, dim=1)
        x_pred_u, x_u = torch.hstack((init, y_pred)), torch.hstack((init, y))
        absf = lambda t: t.abs()
    cluster_loss = (-(cluster_ids * logp).sum(1)).mean()                       # :431
    actual = integrate_frenet_mult(x_u, dyn_params)                            # :439
    pred = integrate_frenet_mult(x_pred_u, dyn_params)                         # :440
    return absf(y_pred - y).mean() + absf(pred - actual).mean() + cluster_loss # :433,441,444


def wcrbfnet_vjp(cfg: dict, params: dict, x: np.ndarray, gout: np.ndarray):
    """Hand-derived VJP of a-4 w.r.t. the parameters, float64 NumPy.
    Only for bases that depend on d^2 alone (gaussian*, inverse_quadratic,
    inverse_multiquadric, multiquadric, quadratic).  Returns a dict with the param
    pytree structure.  Replaces ``jax.value_and_grad`` (scripts/train_nmpc.py:297-298)."""
    p = params["params"] if "params" in params else params
    c = np.asarray(p["rbf_list"]["centers"], np.float64)
    ls = np.asarray(p["rbf_list"]["log_sigs"], np.float64)
    W = np.asarray(p["linear"]["kernel"], np.float64)
    x = np.asarray(x, np.float64)
    g = np.asarray(gout, np.float64)
    nsplit = len(cfg["activation_idx"])
    gamma = region_activation(x, cfg["num_regions"], nsplit, cfg["lower_bounds"],
                              cfg["upper_bounds"], cfg["delta"], cfg["dimension_ranges"])
    diff = x[:, None, None, :] - c[None]
    r2 = (diff ** 2).sum(-1)
    s2 = np.exp(-2.0 * ls)[None]
    d2 = r2 * s2
    name = cfg["basis_func"]
    if name in ("gaussian", "gaussian_wide", "gaussian_wider"):
        a = {"gaussian": 1.0, "gaussian_wide": 0.1, "gaussian_wider": 0.01}[name]
        phi = np.exp(-a * d2)
        dphi = -a * phi
    elif name == "inverse_quadratic":
        phi = 1.0 / (1.0 + d2)
        dphi = -phi ** 2
    elif name == "inverse_multiquadric":
        phi = (1.0 + d2) ** -0.5
        dphi = -0.5 * phi ** 3
    elif name == "multiquadric":
        phi = (1.0 + d2) ** 0.5
        dphi = 0.5 / phi
    elif name == "quadratic":
        phi = d2
        dphi = np.ones_like(d2)
    else:
        raise ValueError(f"hand VJP not defined for basis {name}")
    h = (gamma[:, :, None] * phi).sum(1)
    hbar = g @ W.T                                   # [B,K]
    G = hbar[:, None, :] * gamma[:, :, None]         # [B,R,K]
    t = G * dphi
    g_centers = (t * s2)[..., None] * (-2.0) * diff
    return {"params": {
        "rbf_list": {"centers": g_centers.sum(0), "log_sigs": (t * (-2.0 * d2)).sum(0)},
        "linear": {"kernel": h.T @ g, "bias": g.sum(0)},
    }}


# --------------------------------------------------------------------------
# a-7  roll-out B -- src/irbfn_mpc/dynamics.py:9-91 (one step), :94-100 (scan)
# --------------------------------------------------------------------------
def _unpack_dyn(params):
    # dynamics.py:24-36
    return tuple(params[i] for i in range(13))


def st_rhs(x, accl_in, sv_in, params):
    """Returns (f, f_ks, V) with f the dynamic RHS (dynamics.py:49-76) and f_ks the
    kinematic RHS (:78-88); x[...,7], controls broadcastable.  Clips follow :40-47."""
    xp = _ns(x)
    mu, m, I, lf, lr, C_Sf, C_Sr, h, dt, sv_max, a_max, s_max, v_max = _unpack_dyn(params)
    g = G_ACC
    DELTA = xp.clip(x[..., 2], -s_max, s_max)       # dynamics.py:40
    V = xp.clip(x[..., 3], -v_max, v_max)           # :41
    PSI = x[..., 4]
    PSI_DOT = x[..., 5]
    BETA = x[..., 6]
    ACCL = xp.clip(accl_in, -a_max, a_max)          # :46
    STEER_VEL = xp.clip(sv_in, -sv_max, sv_max)     # :47
    zero = xp.zeros_like(V)
    f = [
        V * xp.cos(PSI + BETA),                     # :51
        V * xp.sin(PSI + BETA),                     # :52
        STEER_VEL + zero,                           # :53
        ACCL + zero,                                # :54
        PSI_DOT,                                    # :55
        ((mu * m) / (I * (lf + lr)))                # :56-66
        * (
            lf * C_Sf * (g * lr - ACCL * h) * DELTA
            + (lr * C_Sr * (g * lf + ACCL * h) - lf * C_Sf * (g * lr - ACCL * h)) * BETA
            - (lf * lf * C_Sf * (g * lr - ACCL * h) + lr * lr * C_Sr * (g * lf + ACCL * h))
            * (PSI_DOT / V)
        ),
        (mu / (V * (lr + lf)))                      # :67-74
        * (
            C_Sf * (g * lr - ACCL * h) * DELTA
            - (C_Sr * (g * lf + ACCL * h) + C_Sf * (g * lr - ACCL * h)) * BETA
            + (C_Sr * (g * lf + ACCL * h) * lr - C_Sf * (g * lr - ACCL * h) * lf) * (PSI_DOT / V)
        )
        - PSI_DOT,
    ]
    f_ks = [
        V * xp.cos(PSI),                            # :80
        V * xp.sin(PSI),                            # :81
        STEER_VEL + zero,                           # :82
        ACCL + zero,                                # :83
        (V / (lr + lf)) * xp.tan(DELTA),            # :84
        zero,                                       # :85
        zero,                                       # :86
    ]
    return xp.stack(f, -1), xp.stack(f_ks, -1), V


def dynamic_st_onestep(x, seq, params):
    """x[B,7], seq[B,2] = (accl, steer_vel).  x_new = x + select(V > 3, f, f_ks) * dt
    (dynamics.py:90)."""
    xp = _ns(x)
    if xp is _NP:
        with np.errstate(divide="ignore", invalid="ignore"):
            f, f_ks, V = st_rhs(x, seq[..., 0], seq[..., 1], params)
    else:
        f, f_ks, V = st_rhs(x, seq[..., 0], seq[..., 1], params)
    dt = params[8]
    return x + xp.where((V > 3.0)[..., None], f, f_ks) * dt


def integrate_st_mult(x_and_pred_u, params, T: Optional[int] = None):
    """x_and_pred_u[B, 7+2T] -> all_states[B,T,7].  seq = u.reshape(T,2,order="F")
    i.e. u = [a_0..a_{T-1}, sv_0..sv_{T-1}] (dynamics.py:98; the reference hard-codes T=5)."""
    xp = _ns(x_and_pred_u)
    nu = x_and_pred_u.shape[1] - 7
    T = nu // 2 if T is None else T
    x = x_and_pred_u[:, :7]
    states = []
    for t in range(T):
        seq = xp.stack([x_and_pred_u[:, 7 + t], x_and_pred_u[:, 7 + T + t]], -1)
        x = dynamic_st_onestep(x, seq, params)
        states.append(x)
    return xp.stack(states, 1)


def integrate_st_ks_mult(x_and_pred_u, params, T: Optional[int] = None):
    """Scan of the kinematic-only one-step map (a-8 applied T times): the BASELINE cfg-4
    'single-track kinematic' roll-out.  Same I/O as integrate_st_mult."""
    xp = _ns(x_and_pred_u)
    nu = x_and_pred_u.shape[1] - 7
    T = nu // 2 if T is None else T
    x = x_and_pred_u[:, :7]
    states = []
    for t in range(T):
        x_u = xp.stack([x[:, i] for i in range(7)]
                       + [x_and_pred_u[:, 7 + t], x_and_pred_u[:, 7 + T + t]], -1)
        x = dynamic_st_onestep_aux(x_u, params)
        states.append(x)
    return xp.stack(states, 1)


# --------------------------------------------------------------------------
# a-8  roll-out B-ks -- src/irbfn_mpc/dynamics.py:103-187
# --------------------------------------------------------------------------
def dynamic_st_onestep_aux(x_u, params):
    """x_u[B,9] = 7 state + [a, sv] -> [B,7]; kinematic RHS only (dynamics.py:186)."""
    xp = _ns(x_u)
    mu, m, I, lf, lr, C_Sf, C_Sr, h, dt, sv_max, a_max, s_max, v_max = _unpack_dyn(params)
    DELTA = xp.clip(x_u[:, 2], -s_max, s_max)       # :135
    V = xp.clip(x_u[:, 3], -v_max, v_max)           # :136
    PSI = x_u[:, 4]
    ACCL = xp.clip(x_u[:, 7], -a_max, a_max)        # :141
    STEER_VEL = xp.clip(x_u[:, 8], -sv_max, sv_max)  # :142
    zero = xp.zeros_like(V)
    f_ks = xp.stack([
        V * xp.cos(PSI), V * xp.sin(PSI), STEER_VEL, ACCL,
        (V / (lr + lf)) * xp.tan(DELTA), zero, zero], -1)   # :173-183
    return x_u[:, :7] + f_ks * dt                    # :186


# --------------------------------------------------------------------------
# a-9  roll-out C -- src/irbfn_mpc/dynamics.py:190-281 (one step), :284-290 (scan)
# --------------------------------------------------------------------------
def dynamic_frenet_onestep(x, seq, params):
    """x[B,8] = [s, ey, delta, vx, vy, wz, epsi, cur]; only deriv_x_ls is applied (:280)."""
    xp = _ns(x)
    MU, M, I, LF, LR, C_SF, C_SR, h, dt, sv_max, a_max, s_max, v_max = _unpack_dyn(params)
    ey = x[:, 1]
    delta = xp.clip(x[:, 2], -s_max, s_max)          # :227
    vx = x[:, 3]                                     # :229 (not clipped)
    epsi = x[:, 6]
    cur = x[:, 7]
    a = xp.clip(seq[:, 0], -a_max, a_max)            # :235
    deltv = xp.clip(seq[:, 1], -sv_max, sv_max)      # :236
    zero = xp.zeros_like(vx)
    deriv_x_ls = xp.stack([
        (vx * xp.cos(epsi)) / (1 - ey * cur),        # :268
        (vx * xp.sin(epsi)),                         # :269
        deltv,                                       # :270
        a,                                           # :271
        zero, zero,                                  # :272-273
        (vx * xp.tan(delta)) / (LR + LF) - cur * ((vx * xp.cos(epsi)) / (1 - cur * ey)),  # :274-275
        zero,                                        # :276
    ], -1)
    return x + deriv_x_ls * dt                       # :280


def integrate_frenet_mult(x_and_pred_u, params, T: Optional[int] = None):
    """x_and_pred_u[B, 8+2T] -> [B,T,8] (dynamics.py:284-290)."""
    xp = _ns(x_and_pred_u)
    nu = x_and_pred_u.shape[1] - 8
    T = nu // 2 if T is None else T
    x = x_and_pred_u[:, :8]
    states = []
    for t in range(T):
        seq = xp.stack([x_and_pred_u[:, 8 + t], x_and_pred_u[:, 8 + T + t]], -1)
        x = dynamic_frenet_onestep(x, seq, params)
        states.append(x)
    return xp.stack(states, 1)


# --------------------------------------------------------------------------
# a-6  roll-out A -- inline kinematic bicycle, scripts/train_nmpc.py:306-374
#      (NumPy twin at scripts/eval_irbfn_dnmpc.py:95-159)
# --------------------------------------------------------------------------
FULLINT_DT, FULLINT_WB = 0.1, 0.33                   # train_nmpc.py:307-308
FULLINT_MAX_SPEED, FULLINT_MIN_SPEED, FULLINT_MAX_STEER = 7.0, 0.0, 0.4189  # :309-311


def rollout_fullint(v0, u, T: Optional[int] = None):
    """v0[B] (clipped to [0,7] as train_nmpc.py:319), u[B,2T] = [a_0.., sv_0..] (:358-359)
    -> states[B,T,5] = (x, y, delta, v, yaw) after every step (the reference keeps steps
    1 and T, :367-374)."""
    xp = _ns(u)
    T = u.shape[1] // 2 if T is None else T
    DT, WB = FULLINT_DT, FULLINT_WB
    x_ = xp.zeros_like(v0)
    y_ = xp.zeros_like(v0)
    delta = xp.zeros_like(v0)
    v = xp.clip(v0, FULLINT_MIN_SPEED, FULLINT_MAX_SPEED)     # :319
    yaw = xp.zeros_like(v0)
    out = []
    for i in range(T):
        a = u[:, i]                                  # :358
        delta_v = u[:, i + T]                        # :359
        x_ = x_ + v * xp.cos(yaw) * DT               # :360
        y_ = y_ + v * xp.sin(yaw) * DT               # :361
        delta = delta + delta_v * DT                 # :362
        delta = xp.clip(delta, -FULLINT_MAX_STEER, FULLINT_MAX_STEER)   # :363
        v = v + a * DT                               # :364
        v = xp.clip(v, FULLINT_MIN_SPEED, FULLINT_MAX_SPEED)            # :365
        yaw = yaw + (v / WB) * xp.tan(delta) * DT    # :366
        out.append(xp.stack([x_, y_, delta, v, yaw], -1))
    return xp.stack(out, 1)


# --------------------------------------------------------------------------
# a-10  roll-out D -- cubic spiral, src/irbfn_mpc/planner_utils.py:8-77
# --------------------------------------------------------------------------
SPIRAL_N = 9                                         # planner_utils.py:8
PARAM_MAT = np.array([                               # planner_utils.py:10-17
    [1.0, 0.0, 0.0, 0.0],
    [-11.0 / 2, 9.0, -9.0 / 2, 1.0],
    [9.0, -45.0 / 2, 18.0, -9.0 / 2],
    [-9.0 / 2, 27.0 / 2, -27.0 / 2, 9.0 / 2],
])


def params_to_coefs(params):
    """params[B,5] = (k0,k1,k2,k3,s) -> coefs[B,4]  (planner_utils.py:20-29)."""
    xp = _ns(params)
    s = params[:, 4]
    k = [params[:, i] for i in range(4)]
    rows = []
    for r in range(4):
        acc = None
        for j in range(4):
            term = xp.const(PARAM_MAT[r, j], params) * k[j]
            acc = term if acc is None else acc + term
        rows.append(acc)
    rows[1] = rows[1] / s                            # :26
    rows[2] = rows[2] / s ** 2                       # :27
    rows[3] = rows[3] / s ** 3                       # :28
    return xp.stack(rows, -1)


def get_curvature_theta(coefs, s_cur):
    """planner_utils.py:32-41."""
    out = 0.0
    out2 = 0.0
    for i in range(4):
        temp = coefs[:, i] * s_cur ** i
        out = out + temp
        out2 = out2 + temp * s_cur / (i + 1)
    return out, out2


def integrate_path_mult(params, N: int = SPIRAL_N):
    """params[B,5] -> all_states[B,N,6] = [x, y, theta, kappa, dx, dy]
    (planner_utils.py:62-77, step :44-59)."""
    xp = _ns(params)
    coefs = params_to_coefs(params)
    s = params[:, 4]
    zero = xp.zeros_like(s)
    st = [zero, zero, zero, coefs[:, 0], zero, zero]  # :67-70
    out = []
    for i in range(N):
        # jnp.linspace(0, s, N)[i] = 0*(1-t) + s*t, t = i/(N-1); endpoint exact.  (:71)
        sk = s * xp.const(i / (N - 1), params) if i < N - 1 else s
        k = float(i + 1)                              # :72
        kappa_k, theta_k = get_curvature_theta(coefs, sk)     # :46
        dx = st[4] * (1 - 1 / k) + (xp.cos(theta_k) + xp.cos(st[2])) / 2 / k   # :47-50
        dy = st[5] * (1 - 1 / k) + (xp.sin(theta_k) + xp.sin(st[2])) / 2 / k   # :51-54
        x = sk * dx                                   # :55
        y = sk * dy                                   # :56
        st = [x, y, theta_k, kappa_k, dx, dy]         # :57
        out.append(xp.stack(st, -1))
    return xp.stack(out, 1)


# --------------------------------------------------------------------------
# loss compositions that define the VJP seeds (callers)
# --------------------------------------------------------------------------
def l2_loss(pred, target):
    """optax.l2_loss = 0.5 * (pred - target)^2."""
    return 0.5 * (pred - target) ** 2


def train_oneint_loss(cfg, params, x, y, dyn_params):
    """scripts/train_nmpc.py:258-295 -- loss of train_step_oneint (works on torch tensors
    for autograd or on numpy)."""
    xp = _ns(x)
    B = x.shape[0]
    zero = xp.zeros_like(x[:, 0])
    init = xp.stack([zero, zero, zero, x[:, 0], zero, x[:, 6], x[:, 5]], -1)   # :260-266
    y_pred = wcrbfnet_apply(cfg, params, x)                                    # :269
    if xp is _NP:
        x_pred_u = np.hstack((init, y_pred))
        x_u = np.hstack((init, y))
    else:
        import torch
        x_pred_u = torch.hstack((init, y_pred))
        x_u = torch.hstack((init, y))
    actual = dynamic_st_onestep_aux(x_u, dyn_params)                           # :275
    pred = dynamic_st_onestep_aux(x_pred_u, dyn_params)                        # :276
    idx = [0, 1, 3, 4]
    return l2_loss(y_pred, y).mean() + l2_loss(pred[:, idx], actual[:, idx]).mean()  # :286-292


def train_fullint_loss(cfg, params, x, y):
    """scripts/train_nmpc.py:303-390 -- loss of train_step_fullint: L1 on the first accel / steer-vel
    outputs + L1 on the final state of the 5-step inline bicycle (the middle term is identically 0)."""
    xp = _ns(x)
    T = y.shape[1] // 2
    y_pred = wcrbfnet_apply(cfg, params, x)                                    # :313
    fin_a = rollout_fullint(x[:, 0], y)[:, -1]                                 # :329-347
    fin_p = rollout_fullint(x[:, 0], y_pred)[:, -1]                            # :356-374
    cols = [0, T]
    absf = np.abs if xp is _NP else (lambda t: t.abs())
    return absf(y_pred[:, cols] - y[:, cols]).mean() + absf(fin_p - fin_a).mean()   # :386-390


def train_frenet_fullint_loss(cfg, params, x, y, dyn_params):
    """scripts/train_nmpc_frenet.py:394-421 -- loss of the Frenet train_step_fullint: L1 on the predictions + L1 on
    ALL states of the T-step Frenet roll-out (works on torch tensors for autograd or on numpy)."""
    xp = _ns(x)
    init = x[:, [0, 0, 1, 2, 3, 5, 6, 7]]                                      # :398
    y_pred = wcrbfnet_apply(cfg, params, x)                                    # :401
    if xp is _NP:
        x_pred_u, x_u = np.hstack((init, y_pred)), np.hstack((init, y))
        absf = np.abs
    else:
        import torch
        x_pred_u, x_u = torch.hstack((init, y_pred)), torch.hstack((init, y))
        absf = lambda t: t.abs()
    actual = integrate_frenet_mult(x_u, dyn_params)                            # :407
    pred = integrate_frenet_mult(x_pred_u, dyn_params)                         # :408
    return absf(y_pred - y).mean() + absf(pred - actual).mean()                # :402,409,412


def clip_by_global_norm(grads_flat: np.ndarray, max_norm: float) -> np.ndarray:
    """optax.clip_by_global_norm (third-party, version un-pinned by pyproject.toml:7 -- parity unpinned):
    g if ||g|| < max_norm else g / ||g|| * max_norm."""
    gn = np.sqrt((grads_flat.astype(np.float64) ** 2).sum())
    return grads_flat if gn < max_norm else grads_flat / gn * max_norm


def adam_update(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """optax.adam = scale_by_adam(b1, b2, eps, eps_root=0) then scale(-lr); t = incremented count.
    (Published algorithm of optax; parity unpinned for the same reason.)"""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat, vhat = m / (1 - b1 ** t), v / (1 - b2 ** t)
    return p - lr * mhat / (np.sqrt(vhat) + eps), m, v


# --------------------------------------------------------------------------
# helpers for tests / fixtures
# --------------------------------------------------------------------------
def cast_params(params: dict, dtype) -> dict:
    p = params["params"] if "params" in params else params
    return {"params": {
        "rbf_list": {"centers": np.asarray(p["rbf_list"]["centers"], dtype),
                     "log_sigs": np.asarray(p["rbf_list"]["log_sigs"], dtype)},
        "linear": {"kernel": np.asarray(p["linear"]["kernel"], dtype),
                   "bias": np.asarray(p["linear"]["bias"], dtype)}}}


def torch_params(params: dict, dtype=None, requires_grad: bool = False) -> dict:
    import torch
    dtype = dtype or torch.float64
    p = params["params"] if "params" in params else params

    def t(a):
        return torch.tensor(np.asarray(a), dtype=dtype, requires_grad=requires_grad)
    return {"params": {
        "rbf_list": {"centers": t(p["rbf_list"]["centers"]), "log_sigs": t(p["rbf_list"]["log_sigs"])},
        "linear": {"kernel": t(p["linear"]["kernel"]), "bias": t(p["linear"]["bias"])}}}


DYN_PARAMS_KAT1 = [1.0, 1.0489, 0.04712, 0.15875, 0.17145, 5.0, 5.0, 0.074, 0.1, 3.2, 9.51,
                   0.4189, 7.0]   # scripts/test_dynamics.ipynb cell 1


# ---------------------------------------------------------------------------------------------------
# planner front end: literal per-row restatements
# ---------------------------------------------------------------------------------------------------
def plan_query_cartesian(pose, goal):
    """One (pose, goal) pair -> (rbf_in [7] float32, mirror, states [7] float32).
    src/irbfn_mpc/irbfn_planner.py:160-166 (pose), :181-201 (query), :240 (states)."""
    x, y, delta, v, theta, angv, beta = [float(t) for t in pose]
    ref_point = np.asarray(goal, np.float64)
    rot = np.array([[np.cos(-theta), -np.sin(-theta)], [np.sin(-theta), np.cos(-theta)]])     # :181-183
    goal_local = np.dot(rot, (ref_point[:2] - np.array([x, y])))                               # :184
    goal_theta = ref_point[2] - theta                                                          # :185
    goal_needs_mirror = goal_local[1] < 0                                                      # :188
    rbf_in = np.array([v, goal_local[0],
                       -goal_local[1] if goal_needs_mirror else goal_local[1],
                       -goal_theta % np.pi if goal_needs_mirror else goal_theta % np.pi,
                       ref_point[3], beta, angv]).astype(np.float32)                           # :189-201
    states = np.array([x, y, delta, v, theta, angv, beta]).astype(np.float32)                 # :240
    return rbf_in, bool(goal_needs_mirror), states


def plan_query_frenet(fr, vx_goal):
    """src/irbfn_mpc/irbfn_planner.py:456-502.  fr = [s, ey, delta, vx, vy, wz, epsi, curv]."""
    s_, ey, delta, vx, vy, wz, epsi, curv = [float(t) for t in fr]
    goal_needs_mirror = ey < -0.05                                                             # :457
    rbf_in = np.array([-ey if goal_needs_mirror else ey, delta, vx,
                       -vy if goal_needs_mirror else vy, float(vx_goal),
                       -wz if goal_needs_mirror else wz,
                       -epsi if goal_needs_mirror else epsi, curv]).astype(np.float32)         # :459-478
    states = np.array([s_, ey, delta, vx, vy, wz, epsi, curv]).astype(np.float32)             # :491-502
    return rbf_in, bool(goal_needs_mirror), states


def unmirror_controls(pred_u, mirror, sv_ind):
    """irbfn_planner.py:203-204 / :487-488: pred_u[b, sv_ind:] *= -1 for mirrored rows."""
    out = np.array(pred_u, copy=True)
    rows = np.flatnonzero(np.asarray(mirror))
    out[rows, sv_ind:] = -out[rows, sv_ind:]
    return out


def lut_grid_lookup(input_keys, shape, lookup):
    """src/irbfn_mpc/explicit_planner.py:165-172, one query: per-axis index and the flat row index."""
    closest_ind = []
    for val_ind, val in enumerate(lookup):
        closest_ind.append(min(shape[val_ind] - 1, np.searchsorted(input_keys[val_ind], val, side="right")))
    return closest_ind, int(np.ravel_multi_index(closest_ind, shape))


def lut_nearest(inputs, lookup):
    """explicit_planner.py:219, :383 -- scipy.spatial.KDTree(inputs).query(lookup) -> (distance, index)."""
    import scipy.spatial as ss
    return ss.KDTree(inputs).query(lookup)


# --------------------------------------------------------------------------
# f-4 way-point geometry -- src/irbfn_mpc/planner_utils.py:109-233 (numba in the reference: the same lines without
# the decorator; numba's typing of the mixed float32 / float64 expressions cannot be checked here: parity unpinned)
# --------------------------------------------------------------------------
def nearest_point(point, trajectory):
    diffs = (trajectory[1:, :] - trajectory[:-1, :]).astype(np.float32)           # :125
    l2s = diffs[:, 0] ** 2 + diffs[:, 1] ** 2                                     # :126
    dots = np.empty((trajectory.shape[0] - 1,))
    for i in range(dots.shape[0]):
        lhs = (point - trajectory[i, :]).astype(np.float32)                       # :129
        dots[i] = np.float32(lhs[0] * diffs[i, 0]) + np.float32(lhs[1] * diffs[i, 1])   # np.dot of float32 pairs :130
    t = dots / l2s
    t[t < 0.0] = 0.0
    t[t > 1.0] = 1.0
    projections = trajectory[:-1, :] + (t * diffs.T).T                            # :134
    dists = np.empty((projections.shape[0],))
    for i in range(dists.shape[0]):
        temp = point - projections[i]
        dists[i] = np.sqrt(np.sum(temp * temp))
    k = int(np.argmin(dists))
    return projections[k], dists[k], t[k], k


def intersect_point(point, radius, trajectory, t=0.0, wrap=False):
    start_i = int(t)
    start_t = np.float32(t % 1.0)
    traj = np.ascontiguousarray(trajectory).astype(np.float32)                    # :160
    n = traj.shape[0]
    radius = np.float32(radius)

    def hit(i, first):
        start = traj[i % n, :]
        end = traj[(i + 1) % n, :] + np.float32(1e-6)                             # :163
        V = end - start
        a = np.float32(V[0] * V[0]) + np.float32(V[1] * V[1])
        d = start.astype(np.float64) - point
        b = 2.0 * (float(V[0]) * d[0] + float(V[1]) * d[1])
        c = (float(np.float32(start[0] * start[0]) + np.float32(start[1] * start[1])) + (point[0] * point[0] + point[1] * point[1])
             - 2.0 * (float(start[0]) * point[0] + float(start[1]) * point[1]) - float(radius) * float(radius))
        disc = b * b - 4.0 * float(a) * c
        if not disc >= 0:
            return None
        disc = np.sqrt(disc)
        t1, t2 = (-b - disc) / (2.0 * float(a)), (-b + disc) / (2.0 * float(a))
        if first:
            if 0.0 <= t1 <= 1.0 and t1 >= start_t:
                tt = t1
            elif 0.0 <= t2 <= 1.0 and t2 >= start_t:
                tt = t2
            else:
                return None
        else:
            if 0.0 <= t1 <= 1.0:
                tt = t1
            elif 0.0 <= t2 <= 1.0:
                tt = t2
            else:
                return None
        return (start.astype(np.float64) + tt * V.astype(np.float64)).astype(np.float32), i, np.float32(tt)

    for i in range(start_i, n - 1):
        r = hit(i, i == start_i)
        if r is not None:
            return r
    if wrap:
        for i in range(-1, start_i):
            r = hit(i, False)
            if r is not None:
                return r
    return None, None, None
