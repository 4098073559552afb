"""NumPy statements of the hand-derived roll-out adjoints that the HIP kernels in
irbfn_amd/csrc/rollout_vjp.hip implement -- TEST INFRASTRUCTURE ONLY.

tests/test_oracle_cpu.py checks them against ``torch.autograd`` of the forward restatement
(oracle/irbfn_oracle.py), i.e. against what ``jax.grad`` computes in the reference
(scripts/train_nmpc.py:275-276,356-374; scripts/train_nmpc_frenet.py:408-409;
deprecated/train_newlut.py:194-199).  clip() gradient: 1 inside, 0 outside, ``tie`` on a bound.
"""
import numpy as np

from . import irbfn_oracle as o


def clipgrad(v, lo, hi, tie=0.5):
    return np.where((v > lo) & (v < hi), 1.0, np.where((v == lo) | (v == hi), tie, 0.0))


def vjp_st_ks(xu, dp, gs, tie=0.5):
    B, T = xu.shape[0], (xu.shape[1] - 7) // 2
    lf, lr, dt, svm, am, sm, vm = dp[3], dp[4], dp[8], dp[9], dp[10], dp[11], dp[12]
    Lw = lr + lf
    g = np.zeros_like(xu)
    s = xu[:, :7].copy()
    pre = []
    for t in range(T):
        pre.append((s[:, 2].copy(), s[:, 3].copy(), s[:, 4].copy()))
        s = o.dynamic_st_onestep_aux(np.hstack([s, xu[:, 7 + t:8 + t], xu[:, 7 + T + t:8 + T + t]]), dp)
    lam = np.zeros((B, 7))
    for t in range(T - 1, -1, -1):
        lam = lam + gs[:, t]
        d, v, psi = pre[t]
        D, V = np.clip(d, -sm, sm), np.clip(v, -vm, vm)
        md, mv = clipgrad(d, -sm, sm, tie), clipgrad(v, -vm, vm, tie)
        ma, ms = clipgrad(xu[:, 7 + t], -am, am, tie), clipgrad(xu[:, 7 + T + t], -svm, svm, tie)
        cp, sp, td = np.cos(psi), np.sin(psi), np.tan(D)
        g[:, 7 + t] = ma * dt * lam[:, 3]
        g[:, 7 + T + t] = ms * dt * lam[:, 2]
        l2 = lam[:, 2] + md * lam[:, 4] * (V / Lw) * (1 + td * td) * dt
        l3 = lam[:, 3] + mv * dt * (lam[:, 0] * cp + lam[:, 1] * sp + lam[:, 4] * td / Lw)
        l4 = lam[:, 4] + dt * V * (-lam[:, 0] * sp + lam[:, 1] * cp)
        lam[:, 2], lam[:, 3], lam[:, 4] = l2, l3, l4
    g[:, :7] = lam
    return g


def vjp_fullint(v0, u, gs, tie=0.5):
    B, T = u.shape[0], u.shape[1] // 2
    DT, WB, VMAX, VMIN, SMAX = 0.1, 0.33, 7.0, 0.0, 0.4189
    s = np.zeros((B, 5))
    s[:, 3] = np.clip(v0, VMIN, VMAX)
    pre = []
    for t in range(T):
        pre.append((s[:, 2].copy(), s[:, 3].copy(), s[:, 4].copy()))
        a, dv = u[:, t], u[:, T + t]
        x = s[:, 0] + s[:, 3] * np.cos(s[:, 4]) * DT
        y = s[:, 1] + s[:, 3] * np.sin(s[:, 4]) * DT
        d = np.clip(s[:, 2] + dv * DT, -SMAX, SMAX)
        v = np.clip(s[:, 3] + a * DT, VMIN, VMAX)
        yaw = s[:, 4] + (v / WB) * np.tan(d) * DT
        s = np.stack([x, y, d, v, yaw], -1)
    lam = np.zeros((B, 5))
    gu = np.zeros_like(u)
    for t in range(T - 1, -1, -1):
        lam = lam + gs[:, t]
        d0, v0_, psi = pre[t]
        dpre, vpre = d0 + u[:, T + t] * DT, v0_ + u[:, t] * DT
        d1, v1 = np.clip(dpre, -SMAX, SMAX), np.clip(vpre, VMIN, VMAX)
        md, mv = clipgrad(dpre, -SMAX, SMAX, tie), clipgrad(vpre, VMIN, VMAX, tie)
        td, cp, sp = np.tan(d1), np.cos(psi), np.sin(psi)
        Ld = lam[:, 2] + lam[:, 4] * (v1 / WB) * (1 + td * td) * DT
        Lv = lam[:, 3] + lam[:, 4] * td * DT / WB
        gu[:, t] = mv * Lv * DT
        gu[:, T + t] = md * Ld * DT
        l2 = md * Ld
        l3 = mv * Lv + DT * (lam[:, 0] * cp + lam[:, 1] * sp)
        l4 = lam[:, 4] + DT * v0_ * (-lam[:, 0] * sp + lam[:, 1] * cp)
        lam[:, 2], lam[:, 3], lam[:, 4] = l2, l3, l4
    return clipgrad(v0, VMIN, VMAX, tie) * lam[:, 3], gu


def vjp_frenet(xu, dp, gs, tie=0.5):
    B, T = xu.shape[0], (xu.shape[1] - 8) // 2
    LF, LR, dt, svm, am, sm = dp[3], dp[4], dp[8], dp[9], dp[10], dp[11]
    Lw = LR + LF
    g = np.zeros_like(xu)
    s = xu[:, :8].copy()
    cur = s[:, 7].copy()
    pre = []
    for t in range(T):
        pre.append((s[:, 1].copy(), s[:, 2].copy(), s[:, 3].copy(), s[:, 6].copy()))
        s = o.dynamic_frenet_onestep(s, np.stack([xu[:, 8 + t], xu[:, 8 + T + t]], -1), dp)
    lam = np.zeros((B, 8))
    for t in range(T - 1, -1, -1):
        lam = lam + gs[:, t]
        ey, d, vx, ep = pre[t]
        dc = np.clip(d, -sm, sm)
        md = clipgrad(d, -sm, sm, tie)
        ma, ms = clipgrad(xu[:, 8 + t], -am, am, tie), clipgrad(xu[:, 8 + T + t], -svm, svm, tie)
        ce, se, td = np.cos(ep), np.sin(ep), np.tan(dc)
        den = 1 - ey * cur
        d0 = vx * ce / den
        A = lam[:, 0] * dt - lam[:, 6] * dt * cur
        g[:, 8 + t] = ma * dt * lam[:, 3]
        g[:, 8 + T + t] = ms * dt * lam[:, 2]
        l1 = lam[:, 1] + A * (vx * ce * cur / den ** 2)
        l2 = lam[:, 2] + md * lam[:, 6] * dt * vx * (1 + td * td) / Lw
        l3 = lam[:, 3] + A * ce / den + lam[:, 1] * dt * se + lam[:, 6] * dt * td / Lw
        l6 = lam[:, 6] + A * (-vx * se / den) + lam[:, 1] * dt * vx * ce
        l7 = lam[:, 7] + A * (vx * ce * ey / den ** 2) - lam[:, 6] * dt * d0
        lam[:, 1], lam[:, 2], lam[:, 3], lam[:, 6], lam[:, 7] = l1, l2, l3, l6, l7
    g[:, :8] = lam
    return g


def vjp_spiral(q, gs, N=9):
    B = q.shape[0]
    c = o.params_to_coefs(q)
    s = q[:, 4]
    st = o.integrate_path_mult(q, N)
    th, dxs, dys = st[:, :, 2], st[:, :, 4], st[:, :, 5]
    gc = np.zeros((B, 4))
    g_s, ldx, ldy, lth = np.zeros(B), np.zeros(B), np.zeros(B), np.zeros(B)
    for i in range(N - 1, -1, -1):
        tau = i / (N - 1) if i < N - 1 else 1.0
        sk, k = s * tau, float(i + 1)
        thi = th[:, i]
        thp = th[:, i - 1] if i > 0 else np.zeros(B)
        gx, gy, gth, gka, gdx, gdy = [gs[:, i, j] for j in range(6)]
        Gdx, Gdy = gdx + ldx + sk * gx, gdy + ldy + sk * gy
        gsk = gx * dxs[:, i] + gy * dys[:, i]
        Gth = gth + lth + (Gdx * (-np.sin(thi)) + Gdy * np.cos(thi)) / (2 * k)
        lth = (Gdx * (-np.sin(thp)) + Gdy * np.cos(thp)) / (2 * k)
        ldx, ldy = Gdx * (1 - 1 / k), Gdy * (1 - 1 / k)
        pw, kap, dkap, pwm1 = np.ones(B), np.zeros(B), np.zeros(B), np.zeros(B)
        for j in range(4):
            gc[:, j] += Gth * (pw * sk) / (j + 1) + gka * pw
            kap += c[:, j] * pw
            dkap += j * c[:, j] * pwm1
            pwm1 = pw
            pw = pw * sk
        gsk = gsk + Gth * kap + gka * dkap
        g_s += gsk * tau
    inv = np.ones(B)
    gq = np.zeros((B, 4))
    for r in range(4):
        for m in range(4):
            gq[:, m] += gc[:, r] * o.PARAM_MAT[r, m] * inv
        g_s += -r * c[:, r] / s * gc[:, r]
        inv = inv / s
    return np.hstack([gq, g_s[:, None]])
