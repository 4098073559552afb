// One-step maps of the roll-outs (device functions shared by the stand-alone roll-out kernels and
// the fused forward+roll-out kernel).  One lane integrates one trajectory; state lives in VGPRs.
#pragma once

#include "common.h"

namespace irbfn {

// Straight-line sin/cos for the roll-outs.  ocml's sinf/cosf/tanf cost ~430 VALU instructions per
// step with divergent range-reduction branches; one step is a serial dependent chain, so the stand-
// alone roll-out was latency bound at 14-24 % of HBM.  Here: Cody-Waite reduction by pi/2 with three
// FMA steps (exact to |x| < 8192, beyond that the ocml path), cephes minimax polynomials on
// [-pi/4, pi/4] (<= 1 ulp each), quadrant select.  Total error <= 2 ulp; one reduction serves sin and cos.
__device__ __forceinline__ void sincos_fast(float x, float& sn, float& cs) {
  if (__builtin_expect(!(fabsf(x) < 8192.0f), 0)) {   // huge, inf or nan: accurate slow path
    sn = sinf(x);
    cs = cosf(x);
    return;
  }
  const float kf = rintf(x * 0.636619772367581343f);  // x * 2/pi
  float r = __builtin_fmaf(-kf, 1.57079625129699707031e+00f, x);
  r = __builtin_fmaf(-kf, 7.54978941586159635335e-08f, r);
  r = __builtin_fmaf(-kf, 5.39030285815811905290e-15f, r);
  const float r2 = r * r;
  float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
  sp = __builtin_fmaf(sp * r2, r, r);
  float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
  cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  const int k = (int)kf;
  const float a = (k & 1) ? cp : sp;                  // sin: sp, cp, -sp, -cp
  const float b = (k & 1) ? sp : cp;                  // cos: cp, -sp, -cp, sp
  sn = (k & 2) ? -a : a;
  cs = ((k + 1) & 2) ? -b : b;
}

__device__ __forceinline__ float tan_fast(float x) {
  float sn, cs;
  sincos_fast(x, sn, cs);
  return sn / cs;
}

__device__ __forceinline__ float clipf(float v, float lo, float hi) {
  // jnp.clip = min(max(v, lo), hi); NaN propagates (fmaxf alone would swallow it)
  return v != v ? v : fminf(fmaxf(v, lo), hi);
}

// Single-track model, src/irbfn_mpc/dynamics.py:9-91.  SELECT=true: x + select(V>3, f, f_ks)*dt (:90);
// SELECT=false: kinematic RHS only = dynamic_st_onestep_aux (:103-187).
template <bool SELECT>
__device__ __forceinline__ void st_step(float (&s)[7], float accl_in, float sv_in, const DynParams& dp) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float g = 9.81f;                                 // dynamics.py:6
  const float mu = dp.p[0], m = dp.p[1], I = dp.p[2], lf = dp.p[3], lr = dp.p[4], C_Sf = dp.p[5],
              C_Sr = dp.p[6], h = dp.p[7], dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10],
              s_max = dp.p[11], v_max = dp.p[12];
  const float DELTA = clipf(s[2], -s_max, s_max);        // :40
  const float V = clipf(s[3], -v_max, v_max);            // :41
  const float PSI = s[4], PSI_DOT = s[5], BETA = s[6];
  const float ACCL = clipf(accl_in, -a_max, a_max);      // :46
  const float SV = clipf(sv_in, -sv_max, sv_max);        // :47
  float f0, f1, f4, f5, f6;
  if (SELECT && V > 3.0f) {                              // :49-76
    float sn, cs;
    sincos_fast(PSI + BETA, sn, cs);
    f0 = V * cs;
    f1 = V * sn;
    f4 = PSI_DOT;
    const float glr = g * lr - ACCL * h, glf = g * lf + ACCL * h;
    f5 = ((mu * m) / (I * (lf + lr))) *
         (lf * C_Sf * glr * DELTA + (lr * C_Sr * glf - lf * C_Sf * glr) * BETA -
          (lf * lf * C_Sf * glr + lr * lr * C_Sr * glf) * (PSI_DOT / V));
    f6 = (mu / (V * (lr + lf))) *
             (C_Sf * glr * DELTA - (C_Sr * glf + C_Sf * glr) * BETA +
              (C_Sr * glf * lr - C_Sf * glr * lf) * (PSI_DOT / V)) -
         PSI_DOT;
  } else {                                               // :78-88
    float sn, cs;
    sincos_fast(PSI, sn, cs);
    f0 = V * cs;
    f1 = V * sn;
    f4 = (V / (lr + lf)) * tan_fast(DELTA);
    f5 = 0.0f;
    f6 = 0.0f;
  }
  s[0] = s[0] + f0 * dt;
  s[1] = s[1] + f1 * dt;
  s[2] = s[2] + SV * dt;
  s[3] = s[3] + ACCL * dt;
  s[4] = s[4] + f4 * dt;
  s[5] = s[5] + f5 * dt;
  s[6] = s[6] + f6 * dt;
}

// Inline kinematic bicycle of train_step_fullint, scripts/train_nmpc.py:329-347 / :356-374.
__device__ __forceinline__ void fullint_step(float (&s)[5], float a, float dv) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float DT = 0.1f, WB = 0.33f, VMAX = 7.0f, VMIN = 0.0f, SMAX = 0.4189f;   // :307-311
  float sn, cs;
  sincos_fast(s[4], sn, cs);
  s[0] = s[0] + s[3] * cs * DT;                          // :360
  s[1] = s[1] + s[3] * sn * DT;                          // :361
  s[2] = clipf(s[2] + dv * DT, -SMAX, SMAX);             // :362-363
  s[3] = clipf(s[3] + a * DT, VMIN, VMAX);               // :364-365
  s[4] = s[4] + (s[3] / WB) * tan_fast(s[2]) * DT;       // :366
}

// Frenet model, low-speed RHS only, src/irbfn_mpc/dynamics.py:190-281 (:267-280).
__device__ __forceinline__ void frenet_step(float (&s)[8], float a_in, float dv_in, const DynParams& dp) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float LF = dp.p[3], LR = dp.p[4], dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10],
              s_max = dp.p[11];
  const float ey = s[1], delta = clipf(s[2], -s_max, s_max), vx = s[3], epsi = s[6], cur = s[7];
  const float a = clipf(a_in, -a_max, a_max);            // :235
  const float dv = clipf(dv_in, -sv_max, sv_max);        // :236
  float se, ce;
  sincos_fast(epsi, se, ce);
  const float d0 = (vx * ce) / (1.0f - ey * cur);        // :268
  const float d1 = vx * se;                              // :269
  const float d6 = (vx * tan_fast(delta)) / (LR + LF) - cur * ((vx * ce) / (1.0f - cur * ey));  // :274-275
  s[0] = s[0] + d0 * dt;
  s[1] = s[1] + d1 * dt;
  s[2] = s[2] + dv * dt;
  s[3] = s[3] + a * dt;
  s[4] = s[4] + 0.0f * dt;
  s[5] = s[5] + 0.0f * dt;
  s[6] = s[6] + d6 * dt;
  s[7] = s[7] + 0.0f * dt;
}

// Cubic spiral: params_to_coefs (planner_utils.py:20-29)
__device__ __forceinline__ void spiral_coefs(const float (&q)[5], float (&c)[4]) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float s = q[4];
  c[0] = 1.0f * q[0] + 0.0f * q[1] + 0.0f * q[2] + 0.0f * q[3];           // PARAM_MAT :10-17
  c[1] = (-11.0f / 2) * q[0] + 9.0f * q[1] + (-9.0f / 2) * q[2] + 1.0f * q[3];
  c[2] = 9.0f * q[0] + (-45.0f / 2) * q[1] + 18.0f * q[2] + (-9.0f / 2) * q[3];
  c[3] = (-9.0f / 2) * q[0] + (27.0f / 2) * q[1] + (-27.0f / 2) * q[2] + (9.0f / 2) * q[3];
  c[1] = c[1] / s;                                       // :26
  c[2] = c[2] / (s * s);                                 // :27
  c[3] = c[3] / (s * s * s);                             // :28
}

// integrate_one_step (planner_utils.py:44-59); st = [x, y, theta, kappa, dx, dy]; i = 0-based sample
__device__ __forceinline__ void spiral_step(float (&st)[6], const float (&c)[4], float s, int i, int N) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float sk = (i < N - 1) ? s * ((float)i / (float)(N - 1)) : s;      // jnp.linspace(0, s, N) :71
  const float k = (float)(i + 1);                                          // :72
  float kap = 0.0f, th = 0.0f, pw = 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                            // :32-41
    const float temp = c[j] * pw;
    kap = kap + temp;
    th = th + temp * sk / (float)(j + 1);
    pw = pw * sk;
  }
  float s_new, c_new, s_old, c_old;
  sincos_fast(th, s_new, c_new);
  sincos_fast(st[2], s_old, c_old);
  const float dx = st[4] * (1.0f - 1.0f / k) + (c_new + c_old) / 2.0f / k;   // :47-50
  const float dy = st[5] * (1.0f - 1.0f / k) + (s_new + s_old) / 2.0f / k;   // :51-54
  st[0] = sk * dx;
  st[1] = sk * dy;
  st[2] = th;
  st[3] = kap;
  st[4] = dx;
  st[5] = dy;
}

}  // namespace irbfn
