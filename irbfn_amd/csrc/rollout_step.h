// One-step maps of the roll-outs (device functions shared by the stand-alone roll-out kernels and
// the fused forward+roll-out kernel).  One lane integrates one trajectory; state lives in VGPRs.
#pragma once

#include "common.h"

#ifndef IRBFN_ROLL_IEEE_DIV
#define IRBFN_ROLL_IEEE_DIV 0      // 1: IEEE division sequences everywhere (A/B; KAT-1 is met to 1e-6 either way)
#endif

namespace irbfn {

// Straight-line sin/cos for the roll-outs.  ocml's sinf/cosf/tanf cost ~430 VALU instructions per step with
// divergent range-reduction branches, and their inlined bodies (one copy per call site) made the unrolled step
// loops instruction-fetch bound.  Here: Cody-Waite reduction by pi/2 with three FMA steps, cephes minimax
// polynomials on [-pi/4, pi/4] (<= 1 ulp each), quadrant select; total error <= 2 ulp, one reduction serves sin
// and cos.  The FMAs keep the float32 reduction accurate while k = rint(x 2/pi) is off by at most one, i.e. for
// |x| < 2^22 (every partial remainder is O(1) and rounded once).  Beyond that (rare, divergent, a dozen f64
// instructions -- no call: a call makes hipcc drain vmcnt, i.e. wait for every outstanding store, in each step)
// the reduction runs in float64, exact for |x| < 2^52; inf / nan give nan.  Float32 angles >= 2^52 have a spacing
// >= 2^29 rad: the remainder is clamped and the result is SOME value in [-1, 1] (documented divergence).
__device__ __forceinline__ void sincos_poly(float r, int k, float& sn, float& cs) {
  const float r2 = r * r;
  float sp = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
  sp = __builtin_fmaf(sp * r2, r, r);
  float cp = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
  cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
  const float a = (k & 1) ? cp : sp;                  // sin: sp, cp, -sp, -cp
  const float b = (k & 1) ? sp : cp;                  // cos: cp, -sp, -cp, sp
  sn = (k & 2) ? -a : a;
  cs = ((k + 1) & 2) ? -b : b;
}

// |x| known to be < 2^22 (e.g. a clipped steering angle); nan in -> nan out
__device__ __forceinline__ void sincos_small(float x, float& sn, float& cs) {
  const float kf = rintf(x * 0.636619772367581343f);  // x * 2/pi
  float r = __builtin_fmaf(-kf, 1.57079625129699707031e+00f, x);
  r = __builtin_fmaf(-kf, 7.54978941586159635335e-08f, r);
  r = __builtin_fmaf(-kf, 5.39030285815811905290e-15f, r);
  sincos_poly(r, (int)kf, sn, cs);
}

__device__ __forceinline__ void sincos_fast(float x, float& sn, float& cs) {
  const float kf = rintf(x * 0.636619772367581343f);
  float r = __builtin_fmaf(-kf, 1.57079625129699707031e+00f, x);
  r = __builtin_fmaf(-kf, 7.54978941586159635335e-08f, r);
  r = __builtin_fmaf(-kf, 5.39030285815811905290e-15f, r);
  int k = (int)kf;
  if (__builtin_expect(!(fabsf(x) < 4194304.0f), 0)) {
    const double xd = (double)x;
    const double kd = __builtin_rint(xd * 0.63661977236758134308);
    double rd = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
    rd = __builtin_fma(-kd, 6.12323399573676603587e-17, rd);
    rd = rd < -0.7853981633974483 ? -0.7853981633974483 : (rd > 0.7853981633974483 ? 0.7853981633974483 : rd);   // nan stays
    r = (float)rd;
    k = (int)(kd - 4.0 * __builtin_floor(kd * 0.25));
  }
  sincos_poly(r, k, sn, cs);
}

// a / b by reciprocal + one correction step (<= 1 ulp; 4 instructions instead of the ~10 of the IEEE sequence).
// Only where b is never 0 / inf / subnormal: cos of a float (never exactly 0), the wheelbase.
__device__ __forceinline__ float fdiv_fast(float a, float b) {
#if IRBFN_ROLL_IEEE_DIV
  return a / b;
#else
  const float rc = __builtin_amdgcn_rcpf(b);
  const float q = a * rc;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), rc, q);
#endif
}

__device__ __forceinline__ float tan_fast(float x) {
  float sn, cs;
  sincos_fast(x, sn, cs);
  return fdiv_fast(sn, cs);
}
// tan of an angle clipped to [-bound, bound]; `small` = bound < 2^22 (wave-uniform)
__device__ __forceinline__ float tan_clipped(float x, bool small) {
  float sn, cs;
  if (small) sincos_small(x, sn, cs);
  else sincos_fast(x, sn, cs);
  return fdiv_fast(sn, cs);
}

__device__ __forceinline__ float clipf(float v, float lo, float hi) {
  // jnp.clip = min(max(v, lo), hi), nan propagates.  v_med3_f32 (nan in -> min3) + one select: 3 instructions
  // (fminf(fmaxf()) canonicalises every operand: 7)
  const float m = __builtin_amdgcn_fmed3f(v, lo, hi);
  return v != v ? v : m;
}

// How a step gets its trigonometry: sincos(A) of the heading-like angle and tan(B) of the (clipped) steering angle.
// TrigDirect: this lane evaluates both (one lane per trajectory).  TrigPair (rollout.hip): two adjacent lanes share
// a trajectory, the even lane evaluates sincos(A), the odd lane sincos(B) and the quotient, results swap by DPP --
// ONE polynomial evaluation per step instead of two, same values bit for bit.
struct TrigDirect {
  __device__ __forceinline__ void sincos_tan(float A, float B, bool small, float& sn, float& cs, float& tn) const {
    sincos_fast(A, sn, cs);
    tn = tan_clipped(B, small);
  }
  __device__ __forceinline__ void sincos(float A, float& sn, float& cs) const { sincos_fast(A, sn, cs); }
};

// Single-track model, src/irbfn_mpc/dynamics.py:9-91.  SELECT=true: x + select(V>3, f, f_ks)*dt (:90);
// SELECT=false: kinematic RHS only = dynamic_st_onestep_aux (:103-187).
template <bool SELECT, typename Trig = TrigDirect>
__device__ __forceinline__ void st_step(float (&s)[7], float accl_in, float sv_in, const DynParams& dp, const Trig trig = Trig()) {
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
  if constexpr (SELECT) {
    // lax.select(V > 3, f, f_ks) (:90) WITHOUT a branch.  The batch mixes fast and slow trajectories in every wave, so a
    // branch ran both right-hand sides one after the other, each with its own trigonometry (191 us against 107 us for the
    // kinematic scan at B = 262144, T = 50).  Both need cos / sin of ONE angle -- psi + beta (:50-51) or psi (:79-80) --
    // so the angle is selected first and evaluated once (with the tan of the kinematic branch beside it: the lane pair of
    // TrigPair shares the two evaluations); the few dozen flops of both branches follow as straight-line code and the
    // results are selected.  Same expressions, same order: a lane gets the bits the branch gave it (an unselected
    // PSI_DOT / V at V = 0 is inf / nan and is dropped by the select, as in the reference's own lax.select).
    const bool dyn = V > 3.0f;
    float sn, cs, tn;
    trig.sincos_tan(dyn ? PSI + BETA : PSI, DELTA, s_max < 4194304.0f, sn, cs, tn);
    f0 = V * cs;                                           // :50 / :79
    f1 = V * sn;                                           // :51 / :80
    const float glr = g * lr - ACCL * h, glf = g * lf + ACCL * h;
    const float pv = PSI_DOT / V;
    const float f5d = ((mu * m) / (I * (lf + lr))) *
                      (lf * C_Sf * glr * DELTA + (lr * C_Sr * glf - lf * C_Sf * glr) * BETA -
                       (lf * lf * C_Sf * glr + lr * lr * C_Sr * glf) * pv);                      // :54-63
    const float f6d = (mu / (V * (lr + lf))) *
                          (C_Sf * glr * DELTA - (C_Sr * glf + C_Sf * glr) * BETA +
                           (C_Sr * glf * lr - C_Sf * glr * lf) * pv) -
                      PSI_DOT;                                                                    // :64-76
    f4 = dyn ? PSI_DOT : fdiv_fast(V, lr + lf) * tn;       // :53 / :84
    f5 = dyn ? f5d : 0.0f;
    f6 = dyn ? f6d : 0.0f;
  } else {                                               // :78-88
    float sn, cs, tn;
    trig.sincos_tan(PSI, DELTA, s_max < 4194304.0f, sn, cs, tn);
    f0 = V * cs;
    f1 = V * sn;
    f4 = fdiv_fast(V, lr + lf) * tn;
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
template <typename Trig = TrigDirect>
__device__ __forceinline__ void fullint_step(float (&s)[5], float a, float dv, const Trig trig = Trig()) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float DT = 0.1f, WB = 0.33f, VMAX = 7.0f, VMIN = 0.0f, SMAX = 0.4189f;   // :307-311
  const float dnew = clipf(s[2] + dv * DT, -SMAX, SMAX); // :362-363
  float sn, cs, tn;
  trig.sincos_tan(s[4], dnew, true, sn, cs, tn);
  s[0] = s[0] + s[3] * cs * DT;                          // :360
  s[1] = s[1] + s[3] * sn * DT;                          // :361
  s[2] = dnew;
  s[3] = clipf(s[3] + a * DT, VMIN, VMAX);               // :364-365
  s[4] = s[4] + fdiv_fast(s[3], WB) * tn * DT;           // :366
}

// Frenet model, low-speed RHS only, src/irbfn_mpc/dynamics.py:190-281 (:267-280).
template <typename Trig = TrigDirect>
__device__ __forceinline__ void frenet_step(float (&s)[8], float a_in, float dv_in, const DynParams& dp, const Trig trig = Trig()) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float LF = dp.p[3], LR = dp.p[4], dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10],
              s_max = dp.p[11];
  const float ey = s[1], delta = clipf(s[2], -s_max, s_max), vx = s[3], epsi = s[6], cur = s[7];
  const float a = clipf(a_in, -a_max, a_max);            // :235
  const float dv = clipf(dv_in, -sv_max, sv_max);        // :236
  float se, ce, td;
  trig.sincos_tan(epsi, delta, s_max < 4194304.0f, se, ce, td);
  const float d0 = (vx * ce) / (1.0f - ey * cur);        // :268
  const float d1 = vx * se;                              // :269
  const float d6 = fdiv_fast(vx * td, LR + LF) - cur * ((vx * ce) / (1.0f - cur * ey));  // :274-275
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

// integrate_one_step (planner_utils.py:44-59); st = [x, y, theta, kappa, dx, dy]; i = 0-based sample.
// sc: (sin, cos) of st[2] on entry, of the new theta on exit.  The step needs cos / sin of the new AND the old heading
// (:47-54); the old one is the previous step's new one, so a caller that walks the samples in order carries the pair instead
// of evaluating it again (sin, cos of theta = 0, the scan's initial heading, are exactly (0, 1)): same bits, half the
// trigonometry.
// Divisions: the reference divides by (j + 1), by k and by 2 (:36-41, :47-54).  IEEE division sequences (~12 instructions
// each, ten per sample) made the spiral kernels compute-bound (forward 2.3, VJP 0.7-1.0 TB/s at N = 100); here: exact
// constants' reciprocals for (j + 1) and 2 (<= 1 ulp each), reciprocal + one correction step for k and N - 1 (fdiv_fast,
// <= 1 ulp).  Results move by a few ulp against the float32 restatement; the tests hold them to 1e-5 of the float64 one.
__device__ __forceinline__ void spiral_step(float (&st)[6], const float (&c)[4], float s, int i, int N, float (&sc)[2]) {
#pragma clang fp contract(off)   // same mul/add sequence in every kernel that inlines this (fused == stand-alone)
  const float sk = (i < N - 1) ? s * fdiv_fast((float)i, (float)(N - 1)) : s;   // jnp.linspace(0, s, N) :71
  const float k = (float)(i + 1);                                          // :72
  const float rk = fdiv_fast(1.0f, k);
  const float rj[4] = {1.0f, 0.5f, 1.0f / 3.0f, 0.25f};
  float kap = 0.0f, th = 0.0f, pw = 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                            // :32-41
    const float temp = c[j] * pw;
    kap = kap + temp;
    th = th + temp * sk * rj[j];
    pw = pw * sk;
  }
  float s_new, c_new;
  const float s_old = sc[0], c_old = sc[1];
  sincos_fast(th, s_new, c_new);
  const float dx = st[4] * (1.0f - rk) + (c_new + c_old) * 0.5f * rk;      // :47-50
  const float dy = st[5] * (1.0f - rk) + (s_new + s_old) * 0.5f * rk;      // :51-54
  st[0] = sk * dx;
  st[1] = sk * dy;
  st[2] = th;
  st[3] = kap;
  st[4] = dx;
  st[5] = dy;
  sc[0] = s_new;
  sc[1] = c_new;
}
__device__ __forceinline__ void spiral_step(float (&st)[6], const float (&c)[4], float s, int i, int N) {
  float sc[2];
  sincos_fast(st[2], sc[0], sc[1]);
  spiral_step(st, c, s, i, N, sc);
}

}  // namespace irbfn
