// Hand-derived one-step adjoints of the roll-out models (reverse-mode of rollout_step.h), shared by the roll-out VJP
// kernels (rollout_vjp.hip) and the loss-seed kernels of the training steps (train_step.hip).  The NumPy statement
// of the same adjoints is oracle/hand_vjp.py (checked against torch.autograd of the restatement to 1e-12).
// clip() gradient: 1 strictly inside, 0 strictly outside, `tie` on a bound (jnp.clip is minimum(maximum(.)), whose
// tie rule is 1/2).
#pragma once

#include "rollout_step.h"

namespace irbfn {

__device__ __forceinline__ float clipgrad(float v, float lo, float hi, float tie) {
  return (v > lo && v < hi) ? 1.0f : ((v == lo || v == hi) ? tie : 0.0f);
}

template <int MODE> struct VjpTraits;
template <> struct VjpTraits<IRBFN_ROLLOUT_ST_KS> { static constexpr int S = 7, S0 = 7, NP = 3; };
template <> struct VjpTraits<IRBFN_ROLLOUT_FULLINT> { static constexpr int S = 5, S0 = 1, NP = 3; };
template <> struct VjpTraits<IRBFN_ROLLOUT_FRENET_LS> { static constexpr int S = 8, S0 = 8, NP = 4; };

template <int MODE>
__device__ __forceinline__ void vjp_park(const float* s, float* p) {
  if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) { p[0] = s[1]; p[1] = s[2]; p[2] = s[3]; p[3] = s[6]; }
  else { p[0] = s[2]; p[1] = s[3]; p[2] = s[4]; }
}

// one reverse step: lam already holds the seeds of this step's output state; returns d/d(a_t), d/d(sv_t)
// Trig: how the step gets sin / cos of the heading and tan of the steering angle (rollout_step.h: TrigDirect, or the lane pair
// of rollout_pair.h -- same values bit for bit)
template <int MODE, typename Trig = TrigDirect>
__device__ __forceinline__ void vjp_back_step(const float* p, float a_in, float sv_in, float* lam, float cur, float tie,
                                              const DynParams& dp, float& ga, float& gsv, const Trig trig = Trig()) {
  if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) {
    const float lf = dp.p[3], lr = dp.p[4], dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10], s_max = dp.p[11], v_max = dp.p[12];
    const float Lw = lr + lf;
    const float d_raw = p[0], v_raw = p[1], psi = p[2];
    const float DELTA = clipf(d_raw, -s_max, s_max), V = clipf(v_raw, -v_max, v_max);
    const float md = clipgrad(d_raw, -s_max, s_max, tie), mv = clipgrad(v_raw, -v_max, v_max, tie);
    const float ma = clipgrad(a_in, -a_max, a_max, tie), ms = clipgrad(sv_in, -sv_max, sv_max, tie);
    float cp, sp, td;
    trig.sincos_tan(psi, DELTA, s_max < 4194304.0f, sp, cp, td);
    ga = ma * dt * lam[3];
    gsv = ms * dt * lam[2];
    const float l2 = lam[2] + md * lam[4] * (V / Lw) * (1.0f + td * td) * dt;
    const float l3 = lam[3] + mv * dt * (lam[0] * cp + lam[1] * sp + lam[4] * td / Lw);
    const float l4 = lam[4] + dt * V * (-lam[0] * sp + lam[1] * cp);
    lam[2] = l2; lam[3] = l3; lam[4] = l4;
  } else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    const float DT = 0.1f, WB = 0.33f, VMAX = 7.0f, VMIN = 0.0f, SMAX = 0.4189f;
    const float d0 = p[0], v0 = p[1], psi = p[2];
    const float dpre = d0 + sv_in * DT, vpre = v0 + a_in * DT;
    const float d1 = clipf(dpre, -SMAX, SMAX), v1 = clipf(vpre, VMIN, VMAX);
    const float md = clipgrad(dpre, -SMAX, SMAX, tie), mv = clipgrad(vpre, VMIN, VMAX, tie);
    float cp, sp, td;
    trig.sincos_tan(psi, d1, true, sp, cp, td);
    const float Ld = lam[2] + lam[4] * (v1 / WB) * (1.0f + td * td) * DT;   // cotangent on delta'
    const float Lv = lam[3] + lam[4] * td * DT / WB;                        // cotangent on v'
    ga = mv * Lv * DT;
    gsv = md * Ld * DT;
    const float l2 = md * Ld;
    const float l3 = mv * Lv + DT * (lam[0] * cp + lam[1] * sp);
    const float l4 = lam[4] + DT * v0 * (-lam[0] * sp + lam[1] * cp);
    lam[2] = l2; lam[3] = l3; lam[4] = l4;
  } else {
    const float LF = dp.p[3], LR = dp.p[4], dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10], s_max = dp.p[11];
    const float Lw = LR + LF;
    const float ey = p[0], d_raw = p[1], vx = p[2], epsi = p[3];
    const float dc = clipf(d_raw, -s_max, s_max);
    const float md = clipgrad(d_raw, -s_max, s_max, tie);
    const float ma = clipgrad(a_in, -a_max, a_max, tie), ms = clipgrad(sv_in, -sv_max, sv_max, tie);
    float ce, se, td;
    trig.sincos_tan(epsi, dc, s_max < 4194304.0f, se, ce, td);
    const float den = 1.0f - ey * cur;
    const float d0 = vx * ce / den;
    const float A = lam[0] * dt - lam[6] * dt * cur;      // total cotangent on d0
    ga = ma * dt * lam[3];
    gsv = ms * dt * lam[2];
    const float l1 = lam[1] + A * (vx * ce * cur / (den * den));
    const float l2 = lam[2] + md * lam[6] * dt * vx * (1.0f + td * td) / Lw;
    const float l3 = lam[3] + A * ce / den + lam[1] * dt * se + lam[6] * dt * td / Lw;
    const float l6 = lam[6] + A * (-vx * se / den) + lam[1] * dt * vx * ce;
    const float l7 = lam[7] + A * (vx * ce * ey / (den * den)) - lam[6] * dt * d0;
    lam[1] = l1; lam[2] = l2; lam[3] = l3; lam[6] = l6; lam[7] = l7;
  }
}

}  // namespace irbfn
