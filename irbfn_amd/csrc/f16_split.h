// (hi, lo) f16 operand pairs for the matrix-core kernels K1h (rbf_forward_f16.hip) and K2h (rbf_vjp_f16.hip).
//
// A float32 factor v is handed to v_mfma_f32_16x16x{16,32}_f16 as TWO f16 numbers
//     vh  = the leading ~11 significant bits of 2^E v
//     vls = f16(2^11 (2^E v - vh))                 <- the residual, PRE-SCALED by 2^11
// and the three significant products of (ah + al)(bh + bl) are accumulated in TWO f32 accumulators,
//     A1 += ah * bh          A2 += als * bh + ah * bls          a*b = 2^-(Ea+Eb) (A1 + 2^-11 A2),
// (f16 x f16 products are exact in f32; al*bl, < 2^-22 relative, is dropped).  Because the residual is scaled back
// into the range of its hi half, it is a NORMAL f16 number wherever the hi half is one: every factor keeps ~22
// significant bits over the whole f16 exponent range -- 2^-14 <= |2^E v| < 2^16 -- i.e. for the scales used here
// (E chosen so that max |2^E v| is in [2^14, 2^15)) down to 2^-29 of the largest magnitude the scale was taken
// from; below that the hi half itself is an f16 subnormal and the absolute error of a factor is <= 2^-39 of that
// magnitude (for the basis value: float32-grade terms for phi >= 2^-28, absolute error <= 2^-38 per unit weight
// below -- measured: tools/probe_single_term.py, profiles/r02_f16_pair_precision.txt).  The first round kept the
// residual unscaled in the same accumulator: it fell into the f16 subnormals as soon as |v| < 2^-3 of the
// maximum and the pair lost a bit per factor of two below that (3e-5 relative at 2^-10).
#pragma once

namespace irbfn {

constexpr int kPhiExp = 14;                      // the basis value arrives as P = 2^14 phi (<= 2^14)
constexpr float kPhiScale = (float)(1 << kPhiExp);
constexpr float kPhiInv = 1.0f / kPhiScale;
constexpr int kWExp = 15;                        // static factors (weights, cotangents): 2^15 v / s, |.| < 2^15
constexpr float kWScale = (float)(1 << kWExp);
constexpr float kLoScale = 1.0f / 2048.0f;       // A1 + kLoScale * A2
constexpr float kLoGain = 2048.0f;

// Two basis values p0, p1 (scaled by 2^kPhiExp, non-negative) -> packed f16 pairs: hi = 2^14 phi to 11 bits,
// lo = 2^11 x (2^14 phi - hi) (LOS) or the unscaled residual (!LOS: one v_mul per value cheaper; the pair then
// keeps ~22 bits only for phi >= 2^-17 and carries an absolute error <= 2^-38 below -- the lo product goes into
// A1).  TERMS == 1: plain f16 operands (reduced precision, reporting only).
template <int TERMS, bool LOS = true>
__device__ __forceinline__ void split_pair_f16(float p0, float p1, unsigned& hi, unsigned& lo) {
  if constexpr (TERMS >= 2 && !LOS) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1));
  } else if constexpr (TERMS >= 2) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((p0 - h0) * kLoGain, (p1 - h1) * kLoGain));
  } else {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
    lo = 0u;
  }
}

// Static factor v with |v| <= 1 (a weight over its column scale, a cotangent over the batch scale), pack time.
__device__ __forceinline__ void split_static_f16(float v, _Float16& hi, _Float16& lo) {
  const float w = v * kWScale;                   // exact (power of two), |w| <= 2^15
  hi = (_Float16)w;                              // round to nearest
  lo = (_Float16)((w - (float)hi) * kLoGain);    // residual exact in f32, |.| <= 2^14
}

}  // namespace irbfn
