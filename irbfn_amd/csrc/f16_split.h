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
// from; below that the absolute error of a factor is <= 2^-36 of that magnitude.  The first round kept the
// residual unscaled in the same accumulator: it fell into the f16 subnormals as soon as |v| < 2^-3 of the
// maximum and the pair lost a bit per factor of two below that (3e-5 relative at 2^-10).
#pragma once

#ifndef IRBFN_F16_SPLIT
#define IRBFN_F16_SPLIT 1                        // 1: mask / subtract / scale; 3: v_fma_mix form (split_pair_f16)
#endif

namespace irbfn {

#if IRBFN_F16_SPLIT == 3
constexpr int kPhiExp = 25;                      // the basis value arrives as 2^25 phi, its hi half is 2^-11 of it
#else
constexpr int kPhiExp = 14;                      // the basis value arrives as P = 2^14 phi (<= 2^14)
#endif
constexpr float kPhiScale = (float)(1 << kPhiExp);
constexpr float kPhiInv = 1.0f / kPhiScale;
constexpr int kWExp = 15;                        // static factors (weights, cotangents): 2^15 v / s, |.| < 2^15
constexpr float kWScale = (float)(1 << kWExp);
constexpr float kLoScale = 1.0f / 2048.0f;       // A1 + kLoScale * A2
constexpr float kLoGain = 2048.0f;

// Two basis values p0, p1 (scaled by 2^kPhiExp, non-negative) -> packed f16 pairs: hi = 2^14 phi to 11 bits,
// lo = 2^11 x (2^14 phi - hi).  TERMS == 1: plain f16 operands (reduced precision, reporting only).
template <int TERMS>
__device__ __forceinline__ void split_pair_f16(float p0, float p1, unsigned& hi, unsigned& lo) {
  if constexpr (TERMS >= 2) {
#if IRBFN_F16_SPLIT == 3
    // hi = f16_rn(2^-11 p) and r = p - 2^11 hi (exact) through the mixed-precision FMAs: 5 instructions per pair
    const float c1 = 4.8828125e-04f, c2 = -2048.0f;
    unsigned hh = 0;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hh) : "v"(p0), "v"(c1));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hh) : "v"(p1), "v"(c1));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh), "v"(c2), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh), "v"(c2), "v"(p1));
    hi = hh;
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
#else
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz((p0 - h0) * kLoGain, (p1 - h1) * kLoGain));
#endif
  } else {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0 * (kPhiInv * 16384.0f), p1 * (kPhiInv * 16384.0f)));
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
