// K1: fused WCRBFNet forward for gfx950.
//
// Replaces WCRBFNet.__call__ (src/irbfn_mpc/model.py:169-198) = region gate (model.py:42-95) +
// vmapped RBFLayer (flax_rbf.py:258-285) + gamma-weighted sum over regions (model.py:193) + Dense
// (model.py:196), in ONE launch, with no [B,R,K] or [B,K] intermediate in HBM.
//
// Mapping ("query-lane"): one lane owns Q queries (x and the Q*OP output accumulators live in
// VGPRs); the NW waves of a workgroup share the same 64*Q queries and split the N = R*K centres
// among themselves.  Everything that depends on the centre only -- c[D], the folded width scale and
// the weight row W[k,:] -- is wave-uniform, so the packed centre records stream through the SCALAR
// cache into SGPRs (s_load_dwordxN) and feed the VALU as the one SGPR operand a gfx9 VALU
// instruction may carry: no LDS traffic, no per-lane loads in the hot loop.  Per (query, centre)
// pair: D v_sub + D v_fma (distance), 1 v_mul, 1 transcendental (v_exp / v_rcp / v_rsq), OP v_fma.
// The NW partial sums are combined through LDS in a fixed order (deterministic), gamma and bias
// are applied and the tile is written with coalesced stores.
#pragma once

#include "common.h"

namespace irbfn {

struct FwdArgs {
  const float* __restrict__ x;     // [B][Dreal]
  const float* __restrict__ rec;   // [N][S]
  const float* __restrict__ bias;  // [OP]
  float* __restrict__ out;         // [B][O]
  GateTables gate;
  long B;
  int Dreal, O, N, K, S, basis;
  // fused roll-out (forward_rollout only)
  const float* __restrict__ state0;
  float* __restrict__ states;
  int T, mode;
  DynParams dp;
  // planner mirror trick (irbfn_planner.py:203-204): rows with mirror[b] != 0 get outputs [sv0, O) negated
  const int* __restrict__ mirror;
  int sv0;
  // caller-provided region weights gamma[B][R] (ClusterWCRBFNet: softmax gate, model.py:341-414) instead of the
  // tanh tables; GATED kernels only
  const float* __restrict__ gamma_ext;
  int R;
};

__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float fast_rsq(float v) { return __builtin_amdgcn_rsqf(v); }

// phi from the squared distance r2 and the record's folded scale sc.
//   BC_GAUSS : sc = -a*log2(e)*exp(-2 log_sig)  -> phi = 2^(r2*sc) = exp(-a d^2)   (flax_rbf.py:35-47)
//   others   : sc = exp(-2 log_sig)             -> d^2 = r2*sc
template <int BC>
__device__ __forceinline__ float basis_from_r2(float r2, float sc, int basis) {
  const float t = r2 * sc;
  if constexpr (BC == BC_GAUSS) {
    return fast_exp2(t);
  } else if constexpr (BC == BC_IQ) {
    return fast_rcp(1.0f + t);                 // flax_rbf.py:50-52
  } else if constexpr (BC == BC_IMQ) {
    return fast_rsq(1.0f + t);                 // flax_rbf.py:73-75
  } else {
    const float d2 = t;
    const float d = sqrtf(d2);                 // flax_rbf.py:280
    switch (basis) {
      case IRBFN_LINEAR: return d;                                         // :55-57
      case IRBFN_QUADRATIC: return d2;                                     // :61-63
      case IRBFN_MULTIQUADRIC: return sqrtf(1.0f + d2);                    // :67-69
      case IRBFN_SPLINE: return d2 * logf(d + 1.0f);                       // :79-81
      case IRBFN_POISSON_ONE: return (d - 1.0f) * expf(-d);                // :85-87
      case IRBFN_POISSON_TWO: return ((d - 2.0f) / 2.0f) * d * expf(-d);   // :91-97
      case IRBFN_MATERN32: return (1.0f + 1.7320508075688772f * d) * expf(-1.7320508075688772f * d);
      case IRBFN_MATERN52:
        return (1.0f + 2.23606797749979f * d + (5.0f / 3.0f) * d2) * expf(-2.23606797749979f * d);
      default: return 0.0f;
    }
  }
}

// ---- batched transcendentals ----------------------------------------------------------------------------
// On gfx950 ONE v_exp_f32 / v_rcp_f32 / v_rsq_f32 inside a stream of plain VALU costs ~33 cycles of SIMD
// time, the same instruction issued back to back costs 8 (tools/ubench_trans.hip,
// profiles/r01_ubench_trans_hazard.txt): the basis of G centres is therefore evaluated as one block.
// basis_arg: everything in front of the transcendental; trans_block: NT of them adjacent (inline asm: the
// compiler would re-interleave them with the distance arithmetic); s_nop: trans -> VALU forwarding hazard.
template <int BC>
__device__ __forceinline__ float basis_arg(float r2, float sc) {
  if constexpr (BC == BC_GAUSS) return r2 * sc;            // phi = 2^(r2*sc)
  else return __builtin_fmaf(r2, sc, 1.0f);                // IQ: 1/(1+d2), IMQ: (1+d2)^-1/2
}

// wait states around the block: a plain VALU instruction right behind (or in front of) a transcendental
// triggers a hardware interlock that costs 10-15 cycles; explicit wait states are cheaper
// (tools/ubench_k1body.hip: 94 -> 82 cycles per centre with s_nop 7)
#ifndef IRBFN_TRANS_PRE
#define IRBFN_TRANS_PRE ""
#endif
#ifndef IRBFN_TRANS_POST
#define IRBFN_TRANS_POST "s_nop 7"
#endif
#define IRBFN_TRANS4(OP, A, B, C, E) OP " %" #A ", %" #A "\n " OP " %" #B ", %" #B "\n " OP " %" #C ", %" #C "\n " OP " %" #E ", %" #E "\n "
template <int BC, int NT>
__device__ __forceinline__ void trans_block(float (&v)[NT]) {
  static_assert(NT == 2 || NT == 4 || NT == 8 || NT == 16, "trans_block: 2, 4, 8 or 16 values");
#define IRBFN_TRANS_EMIT(OP)                                                                                   \
  if constexpr (NT == 2)                                                                                       \
    asm volatile(OP " %0, %0\n " OP " %1, %1\n s_nop 0" : "+v"(v[0]), "+v"(v[1]));                             \
  else if constexpr (NT == 4)                                                                                  \
    asm volatile(IRBFN_TRANS_PRE IRBFN_TRANS4(OP, 0, 1, 2, 3) IRBFN_TRANS_POST : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));      \
  else if constexpr (NT == 8)                                                                                  \
    asm volatile(IRBFN_TRANS_PRE IRBFN_TRANS4(OP, 0, 1, 2, 3) IRBFN_TRANS4(OP, 4, 5, 6, 7) IRBFN_TRANS_POST                            \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])); \
  else                                                                                                         \
    asm volatile(IRBFN_TRANS_PRE IRBFN_TRANS4(OP, 0, 1, 2, 3) IRBFN_TRANS4(OP, 4, 5, 6, 7) IRBFN_TRANS4(OP, 8, 9, 10, 11) \
                 IRBFN_TRANS4(OP, 12, 13, 14, 15) IRBFN_TRANS_POST                                                     \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), \
                   "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
  if constexpr (BC == BC_GAUSS) { IRBFN_TRANS_EMIT("v_exp_f32_e32") }
  else if constexpr (BC == BC_IQ) { IRBFN_TRANS_EMIT("v_rcp_f32_e32") }
  else { IRBFN_TRANS_EMIT("v_rsq_f32_e32") }
#undef IRBFN_TRANS_EMIT
}

// one factor of the smooth indicator: ((tanh(delta*(x-lo))+1)/2) * ((tanh(delta*(hi-x))+1)/2), model.py:83-85.
// (tanh(z)+1)/2 == 1/(1+exp(-2z)): evaluated in that form with v_exp_f32 / v_rcp_f32 (6 instructions instead of
// ocml's ~35-instruction tanhf; <= 3e-7 relative, and without the float32 cancellation of tanh(z)+1 near z << 0).
// FLOAT32 SEMANTICS OF THE ZERO: the reference evaluates this expression in float32 (its default), where tanh(z) IS -1
// -- and the factor exactly 0 -- for every z below -13 ln 2 = -9.0109 (a correctly rounded float32 tanh rounds to -1 once
// 2 e^{2z} < 2^-25).  The exp form alone would return 1.5e-8 ... 1e-38 there; it is flushed to the reference's exact 0
// so that "gamma == 0" means the same in every kernel: a region whose gamma is 0 contributes exactly nothing, which is
// what the region-sparse kernels (rbf_sparse.hip) skip.  A float64 run of the reference (--use_float64) keeps factors
// down to 3e-17 there: the dropped terms are below 1.5e-8 of a region's own output ("K1r").
// Saturates to exactly 1 by itself (1 + 2^-26 rounds to 1) and propagates NaN like the tanh form.
constexpr float kGateSatZ = 9.0109133f;                              // 13 ln 2
__device__ __forceinline__ float half_tanh_plus_one(float z) {
  const float v = fast_rcp(1.0f + fast_exp2(-2.8853900817779268f * z));      // 2*log2(e)
  return z < -kGateSatZ ? 0.0f : v;                                    // NaN: comparison false -> v = NaN
}
__device__ __forceinline__ float gate_factor(float xv, float lo, float hi, float delta) {
  return half_tanh_plus_one(delta * (xv - lo)) * half_tanh_plus_one(delta * (hi - xv));
}

}  // namespace irbfn
