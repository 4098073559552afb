// K1g -- shared pieces of the matrix-core forward kernels that evaluate the squared distances as a Gram expansion
// (rbf_forward_gram.hip: narrow nets, the description of the scheme; rbf_forward_gram_wide.h: 16 < O <= 128): header of the
// expansion, slot tables, the query-side operands, the distance MFMAs, the VALU fallback, the 3-instruction operand split.
#pragma once

#include "rbf_forward_f16_narrow.h"

namespace irbfn {

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

constexpr int kGramHeadBytes = 2 * 64 * 8;                       // [ct][lane] 4 halfs
constexpr int kGramTailBytes = 2 * 2 * 64 * 16;                  // [ct][half][lane] 8 halfs
constexpr int kGramOpBytes = kGramHeadBytes + kGramTailBytes;                     // 5 KiB of distance operands per chunk
constexpr int gram_chunk_bytes(int NT = 1) { return kGramOpBytes + NT * 2 * kF16WBytes; }   // + W hi, W lo per column tile
constexpr int kGramChunkBytes = gram_chunk_bytes(1);                              // 7 KiB (narrow nets)
constexpr int kGramDims = 8;                                    // coordinates the slot tables hold (d <= 8: the Frenet nets have 8)
#ifndef IRBFN_GRAM_RING
#define IRBFN_GRAM_RING 5       // chunk images per centre slice in the LDS ring of the narrow kernel: 3 (a barrier per chunk) or 5 (one per two)
#endif
constexpr int kGramRing = IRBFN_GRAM_RING;
#ifndef IRBFN_GRAM_WAVES
#define IRBFN_GRAM_WAVES 4     // waves per SIMD the register allocation must allow
#endif

struct GramHdr {
  float r[8];                 // origin (midpoint of the centres' box)
  int ex, ec, eq, ea, e2;     // |x'_i| < 2^ex, |C| < 2^ec, Q < 2^eq, |alpha| < 2^ea, |c2| < 2^e2
  int ok;
  float cabs, amax, cmax, c2max;
};

// ---- the slot tables: which product sits in which k-slot ---------------------------------------------------------
struct GramSlot { int kind, dim, p, q; };        // kind: 0 empty, 1 x'_dim part p x C part q, 2 Q part p x alpha part q, 3 1 x c2 part p
__host__ __device__ constexpr GramSlot gram_head_slot(int s) {
  return s < kGramDims ? GramSlot{1, s, 0, 0}
                       : (s == kGramDims ? GramSlot{2, 0, 0, 0}
                                         : (s == kGramDims + 1 ? GramSlot{3, 0, 0, 0} : (s == kGramDims + 2 ? GramSlot{3, 0, 1, 0} : GramSlot{0, 0, 0, 0})));
}
__host__ __device__ constexpr int gram_comb_p(int m) { return m == 0 ? 0 : (m == 1 ? 1 : (m == 2 ? 0 : (m == 3 ? 2 : 1))); }
__host__ __device__ constexpr int gram_comb_q(int m) { return m == 0 ? 1 : (m == 1 ? 0 : (m == 2 ? 2 : (m == 3 ? 0 : 1))); }
__host__ __device__ constexpr GramSlot gram_tail_slot(int s) {
  return s < 5 * kGramDims ? GramSlot{1, s / 5, gram_comb_p(s % 5), gram_comb_q(s % 5)}
                           : (s < 5 * kGramDims + 5 ? GramSlot{2, 0, gram_comb_p(s - 5 * kGramDims), gram_comb_q(s - 5 * kGramDims)}
                                                    : (s < 5 * kGramDims + 7 ? GramSlot{3, 0, s - 5 * kGramDims - 5 + 2, 0} : GramSlot{0, 0, 0, 0}));
}
// power-of-two weight of a slot's product and its split between the two operands (both kept near 2^(T/2))
__host__ __device__ inline int gram_T(const GramHdr& h, const GramSlot sl) {
  return sl.kind == 1 ? h.ex + h.ec - 11 * (sl.p + sl.q) : (sl.kind == 2 ? h.eq + h.ea - 11 * (sl.p + sl.q) : h.e2 - 11 * sl.p);
}
__host__ __device__ inline int gram_ax(int T) { return (T + 1) >> 1; }      // query-side exponent; centre side: T - ax

// The inverse multiquadric arrives as P = 2^7 phi here (K1h: 2^14 phi, argument 2^-28 (1 + t)): an argument scaled by 2^-28
// would push the tail operands of the expansion -- 2^-11 and 2^-22 of the heads -- below the f16 normal range.  2^7 phi is a
// normal f16 number down to phi = 2^-21, which an algebraically decaying basis does not reach.
template <int BC>
__host__ __device__ constexpr float gram_phi_scale() { return BC == BC_IMQ ? 128.0f : kPhiScale; }

// Diagnosis build only (tools/build_variant.py ... -DIRBFN_GRAM_STAMPS): wave 0 of blocks 0 and 1 add up s_memtime per
// phase of the step; no output depends on it and the regular build contains none of it.
#ifdef IRBFN_GRAM_STAMPS
__device__ unsigned long long g_gram_stamps[32];
#define IRBFN_GRAM_T() __builtin_amdgcn_s_memtime()
#else
#define IRBFN_GRAM_T() 0ull
#endif

// ---- kernel ----------------------------------------------------------------------------------------------
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

// The (hi, lo) pair of f16_split.h in three instructions per value instead of four: hi = the packed round-toward-zero
// conversion itself (the leading 11 bits of P wherever P is a normal f16 number), lo = 2^11 (P - hi) as one v_fma_mix_f32
// that reads hi as the f16 number it is: fma(hi, -2^11, 2^11 P) is exact (every term a multiple of the last bit of P).
// (v_pk_add_f32 / v_pk_mul_f32 for the subtraction and the gain: measured slower -- packed f32 VALU costs more than the two
// plain instructions it replaces, MI355X_MICROARCH.md constants table.)
__device__ __forceinline__ void split_pair_mix(float p0, float p1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  const float q0 = p0 * kLoGain, q1 = p1 * kLoGain;
  float d0, d1;
  const float ng = -kLoGain;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "s"(ng), "v"(q0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "s"(ng), "v"(q1));
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
}

// the step's 16 transcendentals back to back (rbf_forward.h, trans_block), from the MFMA result registers into fresh ones
template <int BC>
__device__ __forceinline__ void trans16(const f4_t (&u)[2][2], float (&o)[16]) {
#define IRBFN_T8(OP, T)                                                                                                          \
  asm volatile(OP " %0, %8\n " OP " %1, %9\n " OP " %2, %10\n " OP " %3, %11\n " OP " %4, %12\n " OP " %5, %13\n " OP " %6, %14\n " OP     \
               " %7, %15" IRBFN_T8_TAIL##T                                                                                       \
               : "=&v"(o[8 * T]), "=&v"(o[8 * T + 1]), "=&v"(o[8 * T + 2]), "=&v"(o[8 * T + 3]), "=&v"(o[8 * T + 4]),           \
                 "=&v"(o[8 * T + 5]), "=&v"(o[8 * T + 6]), "=&v"(o[8 * T + 7])                                                   \
               : "v"(u[T][0][0]), "v"(u[T][0][1]), "v"(u[T][0][2]), "v"(u[T][0][3]), "v"(u[T][1][0]), "v"(u[T][1][1]),           \
                 "v"(u[T][1][2]), "v"(u[T][1][3]));
#define IRBFN_T8_TAIL0 ""
#define IRBFN_T8_TAIL1 "\n s_nop 7"
  if constexpr (BC == BC_GAUSS) { IRBFN_T8("v_exp_f32_e32", 0) IRBFN_T8("v_exp_f32_e32", 1) }
  else if constexpr (BC == BC_IQ) { IRBFN_T8("v_rcp_f32_e32", 0) IRBFN_T8("v_rcp_f32_e32", 1) }
  else { IRBFN_T8("v_rsq_f32_e32", 0) IRBFN_T8("v_rsq_f32_e32", 1) }
#undef IRBFN_T8
#undef IRBFN_T8_TAIL0
#undef IRBFN_T8_TAIL1
}

struct GramArgs {
  F16Args f;                              // x, img = K1h's image (records of the VALU path), oscale, bias, out, gate, B, ...
  const unsigned char* __restrict__ gimg; // [nchunks][kGramChunkBytes]
  const GramHdr* __restrict__ hdr;
};

// |v| < 2^E given as vh + vl -> normalised parts (float), as gram_parts_d
__device__ __forceinline__ void gram_parts_f(float vh, float vl, float inv, float (&n)[3]) {
  const float a = vh * inv, b = vl * inv;                    // exact (power of two)
  n[0] = __builtin_rintf(a * 1024.0f) * (1.0f / 1024.0f);
  const float r1 = ((a - n[0]) + b) * 2048.0f;               // a - n0 is exact
  n[1] = (float)(_Float16)r1;
  const float r2 = (r1 - n[1]) * 2048.0f;
  n[2] = (float)(_Float16)r2;
}

// Query-side operands of the expansion for the wave's two query tiles (rows qrow[t] of x): B[k = slot][column = query (lane & 15)],
// slots 4 g + j of the head MFMA, 32 hf + 8 g + j of the tail MFMAs (g = lane >> 4).  Returns whether THIS lane's queries fall
// outside the representable box (|x'_i| >= 2^ex, Q >= 2^eq, NaN, Inf) or the header says the net does not fit.
template <int DC>
__device__ __forceinline__ bool gram_query_operands(const F16Args& a, const GramHdr* hp, const long (&qrow)[2], int g, h4_t (&bhd)[2],
                                                    h8_t (&btl)[2][2]) {
  static_assert(DC <= kGramDims, "eight coordinate slots");
  const int ex = hp->ex, ec = hp->ec, eq = hp->eq, ea = hp->ea, e2 = hp->e2;
  GramHdr hx;                                                // exponents only (gram_T)
  hx.ex = ex; hx.ec = ec; hx.eq = eq; hx.ea = ea; hx.e2 = e2;
  bool bad = hp->ok == 0;
  {
    const float xinv = __builtin_ldexpf(1.0f, -ex), qinv = __builtin_ldexpf(1.0f, -eq);
    const float xlim = __builtin_ldexpf(1.0f, ex) * 0.999f, qlim = __builtin_ldexpf(1.0f, eq) * 0.999f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float nx[kGramDims][3], nq[3];
      float qh = 0.0f, ql = 0.0f;
#pragma unroll
      for (int i = 0; i < kGramDims; ++i) {
        if (i < DC) {
          const float xv = i < a.Dreal ? a.x[qrow[t] * a.Dreal + i] : 0.0f;
          const float rr = -hp->r[i];
          const float sh = xv + rr;                          // TwoSum: x' = sh + sl exactly
          const float bb = sh - xv;
          const float sl = (xv - (sh - bb)) + (rr - bb);
          bad = bad || !(__builtin_fabsf(sh) < xlim);        // NaN / Inf / outside the box
          gram_parts_f(sh, sl, xinv, nx[i]);
          const float ph = sh * sh;                          // Q += x'^2 in double-float
          const float pl = __builtin_fmaf(sh, sh, -ph) + 2.0f * sh * sl;
          const float th = qh + ph;
          const float tb = th - qh;
          ql += ((qh - (th - tb)) + (ph - tb)) + pl;
          qh = th;
        } else {
          nx[i][0] = nx[i][1] = nx[i][2] = 0.0f;
        }
      }
      bad = bad || !(qh < qlim);
      gram_parts_f(qh, ql, qinv, nq);
      auto xval = [&](const GramSlot sl) -> float {
        if (sl.kind == 0) return 0.0f;
        const float sc = __builtin_ldexpf(1.0f, gram_ax(gram_T(hx, sl)));
        return sl.kind == 1 ? nx[sl.dim][sl.p] * sc : (sl.kind == 2 ? nq[sl.p] * sc : sc);
      };
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = xval(gram_head_slot(j)), v1 = xval(gram_head_slot(4 + j)), v2 = xval(gram_head_slot(8 + j)),
                    v3 = xval(gram_head_slot(12 + j));
        bhd[t][j] = (_Float16)(g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3)));
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v0 = xval(gram_tail_slot(32 * hf + j)), v1 = xval(gram_tail_slot(32 * hf + 8 + j)),
                      v2 = xval(gram_tail_slot(32 * hf + 16 + j)), v3 = xval(gram_tail_slot(32 * hf + 24 + j));
          btl[t][hf][j] = (_Float16)(g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3)));
        }
    }
  }
  return bad;
}

// the argument of the transcendental for the 2 x 2 tiles (query tile t, centre tile ct) of the chunk image at `buf`: head sum
// (exact), then the tails
__device__ __forceinline__ void gram_distances(const unsigned char* buf, int lane, const h4_t (&bhd)[2], const h8_t (&btl)[2][2],
                                               f4_t (&u)[2][2]) {
  h4_t ahd[2];
  h8_t atl[2][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    ahd[ct] = *reinterpret_cast<const h4_t*>(buf + ct * 512 + lane * 8);
    atl[ct][0] = *reinterpret_cast<const h8_t*>(buf + kGramHeadBytes + (ct * 2 + 0) * 1024 + lane * 16);
    atl[ct][1] = *reinterpret_cast<const h8_t*>(buf + kGramHeadBytes + (ct * 2 + 1) * 1024 + lane * 16);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
      u[t][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahd[ct], bhd[t], f4_t{0, 0, 0, 0}, 0, 0, 0);   // exact head sum
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        u[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(atl[ct][hf], btl[t][hf], u[t][ct], 0, 0, 0);
}

// The same arguments on the VALU from K1h's centre records `recs` of the chunk (a wave with a query outside the box): t16[t * 8 + j] for
// centre 16 (j >> 2) + 4 g + (j & 3) -- the k order of the Phi x W product here
template <int DC, int BC>
__device__ __forceinline__ void gram_valu_args(const F16Args& a, const long (&qrow)[2], int g, const float* recs, float (&t16)[16]) {
  constexpr int RF = f16_rf(DC);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float xq[DC];
#pragma unroll
    for (int d = 0; d < DC; ++d) xq[d] = d < a.Dreal ? a.x[qrow[t] * a.Dreal + d] : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* rp = recs + ((j >> 2) * 16 + 4 * g + (j & 3)) * RF;
      float r2 = 0.0f;
#pragma unroll
      for (int d = 0; d < DC; ++d) {
        const float df = xq[d] - rp[d];                      // flax_rbf.py:280
        r2 = __builtin_fmaf(df, df, r2);
      }
      float arg = f16_arg<BC>(r2, rp[RF - 1]);
      if constexpr (BC == BC_IMQ) arg *= kPhiScale;          // 2^7 phi here (gram_phi_scale), K1h's records are scaled for 2^14 phi
      t16[t * 8 + j] = arg;
    }
  }
}

// host side (rbf_forward_gram.hip, rbf_forward_gram_wide.hip, plan_tick_wide.hip)
void gram_wide_geometry(const irbfn_net* net, int64_t B, int* SW, int* QG);
size_t gram_wide_lds_bytes(const irbfn_net* net, int SW, int QG, size_t extra_red_floats);
void gram_fill_args(const irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, GramArgs* a);
int launch_forward_gram_wide(irbfn_net* net, const float* x, float* out, int64_t B, int SW, int QG, hipStream_t s);

}  // namespace irbfn
