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
// A lane of an operand holds the slots of ONE lane group g = lane >> 4: head slots 4 g + j (j < 4), tail slots 32 hf + 8 g + j (j < 8).
// The products of coordinate i live with group i / 2 -- its head in head slot 4 (i / 2) + (i & 1), its five tail combinations in tail
// slots 32 (i & 1) + 8 (i / 2) + m -- so a lane of the query side splits just ITS two coordinates (gram_query_operands); the
// Q x alpha and 1 x c2 products fill free slots of the groups 0..2.  (Until round 3's last version the slots ran coordinate by
// coordinate across the groups: every lane split all eight coordinates and selected -- 700 instructions of prologue per wave.)
struct GramSlot { int kind, dim, p, q; };        // kind: 0 empty, 1 x'_dim part p x C part q, 2 Q part p x alpha part q, 3 1 x c2 part p
__host__ __device__ constexpr int gram_comb_p(int m) { return m == 0 ? 0 : (m == 1 ? 1 : (m == 2 ? 0 : (m == 3 ? 2 : 1))); }
__host__ __device__ constexpr int gram_comb_q(int m) { return m == 0 ? 1 : (m == 1 ? 0 : (m == 2 ? 2 : (m == 3 ? 0 : 1))); }
__host__ __device__ constexpr GramSlot gram_head_slot(int s) {
  const int g = s >> 2, j = s & 3;
  return j < 2 ? GramSlot{1, 2 * g + j, 0, 0}
               : (j == 2 ? (g == 0 ? GramSlot{2, 0, 0, 0} : (g == 1 ? GramSlot{3, 0, 0, 0} : (g == 2 ? GramSlot{3, 0, 1, 0} : GramSlot{0, 0, 0, 0})))
                         : GramSlot{0, 0, 0, 0});
}
__host__ __device__ constexpr GramSlot gram_tail_slot(int s) {
  const int hf = s >> 5, g = (s >> 3) & 3, j = s & 7;
  const int m = hf * 3 + (j - 5);                            // free slots of group 0: Q x alpha combinations 0..4, then the third c2 part
  return j < 5 ? GramSlot{1, 2 * g + hf, gram_comb_p(j), gram_comb_q(j)}
               : (g == 0 ? (m < 5 ? GramSlot{2, 0, gram_comb_p(m), gram_comb_q(m)} : GramSlot{3, 0, 2, 0})
                         : (g == 1 && hf == 0 && j == 5 ? GramSlot{3, 0, 3, 0} : GramSlot{0, 0, 0, 0}));
}
static_assert(kGramDims == 8, "four lane groups x two coordinates");
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
// plain instructions it replaces, MI355X_MICROARCH.md constants table.  v_fma_mixlo_f16 + v_fma_mixhi_f16 writing the two halves of
// the lo pair directly -- 2 instructions instead of 3 -- measured 5 % SLOWER at config 2: 92.1 vs 87.4 us, profiles/r03_gram_experiments.txt.)
__device__ __forceinline__ void split_pair_mix(float p0, float p1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  const float q0 = p0 * kLoGain, q1 = p1 * kLoGain;
  const float ng = -kLoGain;
  float d0, d1;
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
// slots 4 g + j of the head MFMA, 32 hf + 8 g + j of the tail MFMAs (g = lane >> 4).  A lane splits the two coordinates 2 g and
// 2 g + 1 of its queries (the slot tables above put exactly their products into its slots); Q = |x'|^2 is summed in double-float
// over the four lanes that hold a query's coordinates (a commutative butterfly: the four lanes end with the same bits).  Returns
// whether THIS lane's coordinates fall outside the representable box (|x'_i| >= 2^ex, Q >= 2^eq, NaN, Inf) or the header says the
// net does not fit -- the callers take the ballot of the wave.
template <int DC>
__device__ __forceinline__ bool gram_query_operands(const F16Args& a, const GramHdr* hp, const long (&qrow)[2], int g, h4_t (&bhd)[2],
                                                    h8_t (&btl)[2][2]) {
  // The error-free sums below are written operation by operation: contracted into FMAs (hipcc's default, -ffp-contract=fast:
  // th = qh + sh * sh as one fma, tb = th - qh as fma(-sh', sh', th) when qh is itself a product) they lose the low word of Q --
  // 6e-8 of |x'|^2, which the cancellation against 2 c'x' and |c'|^2 turns into 4e-5 of the result for a query 28 widths from
  // the origin of the expansion (tests/test_gpu_gram.py::test_gram_ill_conditioned_columns[outlier_far_centre])
#pragma clang fp contract(off)
  static_assert(DC <= kGramDims, "eight coordinate slots");
  const int ex = hp->ex, ec = hp->ec, eq = hp->eq, ea = hp->ea, e2 = hp->e2;
  GramHdr hx;                                                // exponents only (gram_T)
  hx.ex = ex; hx.ec = ec; hx.eq = eq; hx.ea = ea; hx.e2 = e2;
  bool bad = hp->ok == 0;
  const float xinv = __builtin_ldexpf(1.0f, -ex), qinv = __builtin_ldexpf(1.0f, -eq);
  const float xlim = __builtin_ldexpf(1.0f, ex) * 0.999f, qlim = __builtin_ldexpf(1.0f, eq) * 0.999f;
  auto scale_of = [&](const GramSlot sl) -> float { return sl.kind == 0 ? 0.0f : __builtin_ldexpf(1.0f, gram_ax(gram_T(hx, sl))); };
  float rr[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) rr[c] = (2 * g + c < DC) ? -hp->r[2 * g + c] : 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float nx[2][3], nq[3];
    float qh = 0.0f, ql = 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int i = 2 * g + c;
      const float xv = (i < DC && i < a.Dreal) ? a.x[qrow[t] * a.Dreal + i] : 0.0f;
      const float sh = xv + rr[c];                           // TwoSum: x' = sh + sl exactly
      const float bb = sh - xv;
      const float sl = (xv - (sh - bb)) + (rr[c] - bb);
      bad = bad || !(__builtin_fabsf(sh) < xlim);            // NaN / Inf / outside the box
      gram_parts_f(sh, sl, xinv, nx[c]);
      const float ph = sh * sh;                              // Q += x'^2 in double-float
      const float pl = __builtin_fmaf(sh, sh, -ph) + 2.0f * sh * sl;
      const float th = qh + ph;
      const float tb = th - qh;
      ql += ((qh - (th - tb)) + (ph - tb)) + pl;
      qh = th;
    }
#pragma unroll
    for (int off = 16; off <= 32; off *= 2) {                // the other lane groups' coordinates
      const float oh = __shfl_xor(qh, off), ol = __shfl_xor(ql, off);
      const float th = qh + oh;
      const float tb = th - qh;
      ql = ((qh - (th - tb)) + (oh - tb)) + (ql + ol);
      qh = th;
    }
    bad = bad || !(qh < qlim);
    gram_parts_f(qh, ql, qinv, nq);
    // head slots 4 g + j: the two coordinates' heads, then (group 0) Q x alpha, (groups 1, 2) the two c2 heads
    {
      const float sc = scale_of(gram_head_slot(0));          // kind 1, p = q = 0: the same weight in every group
      const float f2 = g == 0 ? nq[0] * scale_of(gram_head_slot(2))
                              : (g == 1 ? scale_of(gram_head_slot(4 + 2)) : (g == 2 ? scale_of(gram_head_slot(8 + 2)) : 0.0f));
      bhd[t][0] = (_Float16)(nx[0][0] * sc);
      bhd[t][1] = (_Float16)(nx[1][0] * sc);
      bhd[t][2] = (_Float16)f2;
      bhd[t][3] = (_Float16)0.0f;
    }
    // tail slots 32 hf + 8 g + j: j < 5 the combinations of coordinate 2 g + hf; j >= 5: group 0 the Q x alpha combinations and
    // the third c2 part, group 1 (hf = 0, j = 5) the fourth
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const GramSlot sl = gram_tail_slot(32 * hf + j);     // group 0's entry: p, q and the weight do not depend on the group
        btl[t][hf][j] = (_Float16)(nx[hf][sl.p] * scale_of(sl));
      }
#pragma unroll
      for (int j = 5; j < 8; ++j) {
        const GramSlot s0 = gram_tail_slot(32 * hf + j), s1 = gram_tail_slot(32 * hf + 8 + j);
        const float v0 = s0.kind == 2 ? nq[s0.p] * scale_of(s0) : scale_of(s0);      // kind 3 / empty: the weight itself / 0
        const float v1 = scale_of(s1);                                                // kind 3 or empty
        btl[t][hf][j] = (_Float16)(g == 0 ? v0 : (g == 1 ? v1 : 0.0f));
      }
    }
  }
  return bad;
}
// the tables say what the code above assumes
static_assert(gram_head_slot(0).kind == 1 && gram_head_slot(4 * 3 + 1).dim == 7 && gram_head_slot(2).kind == 2 && gram_head_slot(6).kind == 3 &&
              gram_head_slot(10).kind == 3 && gram_head_slot(10).p == 1 && gram_head_slot(14).kind == 0 && gram_head_slot(3).kind == 0, "head slots");
static_assert(gram_tail_slot(32 + 8 * 2 + 4).kind == 1 && gram_tail_slot(32 + 8 * 2 + 4).dim == 5 && gram_tail_slot(5).kind == 2 &&
              gram_tail_slot(32 + 6).kind == 2 && gram_tail_slot(32 + 7).kind == 3 && gram_tail_slot(32 + 7).p == 2 && gram_tail_slot(8 + 5).kind == 3 &&
              gram_tail_slot(8 + 5).p == 3 && gram_tail_slot(8 + 6).kind == 0 && gram_tail_slot(16 + 5).kind == 0 && gram_tail_slot(32 + 8 + 5).kind == 0, "tail slots");

// The exact head sums u[t][ct] = A[ct] x B[t] (v_mfma_f32_16x16x16_f16, C = 0) of the 2 x 2 tiles, as ONE asm block that ends in
// five wait states.  gfx950 does NOT interlock a 16x16x32 MFMA that accumulates onto the result of a 16x16x16 one, and hipcc does
// not pad the pair: with fewer than 5 wait states (or one other MFMA) between the two the second reads a stale accumulator --
// half of its outputs wrong, tools/probe_mfma_dependent.hip (16x16x32 -> 16x16x32, 16x16x16 -> 16x16x16 and MFMA -> VALU are
// interlocked).  The compiler's schedule had kept the pairs apart in the shipped kernels and put them back to back in faster
// variants of K2g (four waves per SIMD: all 24 instances wrong), which is how this was found.
__device__ __forceinline__ void gram_heads(const h4_t (&a)[2], const h4_t (&b)[2], f4_t (&u)[2][2]) {
  asm("v_mfma_f32_16x16x16_f16 %0, %4, %6, 0\n v_mfma_f32_16x16x16_f16 %1, %5, %6, 0\n v_mfma_f32_16x16x16_f16 %2, %4, %7, 0\n "
      "v_mfma_f32_16x16x16_f16 %3, %5, %7, 0\n s_nop 4"
      : "=&v"(u[0][0]), "=&v"(u[0][1]), "=&v"(u[1][0]), "=&v"(u[1][1])
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
}
// the same for one query half against two centre tiles (K2g): u[ct] = A x B[ct]
__device__ __forceinline__ void gram_heads2(const h4_t& a, const h4_t (&b)[2], f4_t (&u)[2]) {
  asm("v_mfma_f32_16x16x16_f16 %0, %2, %3, 0\n v_mfma_f32_16x16x16_f16 %1, %2, %4, 0\n s_nop 4"
      : "=&v"(u[0]), "=&v"(u[1])
      : "v"(a), "v"(b[0]), "v"(b[1]));
}

// ... with a start value c in the accumulator (K2g, gaussian: c = -14 takes the 2^14 of the basis value out again, so that the
// result IS alpha d^2; a multiple of the head grid and inside the budget that holds the +14 of c2, hence still exact)
__device__ __forceinline__ void gram_heads2c(const h4_t& a, const h4_t (&b)[2], const f4_t& c, f4_t (&u)[2]) {
  asm("v_mfma_f32_16x16x16_f16 %0, %2, %3, %5\n v_mfma_f32_16x16x16_f16 %1, %2, %4, %5\n s_nop 4"
      : "=&v"(u[0]), "=&v"(u[1])
      : "v"(a), "v"(b[0]), "v"(b[1]), "v"(c));
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
  gram_heads(ahd, bhd, u);                                     // exact head sums
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
