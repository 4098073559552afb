// K2g: parameter VJP of the single-region RBF net (jax.value_and_grad at scripts/train_nmpc.py:297-298) with
// ALL of its GEMM-shaped pieces on the f16 matrix cores at float32 accuracy:
//     hbar[q,k] = sum_o gamma_q g[q,o] W[k,o]                 (rows = queries, cols = centres, k = outputs)           as K2h
//     u[q,k]    = alpha_k |x_q - c_k|^2 + beta                (the exactly-cancelling Gram expansion of K1g, rbf_forward_gram.hip:
//                                                              rows = queries, cols = centres, k = slots -- the layout of hbar)
//     dW[k,o]   = sum_q gamma_q phi[q,k] g[q,o]               (rows = outputs, cols = centres, k = 32 queries)         as K2h
//     dC[k,i]   = sum_q tt[q,k] x'_qi,  stt[k] = sum_q tt[q,k] (rows = centres, cols = the coordinates + a column of ones, k = 32 queries)
// with tt = hbar * gamma * dphi/dd2 and   d centers[k,i] = -2/sigma_k^2 (dC[k,i] - c'_ki stt[k]),
//                                         d log_sigs[k]  = -2/sigma_k^2 sum_q tt[q,k] d2[q,k]   (d2 recovered from u per pair).
// K2h (rbf_vjp_f16.hip) computes u and the centre gradients on the VALU: 15 + 7 of its ~30 instructions per (query, centre)
// pair; here the VALU keeps the transcendental, tt, one FMA for d log_sigs and two 3-instruction (hi, lo) operand splits.
//
// Structure: a block = 4 waves = 4 chunks of 32 centres (the chunk images of K1g's pack supply the centre-side operands of u:
// same per-lane content, B operand here); the 4 waves walk the SAME slice of 32-query blocks and share an LDS ring of three block
// images (12 KiB each: u's query-side operands, g as hbar / dW operands, x' as dC operand) filled by LDS-DMA, one barrier per
// block -- K2h's waves stream the blocks privately, which at this image size would exceed the LDS-DMA rate.  A lane (g = l >> 4,
// n = l & 15) holds centre n of each of its wave's two 16-centre tiles and, per query block, the 8 queries 16 (j >> 2) + 4 g +
// (j & 3): the rows the 16x16 MFMAs return to it and, read as k = 8 g + j, valid operands of the 16x16x32 dW / dC products.
// Slabs part[QSB][V][Npad] in K2's format: vjp_reduce_kernel and the bias column sums are shared.
//
// A query outside K1g's representable box (rbf_forward_gram.h) cannot be expanded: the pre-pass raises a flag, this kernel
// returns at once and K2h -- launched behind it with the complementary test -- does the work.
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "rbf_forward_gram.h"
#include "rbf_vjp_f16.h"

namespace irbfn {

constexpr int kVgDist = 2 * (512 + 2 * 1024);            // per half: head [lane] 8 B, two tails [lane] 16 B
constexpr int kVgGA = 1024 + 2048;                       // hbar A operands: [s][lane] 8 B (hi, k = 16) + [s][lane] 16 B ((lo | hi), k = 32)
constexpr int kVgBlock = kVgDist + kVgGA + 2 * 2048;     // + gT [part][lane] 16 B, xB [part][lane] 16 B = 12 KiB
constexpr int kVgPieces = kVgBlock / 1024;
constexpr int kVgOC = 10;                                // up to this many outputs hbar is ONE 16x16x32 MFMA (3 x 10 <= 32 slots)
// hbar's operands carry 2^-33 of the factor 2^-ET between hbar and tt themselves: the cotangent side 2^-16 (|operand| <= 1/2), the
// weight side 2^-17 (<= 1/4) on top of the 2^15 of split_static_f16 -- small entries sink into the f16 subnormals, which the
// matrix cores honour: a (hi, lo) pair then keeps 2^-36 ABSOLUTE against operands of size 2^-2, beyond float32.  The gaussian
// (ET = 33) needs no multiplication of hbar at all any more, the other bases one by 2^-(ET - 33).
// (The gaussian's basis value is taken as phi itself here -- see kVgShift in the kernel -- so its factor is 2^-19: 2^-9 and 2^-10.)
__host__ __device__ constexpr int vg_exp_a(int bc) { return bc == BC_GAUSS ? 9 : 16; }
__host__ __device__ constexpr int vg_exp_b(int bc) { return bc == BC_GAUSS ? 10 : 17; }
constexpr float kVgCrossA = 1.0f / 32.0f, kVgCrossB = 1.0f / 64.0f;      // kVgCrossA * kVgCrossB = kLoScale (2^-11)
static_assert(kVgCrossA * kVgCrossB == kLoScale, "cross terms of the (hi, lo) scheme");

__device__ __forceinline__ float vg_pow2_ceil_scale(float mx) {
  if (!(mx > 0.0f) || !(mx < 3.0e38f)) return 1.0f;          // zero / Inf / NaN: unscaled
  int e;
  (void)frexpf(mx, &e);
  return ldexpf(1.0f, e);
}

// ---- pre-pass: one block of three waves per 32-query block: wave 0 the query-side operands of u, wave 1 the A operands of hbar,
// wave 2 the dW / dC operands -- three dependent chains side by side instead of one long one (one wave per block: 20 us at config 3;
// now 17-19 us.  Ablation, roles switched off one by one: role 0 alone 8.4 us, role 1 alone 11.7, role 2 alone 9.9, all three 18.6 --
// 6144 waves of > 64 VGPRs are a round and a half of the chip, so the chains overlap only in part) ----------------------------
template <int DC>
__global__ __launch_bounds__(192) void vjp_pack_blocks_gram_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                                  const float* __restrict__ bmax, int nbmax,
                                                                  const float* __restrict__ oscale, const GramHdr* __restrict__ hdr,
                                                                  unsigned char* __restrict__ qblk, float* __restrict__ scales,
                                                                  int* __restrict__ flag, int gen, GateTables gt, long B, int D, int O, int expA) {
  const int lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long q0 = (long)blockIdx.x * 32;
  unsigned char* p = qblk + (size_t)blockIdx.x * kVgBlock;
  // query-side operands of u for the halves s = 0, 1 (queries q0 + 16 s + n): K1g's, as A operands here
  if (role == 0) {
    F16Args a;
    a.x = x; a.B = B; a.Dreal = D;
    long qrow[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      long q = q0 + 16 * s + n;
      q = q < B ? q : B - 1;
      qrow[s] = q < 0 ? 0 : q;
    }
    h4_t bhd[2];
    h8_t btl[2][2];
    const bool bad = gram_query_operands<DC>(a, hdr, qrow, g, bhd, btl);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) atomicExch(flag, gen);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned char* ps = p + s * (kVgDist / 2);
      *reinterpret_cast<h4_t*>(ps + lane * 8) = bhd[s];
      *reinterpret_cast<h8_t*>(ps + 512 + lane * 16) = btl[s][0];
      *reinterpret_cast<h8_t*>(ps + 512 + 1024 + lane * 16) = btl[s][1];
    }
    return;
  }
  float mx = 0.0f;                             // max |g| over the batch from the per-block maxima of colsum_partial_kernel
  for (int i = lane; i < nbmax; i += 64) {
    const float v = bmax[i];
    mx = (v > mx || v != v) ? v : mx;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o2 = __shfl_xor(mx, off);
    mx = (o2 > mx || o2 != o2) ? o2 : mx;
  }
  const float sg = vg_pow2_ceil_scale(mx);
  float somax = 0.0f;
  for (int o = 0; o < O; ++o) somax = fmaxf(somax, oscale[o]);
  const float sh = sg * somax;
  if (blockIdx.x == 0 && role == 1 && lane == 0) { scales[0] = sg; scales[1] = sh; }
  // gate of the single region (model.py:42-95) of query q0 + lane (lanes >= 32: 0)
  float gm = 0.0f;
  if (lane < 32 && q0 + lane < B) {
    const long q = q0 + lane;
    gm = gt.n_ranges > 0 ? 1.0f : 0.0f;
    for (int d = 0; d < gt.nsplit && gt.n_ranges > 0; ++d) {
      const int e = d * gt.max_ranges + gt.dim_ranges[d];
      gm *= gate_factor(x[q * D + d], gt.lo[e], gt.hi[e], gt.delta[d]);
    }
  }
  // A operands of the hbar MFMAs per half (rows = queries 16 s + n), v = gamma_q g / s_h x s_o as an (hi, lo) pair:
  //   A1 (16x16x16): k = output 4 g + j, hi                     x  W hi
  //   A2 (16x16x32): k = 8 g + j: (g < 2: lo, g >= 2: hi) of output 8 (g & 1) + j   x  (W hi | W lo) -- lo x hi + hi x lo in ONE MFMA
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  const float sclA = __builtin_ldexpf(1.0f, -expA);
  h4v* gA1 = reinterpret_cast<h4v*>(p + kVgDist);
  h8_t* gA2 = reinterpret_cast<h8_t*>(p + kVgDist + 1024);
  if (role == 1) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const long q = q0 + 16 * s + n;
    const float gq = __shfl(gm, 16 * s + n);
    h4v hi;
    h8_t mix;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 4 * g + j;
      float v = 0.0f;
      if (q < B && o < O) v = gq * gout[q * O + o] * (oscale[o] / sh) * sclA;
      _Float16 h, l;
      split_static_f16(v, h, l);
      hi[j] = h;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // O <= kVgOC: ONE operand, k = 8 g + j: [0, 10) hi of output k, [10, 20) lo of output k - 10, [20, 30) hi of output k - 20 --
      // hi x hi + lo x hi + hi x lo as one MFMA against (W hi | 2^-11 W hi | 2^-11 W lo); else A2 as described above
      const int k = 8 * g + j;
      const int o = O <= kVgOC ? (k < 30 ? k % 10 : 16) : 8 * (g & 1) + j;
      const bool want_lo = O <= kVgOC ? (k >= 10 && k < 20) : g < 2;
      float v = 0.0f;
      if (q < B && o < O) v = gq * gout[q * O + o] * (oscale[o] / sh) * sclA;
      _Float16 h, l;
      split_static_f16(v, h, l);
      // O <= kVgOC: the cross terms' 2^-11 is shared between the two sides (2^-5 here, 2^-6 on the weights) so that neither
      // operand sinks deeper into the f16 subnormals than 2^-17 of its largest entry
      mix[j] = (O <= kVgOC && k >= 10) ? (_Float16)((float)(want_lo ? l : h) * kVgCrossA) : (want_lo ? l : h);
    }
    gA1[s * 64 + lane] = hi;
    gA2[s * 64 + lane] = mix;
  }
  return;
  }
  // A operand of the dW MFMA (16x16x32): rows = outputs n, k = 8 g + j <-> query 16 (j >> 2) + 4 g + (j & 3), gamma_q g / s_g
  // B operand of the dC MFMA (16x16x32): k as above, cols = coordinate n of x' / 2^ex (n < D), 1 (n = 8), 0
  h8_t* gT = reinterpret_cast<h8_t*>(p + kVgDist + kVgGA);
  h8_t* xB = reinterpret_cast<h8_t*>(p + kVgDist + kVgGA + 2048);
  const float xinv = __builtin_ldexpf(1.0f, -hdr->ex);
  const float rn = n < D && n < kGramDims ? hdr->r[n] : 0.0f;
  h8_t th, tl, xh, xl;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ql = 16 * (j >> 2) + 4 * g + (j & 3);
    const long q = q0 + ql;
    const float gq = __shfl(gm, ql);
    float v = 0.0f, xv = 0.0f;
    if (q < B && n < O) v = gq * gout[q * O + n] / sg;
    if (q < B) xv = n < D ? (x[q * D + n] - rn) * xinv : (n == kGramDims ? 1.0f : 0.0f);
    _Float16 h, l;
    split_static_f16(v, h, l);
    th[j] = h; tl[j] = l;
    split_static_f16(xv, h, l);
    xh[j] = h; xl[j] = l;
  }
  gT[lane] = th; gT[64 + lane] = tl;
  xB[lane] = xh; xB[64 + lane] = xl;
}

// ---- main kernel -------------------------------------------------------------------------------------------
// (hi, lo) of two values WITHOUT the 2^11 gain on lo: hi = the packed toward-zero conversion, lo = p - hi (one v_fma_mix_f32 that
// reads hi as the f16 number it is: exact) converted as it is -- 4 instructions per pair instead of the 6 of split_pair_mix.
// lo < 2^-10 |p| falls into the f16 subnormals for |p| < 2^-4 and is then good to 2^-25 ABSOLUTE only (the matrix cores honour
// subnormal inputs): harmless here, where p <= 2^14..2^15 and a gradient is judged against the largest entry of its leaf (the
// forward, judged per output, keeps the gain).  The products with an ungained lo carry the scale of hi x hi and share its accumulator.
__device__ __forceinline__ void split_pair_plain(float p0, float p1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  float d0, d1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "v"(p1));
  lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d0, d1));
}

// The 8 transcendentals of a query half straight from the MFMA result registers into fresh ones (the in-place form costs 8 v_mov
// copies per half: u is needed again for d log_sigs).  The hazard recogniser does not look into inline asm: the wait states an
// MFMA result needs before a VALU instruction may read it are the leading s_nop 7.
template <int BC>
__device__ __forceinline__ void trans8_from_mfma(const f4_t (&u)[2], float (&o)[8]) {
#define IRBFN_T8G(OP)                                                                                                              \
  asm volatile("s_nop 7\n " OP " %0, %8\n " OP " %1, %9\n " OP " %2, %10\n " OP " %3, %11\n " OP " %4, %12\n " OP " %5, %13\n " OP         \
               " %6, %14\n " OP " %7, %15\n s_nop 0"                                                                                \
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])           \
               : "v"(u[0][0]), "v"(u[0][1]), "v"(u[0][2]), "v"(u[0][3]), "v"(u[1][0]), "v"(u[1][1]), "v"(u[1][2]), "v"(u[1][3]));
  if constexpr (BC == BC_GAUSS) { IRBFN_T8G("v_exp_f32_e32") }
  else if constexpr (BC == BC_IQ) { IRBFN_T8G("v_rcp_f32_e32") }
  else { IRBFN_T8G("v_rsq_f32_e32") }
#undef IRBFN_T8G
}

struct VjpGArgs {
  const unsigned char* __restrict__ qblk;   // [nqb][kVgBlock]
  const float* __restrict__ scales;         // [0] = s_g, [1] = s_h
  const unsigned char* __restrict__ gimg;   // K1g's chunk images: centre-side operands of u
  const GramHdr* __restrict__ hdr;
  const int* __restrict__ flag;             // == gen: a query of THIS call outside the box -- K2h does the work
  int gen;
  const float* __restrict__ rec;            // [N][S] K1 records: c[DC], scale, W[OP]
  const float* __restrict__ sig2;           // [N]
  const float* __restrict__ oscale;         // [O]
  float* __restrict__ part;                 // [QSB][V][Npad]
  long nqb;
  int O, OP, N, S, Npad, bpb, cstride, nchunks;
  float gscale;
};

#ifndef IRBFN_K2G_WAVES
#define IRBFN_K2G_WAVES 3       // waves per SIMD the register allocation must allow where hbar takes two MFMAs (O > 10): 138 VGPRs
#endif
#ifndef IRBFN_K2G_WAVES_OC
#define IRBFN_K2G_WAVES_OC 4    // O <= 10: fits 128 VGPRs without a spill; config 3: 154 against 158 us (with 1024 blocks per launch, rbf_vjp.hip).
                                // (Until the head sums went out through gram_heads2 -- rbf_forward_gram.h -- every 4-waves build was WRONG.)
#endif
// OC: O <= kVgOC -- hbar's three products (hi x hi, lo x hi, hi x lo over <= 10 outputs: 30 of 32 slots) in ONE 16x16x32 MFMA; the
// cross terms' B operands carry the 2^-11 of the (hi, lo) scheme, so nothing is left to combine on the VALU (28 instead of 32
// MFMAs and 16 VALU instructions fewer per 32 x 32 pairs)
template <int DC, int BC, bool OC>
__global__ __launch_bounds__(256, OC ? IRBFN_K2G_WAVES_OC : IRBFN_K2G_WAVES) void rbf_vjp_f16gram(const VjpGArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (*a.flag == a.gen) return;
  // (centre block, query slice) in plain grid order: placing the blocks that stream the same query slice on one XCD -- one L2 --
  // was measured and changes nothing (profiles/r03_vjp_qsb_sweep.txt): the 12 KiB query blocks are not what the kernel waits for
  const int bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int chunk = bx * 4 + wave;                   // this wave's 32 centres
  const bool active = chunk < a.nchunks;
  const int cb = chunk * 32;
  const float sg = a.scales[0], sh = a.scales[1];
  // Gaussian: the head sum starts from -14 (gram_heads2c), so u = alpha d^2 itself and the transcendental returns phi (PS = 1):
  // no subtraction per pair for d log_sigs.  phi <= 1 as an ungained (hi, lo) pair is good to 2^-25 absolute -- of a gradient's
  // largest terms, which is what a gradient is judged against.  The other bases keep P = 2^14 phi / 2^7 phi.
  constexpr bool kVgShift = BC == BC_GAUSS;
  constexpr float PS = kVgShift ? 1.0f : gram_phi_scale<BC>();                 // the basis value arrives as P = PS phi
  // tt_true = tts * KT:  tts = hb * P^p * 2^-ET (|tts| < 2^15), hb = gamma hbar 2^30 / s_h, dphi/dd2 = const * phi^p
  constexpr int ET = BC == BC_GAUSS ? 19 : (BC == BC_IQ ? 47 : 40);       // 34 + p log2(PS) - 15
  constexpr int kVgExpA = vg_exp_a(BC), kVgExpB = vg_exp_b(BC);
  const float cE = __builtin_ldexpf(1.0f, -(ET - kVgExpA - kVgExpB));      // what the operand scales leave of 2^-ET (gaussian: 1)
  static_assert(ET >= kVgExpA + kVgExpB, "operand scales");
  const float KT = sh * (1.0f / (kWScale * kWScale)) * __builtin_ldexpf(1.0f, ET) *
                   (BC == BC_GAUSS ? -a.gscale / PS : (BC == BC_IQ ? -1.0f / (PS * PS) : -0.5f / (PS * PS * PS)));

  // centre-side operands (loop invariant): u's B operands from K1g's chunk image, hbar's B operands from the records
  h4_t cbh[2];
  h8_t cbt[2][2];
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  const float sclB = __builtin_ldexpf(1.0f, -kVgExpB);
  h4v wth[2];                                                // B of A1: k = output 4 g + j, W hi
  h8_t wt2[2];                                               // B of A2: k = 8 g + j: (g < 2: W hi, g >= 2: W lo) of output 8 (g & 1) + j
  {
    const unsigned char* ci = a.gimg + (size_t)(active ? chunk : 0) * a.cstride;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      cbh[ct] = *reinterpret_cast<const h4_t*>(ci + ct * 512 + lane * 8);
      cbt[ct][0] = *reinterpret_cast<const h8_t*>(ci + kGramHeadBytes + (ct * 2 + 0) * 1024 + lane * 16);
      cbt[ct][1] = *reinterpret_cast<const h8_t*>(ci + kGramHeadBytes + (ct * 2 + 1) * 1024 + lane * 16);
      int cid = cb + ct * 16 + n;
      cid = cid < a.N ? cid : a.N - 1;
      const float* rp = a.rec + (size_t)cid * a.S;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = 4 * g + j;
        const float w = o < a.O ? rp[DC + 1 + o] / a.oscale[o] * sclB : 0.0f;
        _Float16 h, l;
        split_static_f16(w, h, l);
        wth[ct][j] = h;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        const int o = OC ? (k < 30 ? k % 10 : 16) : 8 * (g & 1) + j;
        const float w = o < a.O ? rp[DC + 1 + o] / a.oscale[o] * sclB : 0.0f;
        _Float16 h, l;
        split_static_f16(w, h, l);
        if (OC) wt2[ct][j] = k < 10 ? h : (_Float16)((float)(k < 20 ? h : l) * kVgCrossB);   // exact (power of two) down to the subnormals
        else wt2[ct][j] = g < 2 ? h : l;
      }
    }
  }
  f4_t dW[2], dWl[2], dC[2], dCl[2];
  float gls[2] = {0.0f, 0.0f};
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) { dW[ct] = f4_t{0, 0, 0, 0}; dWl[ct] = f4_t{0, 0, 0, 0}; dC[ct] = f4_t{0, 0, 0, 0}; dCl[ct] = f4_t{0, 0, 0, 0}; }

  const long qb0 = (long)by * a.bpb;
  long qb1 = qb0 + a.bpb;
  qb1 = qb1 < a.nqb ? qb1 : a.nqb;
  const int nb = qb1 > qb0 ? (int)(qb1 - qb0) : 0;
  auto stage = [&](int k, int buf) {                         // block qb0 + k -> ring slot buf; the 4 waves share the copy
    if (k >= nb) return;
    const unsigned char* gp = a.qblk + (size_t)(qb0 + k) * kVgBlock + lane * 16;
    unsigned char* dst = lds + buf * kVgBlock;
    for (int v = wave; v < kVgPieces; v += 4)
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  stage(0, 0);
  stage(1, 1);
  step_barrier();
  stage(2, 2);
  int b0 = 0;
  for (int i = 0; i < nb; ++i) {
    const unsigned char* cur = lds + b0 * kVgBlock;
    if (active) {
      float hq[2][8], tq[2][8];                              // P and tts of the lane's 8 queries x 2 centre tiles
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned char* ps = cur + s * (kVgDist / 2);
        const h4_t qh = *reinterpret_cast<const h4_t*>(ps + lane * 8);
        const h8_t qt0 = *reinterpret_cast<const h8_t*>(ps + 512 + lane * 16);
        const h8_t qt1 = *reinterpret_cast<const h8_t*>(ps + 512 + 1024 + lane * 16);
        const h4v gah = *reinterpret_cast<const h4v*>(cur + kVgDist + (s * 64 + lane) * 8);
        const h8_t ga2 = *reinterpret_cast<const h8_t*>(cur + kVgDist + 1024 + (s * 64 + lane) * 16);
        f4_t hb[2], u[2];
        if constexpr (kVgShift) gram_heads2c(qh, cbh, f4_t{-(float)kPhiExp, -(float)kPhiExp, -(float)kPhiExp, -(float)kPhiExp}, u);
        else gram_heads2(qh, cbh, u);                        // exact head sums (rbf_forward_gram.h: why not the builtin)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          f4_t hl = f4_t{0, 0, 0, 0};
          if constexpr (OC) {
            hb[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga2, wt2[ct], f4_t{0, 0, 0, 0}, 0, 0, 0);          // all three products
          } else {
            hb[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(gah, wth[ct], f4_t{0, 0, 0, 0}, 0, 0, 0);
            hl = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga2, wt2[ct], f4_t{0, 0, 0, 0}, 0, 0, 0);              // lo x hi + hi x lo
          }
          u[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qt0, cbt[ct][0], u[ct], 0, 0, 0);
          u[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qt1, cbt[ct][1], u[ct], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (!OC) hb[ct][r] = __builtin_fmaf(hl[r], kLoScale, hb[ct][r]);
            if (ET > kVgExpA + kVgExpB) hb[ct][r] *= cE;
          }
        }
        float t8[8];
        trans8_from_mfma<BC>(u, t8);                         // P = PS phi
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float P = t8[ct * 4 + r];
            float pw = P;
            if constexpr (BC == BC_IQ) pw = P * P;
            if constexpr (BC == BC_IMQ) pw = P * P * P;
            const float tts = hb[ct][r] * pw;
            // alpha d2 (gaussian: u - 14) resp. d2 / sigma^2 (others: 2^14 u - 1) per pair; the centre's own factor at the end.
            // (sum tts u - 14 sum tts with the sum from the dC product's column of ones saves this instruction and costs a factor
            // 14 / <alpha d2> in accuracy: 7e-6 instead of 1e-6 of max |d log_sigs| at config 3 -- measured, not taken)
            const float v = kVgShift ? u[ct][r] : __builtin_fmaf(u[ct][r], kPhiScale, -1.0f);
            gls[ct] = __builtin_fmaf(tts, v, gls[ct]);
            hq[ct][4 * s + r] = P;
            tq[ct][4 * s + r] = tts;
          }
      }
      const h8_t gth = *reinterpret_cast<const h8_t*>(cur + kVgDist + kVgGA + lane * 16);
      const h8_t gtl = *reinterpret_cast<const h8_t*>(cur + kVgDist + kVgGA + 1024 + lane * 16);
      const h8_t xbh = *reinterpret_cast<const h8_t*>(cur + kVgDist + kVgGA + 2048 + lane * 16);
      const h8_t xbl = *reinterpret_cast<const h8_t*>(cur + kVgDist + kVgGA + 2048 + 1024 + lane * 16);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        unsigned ph[4], pl[4], th[4], tl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          split_pair_plain(hq[ct][2 * jj], hq[ct][2 * jj + 1], ph[jj], pl[jj]);
          split_pair_plain(tq[ct][2 * jj], tq[ct][2 * jj + 1], th[jj], tl[jj]);
        }
        const h8_t bh = __builtin_bit_cast(h8_t, u4_t{ph[0], ph[1], ph[2], ph[3]});
        const h8_t bl = __builtin_bit_cast(h8_t, u4_t{pl[0], pl[1], pl[2], pl[3]});
        const h8_t ah = __builtin_bit_cast(h8_t, u4_t{th[0], th[1], th[2], th[3]});
        const h8_t al = __builtin_bit_cast(h8_t, u4_t{tl[0], tl[1], tl[2], tl[3]});
        // the lo halves of P and tts carry no gain (split_pair_plain): their products join the hi x hi accumulators; the pre-packed
        // g lo and x' lo carry 2^11 and keep the second ones
        dW[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gth, bh, dW[ct], 0, 0, 0);
        dWl[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gtl, bh, dWl[ct], 0, 0, 0);
        dC[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xbh, dC[ct], 0, 0, 0);
        dCl[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xbl, dCl[ct], 0, 0, 0);
        dW[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gth, bl, dW[ct], 0, 0, 0);
        dC[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xbh, dC[ct], 0, 0, 0);
      }
    }
    step_barrier();                                          // block i + 2 is there; everybody has left block i
    stage(i + 3, b0);
    b0 = b0 == 2 ? 0 : b0 + 1;
  }
  if (!active) return;

  // ---- this wave's 32 centres: slab rows (format of rbf_vjp_kernel): d centers [0, DC), d log_sigs DC, d kernel DC + 1 + o
  const int V = DC + 1 + a.OP;
  float* dst = a.part + (size_t)by * V * a.Npad;
  const float wscale = sg * (1.0f / (PS * kWScale));         // s_g / (scale of the basis values x 2^15)
  const float xs = __builtin_ldexpf(1.0f, a.hdr->ex - kWExp);   // x' operand: x' 2^-ex 2^15
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    // d log_sigs: the 4 lane groups hold different queries of centre n
    float v = gls[ct];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    const int cid = cb + ct * 16 + n;
    if (g == 0 && cid < a.Npad) {
      const int cc = cid < a.N ? cid : a.N - 1;
      const float scr = a.rec[(size_t)cc * a.S + DC];        // gaussian: alpha = -a log2(e) / sigma^2; others: 1 / sigma^2
      dst[(size_t)DC * a.Npad + cid] = cid < a.N ? -2.0f * a.sig2[cc] * KT * v / scr : 0.0f;
    }
    // d kernel: D rows = outputs 4 g + r, column = centre n
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * g + r;
      if (o < a.OP && cid < a.Npad) dst[(size_t)(DC + 1 + o) * a.Npad + cid] = __builtin_fmaf(dWl[ct][r], kLoScale, dW[ct][r]) * wscale;
    }
    // d centers: D rows = centres 4 g + r, column n = coordinate (n < DC) / the sum of tt (n = 8)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dv = __builtin_fmaf(dCl[ct][r], kLoScale, dC[ct][r]);
      const float st = __shfl(dv, g * 16 + kGramDims) * (1.0f / kWScale);   // sum of tts over the slice
      const int c2 = cb + ct * 16 + 4 * g + r;
      if (n < DC && c2 < a.Npad) {
        float out = 0.0f;
        if (c2 < a.N) {
          const float cp = n < kGramDims ? a.rec[(size_t)c2 * a.S + n] - a.hdr->r[n] : 0.0f;
          out = -2.0f * a.sig2[c2] * KT * (dv * xs - cp * st);
        }
        dst[(size_t)n * a.Npad + c2] = out;
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
bool vjpg_eligible(const irbfn_net* net) {
  return vjph_eligible(net) && net->gram_img != nullptr && net->gram_ok && net->DC <= kGramDims;
}

size_t vjpg_block_bytes() { return kVgBlock; }

template <int DC>
static int launch_vjpg_dc(const VjpGArgs& a, int bc, dim3 grid, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: if (a.O <= kVgOC) hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_GAUSS, true>), grid, dim3(256), lds, s, a);
                   else hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_GAUSS, false>), grid, dim3(256), lds, s, a);
                   break;
    case BC_IQ: if (a.O <= kVgOC) hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_IQ, true>), grid, dim3(256), lds, s, a);
                else hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_IQ, false>), grid, dim3(256), lds, s, a);
                break;
    case BC_IMQ: if (a.O <= kVgOC) hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_IMQ, true>), grid, dim3(256), lds, s, a);
                 else hipLaunchKernelGGL((rbf_vjp_f16gram<DC, BC_IMQ, false>), grid, dim3(256), lds, s, a);
                 break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// x, gout -> slabs part[QSB][V][Npad].  qblk / scales / flag: workspace; bmax: per-block max |g| written by colsum_partial_kernel
int launch_vjp_gram(irbfn_net* net, const float* x, const float* gout, int64_t B, unsigned char* qblk, const float* bmax, int nbmax,
                    float* scales, int* flag, int gen, float* part, int QSB, int Npad, hipStream_t s) {
  if (!vjpg_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  const long nqb = (B + 31) / 32;
  const GramHdr* hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
  const dim3 pg((unsigned)nqb), pb(192);
  switch (net->DC) {
    case 3: hipLaunchKernelGGL((vjp_pack_blocks_gram_kernel<3>), pg, pb, 0, s, x, gout, bmax, nbmax, net->f16_oscale, hdr, qblk, scales, flag, gen, net->gate(), (long)B, net->D, net->O, vg_exp_a(net->bclass)); break;
    case 4: hipLaunchKernelGGL((vjp_pack_blocks_gram_kernel<4>), pg, pb, 0, s, x, gout, bmax, nbmax, net->f16_oscale, hdr, qblk, scales, flag, gen, net->gate(), (long)B, net->D, net->O, vg_exp_a(net->bclass)); break;
    case 7: hipLaunchKernelGGL((vjp_pack_blocks_gram_kernel<7>), pg, pb, 0, s, x, gout, bmax, nbmax, net->f16_oscale, hdr, qblk, scales, flag, gen, net->gate(), (long)B, net->D, net->O, vg_exp_a(net->bclass)); break;
    case 8: hipLaunchKernelGGL((vjp_pack_blocks_gram_kernel<8>), pg, pb, 0, s, x, gout, bmax, nbmax, net->f16_oscale, hdr, qblk, scales, flag, gen, net->gate(), (long)B, net->D, net->O, vg_exp_a(net->bclass)); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  VjpGArgs a;
  a.qblk = qblk; a.scales = scales; a.gimg = net->gram_img; a.hdr = hdr; a.flag = flag; a.gen = gen; a.rec = net->rec; a.sig2 = net->sig2;
  a.oscale = net->f16_oscale; a.part = part;
  a.nqb = nqb; a.O = net->O; a.OP = net->OP; a.N = net->N; a.S = net->S; a.Npad = Npad;
  a.bpb = (int)((nqb + QSB - 1) / QSB);
  a.cstride = gram_chunk_bytes((net->O + 15) / 16);
  a.nchunks = (net->N + 31) / 32;
  a.gscale = gauss_scale(net->basis);
  const dim3 grid((a.nchunks + 3) / 4, QSB);
  const size_t lds = (size_t)3 * kVgBlock;
  snprintf(net->last_name, sizeof(net->last_name), "rbf_vjp_f16gram<D=%d,BC=%d,QSB=%d>", net->DC, net->bclass, QSB);
  net->last_grid = (int)(grid.x * grid.y);
  net->last_block = 256;
  switch (net->DC) {
    case 3: return launch_vjpg_dc<3>(a, net->bclass, grid, lds, s);
    case 4: return launch_vjpg_dc<4>(a, net->bclass, grid, lds, s);
    case 7: return launch_vjpg_dc<7>(a, net->bclass, grid, lds, s);
    case 8: return launch_vjpg_dc<8>(a, net->bclass, grid, lds, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

}  // namespace irbfn
