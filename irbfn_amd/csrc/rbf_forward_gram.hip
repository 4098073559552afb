// K1g: the narrow forward (one region, O <= 16, fast bases) with BOTH of its GEMM-shaped pieces on the f16 matrix cores:
// the squared distances as a Gram expansion and Phi x W (the hi/lo pairs of K1h).  Same mathematics as K1 / K1h
// (src/irbfn_mpc/model.py:169-198; RBF stage flax_rbf.py:258-285).
//
// Why: K1h is VALU-issue bound, and 14 of its ~21.5 VALU instructions per (query, centre) pair are the distance
// (x_i - c_i, fma) x 7.  With u = alpha_k |x - c_k|^2 + beta (the argument of the transcendental) written as
//     u = alpha_k Q  +  sum_i (-2 alpha_k c'_ki) x'_i  +  (alpha_k |c'_k|^2 + beta),      x' = x - r, c' = c - r, Q = |x'|^2
// it is a [centres x slots] x [slots x queries] product whose D layout (v_mfma: lane = query column, registers = 4 centre
// rows) is already the A layout of the Phi x W product: no shuffle between the two.  What is left on the VALU per pair is the
// transcendental and the hi/lo split of its result.
//
// Accuracy -- the expansion cancels (terms of size M = alpha (|x'|^2 + |c'|^2) sum to u <~ 30), and the matrix core adds its
// products in a tree of truncating ~24-bit adders (tools/probe_mfma_accum.hip, profiles/r03_mfma_accumulation_probe.txt), so a
// naive expansion carries an error of 2^-24 M.  Here the cancellation is EXACT:
//  * every factor v in {x'_i, Q, C_ki = -2 alpha_k c'_ki, alpha_k, c2_k} with |v| < 2^E is cut into a FIXED-POINT head
//    n0 = rint(v 2^(10-E)) 2^-10 (11 bits: exact in f16) and two float tails n1, n2 (f16 roundings of the scaled residuals:
//    v = 2^E (n0 + 2^-11 n1 + 2^-22 n2) to 2^(E-34)); c2 has two fixed-point heads;
//  * the head x head products of one (query, centre) element are integer multiples of one grid 2^(EX+EC-20) and their partial
//    sums stay below 2^24 grid units (checked at pack time): whatever the order of the adder tree, nothing is truncated --
//    the first MFMA (16x16x16, 10 of its 16 k-slots) returns the cancelled head sum S1 exactly;
//  * the 42 tail products (each < 2^-10 of a head product) are added to S1 by two 16x16x32 MFMAs: truncation there is
//    2^-24 of max(|u|, 2^-10 M).
// So u carries the error of a float32 evaluation of the direct form (a few 2^-24 |u|), not 2^-24 M.  Exponents EX .. E2 are
// chosen per net from the centres (gram_stats_kernel); a query outside the representable box (|x'_i| >= 2^EX, non-finite)
// makes its WAVE take the VALU distances of K1h for its 32 queries (same records, same f16_arg): results there are K1h's.
// A net whose exponents do not fit (ok = 0 in the header, read back once by irbfn_net_set_params) is not dispatched here.
//
// Layout: v_mfma_f32_16x16x16_f16 / 16x16x32_f16 with rows = 16 centres, k = slots, columns = 16 queries.  A lane owns
// query (l & 15) of its wave's two query tiles; the D registers of centre tile ct hold centres 16 ct + 4 g + r (g = l >> 4),
// which become k-slots 8 g + 4 ct + r of the Phi x W product (the W rows of the image are permuted to match).
// Chunk image (32 centres, 7 KiB): head operands [ct][lane] 8 B, tail operands [ct][half][lane] 16 B, W hi, W lo (1 KiB each);
// the QG waves of a centre slice share an LDS ring of five images filled by LDS-DMA, one barrier per two chunks.
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rbf_forward_gram.h"

namespace irbfn {

// ---- pack ------------------------------------------------------------------------------------------------
template <int BC>
__device__ inline void gram_alpha_beta(double s2, double gscale, double& alpha, double& beta) {
  if (BC == BC_GAUSS) { alpha = -gscale * 1.4426950408889634 * s2; beta = (double)kPhiExp; }     // P = 2^(alpha d2 + 14)
  else if (BC == BC_IQ) { alpha = s2 * (double)kPhiInv; beta = (double)kPhiInv; }                // P = 1 / (2^-14 (1 + d2 s2))
  else { alpha = s2 * (double)kPhiInv; beta = (double)kPhiInv; }                                 // P = rsqrt(2^-14 (1 + d2 s2)) = 2^7 phi
}
__device__ inline double gram_pow2(int e) {                  // 2^e, -1022 <= e <= 1023, without the library's ldexp
  return __builtin_bit_cast(double, (unsigned long long)(1023 + e) << 52);
}
__device__ inline int gram_exp_above(double v) {            // smallest e with |v| < 2^e
  if (!(v > 0.0)) return -40;
  int e;
  (void)frexp(v, &e);                                        // v = f 2^e, f in [0.5, 1)
  return e;
}

// one block: origin, exponents, the exactness budget
__global__ __launch_bounds__(1024) void gram_stats_kernel(const float* __restrict__ centers, const float* __restrict__ log_sigs,
                                                          GramHdr* __restrict__ hdr, int N, int D, int bclass,
                                                          float gscale) {
  __shared__ float red[16][16];                              // [wave][value]
  __shared__ float r_sh[8];
  __shared__ float tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // max of NV values per thread over the block: wave shuffles, one LDS pass, results in tot[]
  auto block_max_n = [&](float (&v)[16], int NV) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < NV) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] = fmaxf(v[i], __shfl_xor(v[i], off));
      }
    if (lane == 0)
      for (int i = 0; i < NV; ++i) red[wave][i] = v[i];
    __syncthreads();
    if (tid < NV) {
      float m = red[0][tid];
      for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w][tid]);
      tot[tid] = m;
    }
    __syncthreads();
  };
  // ONE pass: per coordinate max and -min of the centres (origin = midpoint, half widths), the largest |alpha|, finiteness.  The
  // bounds on C = -2 alpha c' and on c2 = alpha |c'|^2 + beta are products of these maxima (at most the true maxima x the spread
  // of alpha over the centres: a coarser grid where it matters, never a wrong one) -- a second pass over the centres with the
  // origin known would make them tight at twice the kernel time, and this kernel runs at every irbfn_net_set_params.
  bool finite = true;
  float mm[16];                                              // [0, 8): max_i, [8, 16): -min_i
  for (int i = 0; i < 16; ++i) mm[i] = -3.0e38f;
  float am = 0.0f;
  for (int k = tid; k < N; k += 1024) {
    for (int i = 0; i < D && i < 8; ++i) {
      const float c = centers[(size_t)k * D + i];
      finite = finite && (fabsf(c) < 3.0e38f);
      mm[i] = fmaxf(mm[i], c);
      mm[8 + i] = fmaxf(mm[8 + i], -c);
    }
    const double s2 = (double)expf(-2.0f * log_sigs[k]);     // 1/sigma^2 as K1 / K1h have it (float32, flax_rbf.py:280)
    double alpha, beta;
    if (bclass == BC_GAUSS) gram_alpha_beta<BC_GAUSS>(s2, gscale, alpha, beta);
    else if (bclass == BC_IQ) gram_alpha_beta<BC_IQ>(s2, gscale, alpha, beta);
    else gram_alpha_beta<BC_IMQ>(s2, gscale, alpha, beta);
    const float fa1 = (float)fabs(alpha);
    finite = finite && (fa1 < 1.0e30f);
    am = fmaxf(am, fa1);
  }
  block_max_n(mm, 16);
  float hw2 = 0.0f, hwm = 0.0f;                              // sum of the squared half widths, largest half width
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) {
      r_sh[i] = i < D ? 0.5f * tot[i] - 0.5f * tot[8 + i] : 0.0f;
      const float hw = i < D ? fmaxf(tot[i] - r_sh[i], r_sh[i] + tot[8 + i]) * 1.0000005f : 0.0f;     // the rounded midpoint's two sides
      hw2 += hw * hw;
      hwm = fmaxf(hwm, hw);
    }
  }
  float st[16];
  for (int i = 0; i < 16; ++i) st[i] = 0.0f;
  st[0] = am; st[1] = finite ? 0.0f : 1.0f;
  block_max_n(st, 2);
  double beta0, alpha0;
  if (bclass == BC_GAUSS) gram_alpha_beta<BC_GAUSS>(1.0, gscale, alpha0, beta0);
  else if (bclass == BC_IQ) gram_alpha_beta<BC_IQ>(1.0, gscale, alpha0, beta0);
  else gram_alpha_beta<BC_IMQ>(1.0, gscale, alpha0, beta0);
  const float fc = hwm, fa = tot[0], fC = 2.0f * tot[0] * hwm, f2 = tot[0] * hw2 + (float)fabs(beta0), bad = tot[1];
  if (tid == 0) {
    GramHdr h;
    for (int i = 0; i < 8; ++i) h.r[i] = r_sh[i];
    // the box of representable queries: the centres' box with a quarter to spare (queries beyond it take the VALU distances;
    // the card's own bounds are NOT added: a wide gate around a compact set of centres would coarsen every head)
    const double xm = fc;
    h.ex = gram_exp_above(1.25 * xm * 1.0000002);
    h.ea = gram_exp_above(fa * 1.0000002);
    h.ec = gram_exp_above(fC * 1.0000002);
    int dbits = 0;
    while ((1 << dbits) < D) ++dbits;
    h.cabs = fc; h.amax = fa; h.cmax = fC; h.c2max = f2;
    // exactness budget of the head sum: the sum of the magnitudes of its terms -- a bound on every partial sum of the adder
    // tree -- stays below 2^24 grid units, grid = 2^(ex + ec - 20).  |x'_i| < 2^ex is what the kernel lets through.  A coarser
    // grid (ec + 1: heads of C one bit shorter, its tails one bit larger) buys a factor of two.
    const double xb = ldexp(1.0, h.ex);
    const double worst = ((double)fa * D * xb * xb + (double)D * fC * xb + (double)f2) * (1.0 + 1.0 / 256.0);
    while (worst >= ldexp(1.0, h.ex + h.ec + 4) && h.ec < 40) ++h.ec;
    h.eq = 2 * h.ex + dbits;
    if (h.eq + h.ea < h.ex + h.ec) h.eq = h.ex + h.ec - h.ea;                 // Q x alpha heads on the cross grid
    h.e2 = gram_exp_above(f2 * 1.0000002);
    if (h.e2 < h.ex + h.ec + 1) h.e2 = h.ex + h.ec + 1;                        // second c2 head on the cross grid
    bool ok = bad == 0.0f && fc > 0.0f && fa > 0.0f;
    // truncation in the two tail MFMAs: at most 2^-24 of D tail products of 2^(ex + ec - 10) each -- kept below 2^-20 in u
    ok = ok && h.ex + h.ec + dbits <= 14;
    // f16 range of every operand: |operand| <= 2^ax resp. 2^(T - ax), heads need ax - 10 >= -24
    const int Ts[3] = {h.ex + h.ec, h.eq + h.ea, h.e2};
    for (int t = 0; t < 3; ++t) ok = ok && gram_ax(Ts[t]) <= 14 && Ts[t] - gram_ax(Ts[t]) <= 14 && Ts[t] >= -20;
    ok = ok && h.ex <= 12 && h.ex >= -12;
    h.ok = ok ? 1 : 0;
    *hdr = h;
  }
}

// v (|v| < 2^E) -> n0 (fixed point, grid 2^-10), n1, n2 (f16 values), v = 2^E (n0 + 2^-11 n1 + 2^-22 n2)
__device__ inline void gram_parts_d(double v, int E, double (&n)[3]) {
  const double a = v * gram_pow2(-E);
  n[0] = __builtin_rint(a * 1024.0) * (1.0 / 1024.0);
  const double r1 = (a - n[0]) * 2048.0;
  n[1] = (double)(_Float16)(float)r1;
  const double r2 = (r1 - n[1]) * 2048.0;
  n[2] = (double)(_Float16)(float)r2;
}
// c2: two fixed-point heads, two float tails
__device__ inline void gram_parts_c2(double v, int E, double (&n)[4]) {
  const double a = v * gram_pow2(-E);
  n[0] = __builtin_rint(a * 1024.0) * (1.0 / 1024.0);
  const double r1 = (a - n[0]) * 2048.0;
  n[1] = __builtin_rint(r1 * 1024.0) * (1.0 / 1024.0);
  const double r2 = (r1 - n[1]) * 2048.0;
  n[2] = (double)(_Float16)(float)r2;
  const double r3 = (r2 - n[2]) * 2048.0;
  n[3] = (double)(_Float16)(float)r3;
}

// sixteen threads per (chunk, centre-in-chunk): thread `part` < 8 writes the slots of coordinate `part` (1 head, 5 tails), thread 8
// the slots of Q x alpha and of c2 (3 heads, 7 tails) and the empty ones; every thread the W values of output `part` of every
// column tile
constexpr int kGramPackThreads = 16;
static_assert(kGramDims + 1 <= kGramPackThreads && kGramDims + 3 <= 16 && 5 * kGramDims + 7 <= 64, "slots of the expansion");
template <int BC>
__global__ __launch_bounds__(256) void gram_pack_kernel(const float* __restrict__ centers, const float* __restrict__ log_sigs,
                                                        const float* __restrict__ kernel, const float* __restrict__ oscale,
                                                        const GramHdr* __restrict__ hdr, unsigned char* __restrict__ img, int N,
                                                        int K, int D, int O, int NT, float gscale, int nchunks) {
  const int tix = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = tix / kGramPackThreads, part = tix % kGramPackThreads;
  if (idx >= nchunks * kF16Chunk) return;
  const int c = idx / kF16Chunk, kk = idx % kF16Chunk;
  const int n = idx;
  const bool real = n < N;
  const GramHdr h = *hdr;
  unsigned char* p = img + (size_t)c * gram_chunk_bytes(NT);
  const int ct = kk >> 4, row = kk & 15;                     // centre tile, A-operand row
  _Float16* head = reinterpret_cast<_Float16*>(p + ct * 512);                  // lane (g, row): k = 4 g + j
  auto put_head = [&](int s, double v, int T) { head[((s >> 2) * 16 + row) * 4 + (s & 3)] = (_Float16)(float)(v * gram_pow2(T - gram_ax(T))); };
  auto put_tail = [&](int s, double v, int T) {
    const int half = s >> 5, g = (s >> 3) & 3, j = s & 7;
    _Float16* tail = reinterpret_cast<_Float16*>(p + kGramHeadBytes + (ct * 2 + half) * 1024);
    tail[(g * 16 + row) * 8 + j] = (_Float16)(float)(v * gram_pow2(T - gram_ax(T)));
  };
  double alpha = 0.0, beta = 0.0;
  if (h.ok) {
    const double s2 = real ? (double)expf(-2.0f * log_sigs[n]) : 1.0;     // 1/sigma^2 as K1 / K1h have it (float32, flax_rbf.py:280); every
    gram_alpha_beta<BC>(s2, gscale, alpha, beta);                          // term of the expansion uses this one value
    if (!real) alpha = 0.0;                                  // a padding centre: u = beta (P finite), its W rows are 0
  }
  if (part < kGramDims) {
    double nC[3] = {0.0, 0.0, 0.0};
    if (real && h.ok && part < D) gram_parts_d(-2.0 * alpha * ((double)centers[(size_t)n * D + part] - (double)h.r[part]), h.ec, nC);
    put_head(part, nC[0], h.ex + h.ec);
    for (int m = 0; m < 5; ++m) {
      const int q = gram_comb_q(m);
      put_tail(5 * part + m, nC[q], h.ex + h.ec - 11 * (gram_comb_p(m) + q));
    }
  } else if (part == kGramDims) {
    double nA[3] = {0.0, 0.0, 0.0}, n2[4] = {0.0, 0.0, 0.0, 0.0};
    if (h.ok) {
      double c2 = beta;
      if (real)
        for (int i = 0; i < D && i < kGramDims; ++i) {
          const double cp = (double)centers[(size_t)n * D + i] - (double)h.r[i];
          c2 += alpha * cp * cp;
        }
      gram_parts_d(alpha, h.ea, nA);
      gram_parts_c2(c2, h.e2, n2);
    }
    put_head(kGramDims, nA[0], h.eq + h.ea);
    put_head(kGramDims + 1, n2[0], h.e2);
    put_head(kGramDims + 2, n2[1], h.e2 - 11);
    for (int s = kGramDims + 3; s < 16; ++s) put_head(s, 0.0, 0);
    for (int m = 0; m < 5; ++m) {
      const int q = gram_comb_q(m);
      put_tail(5 * kGramDims + m, nA[q], h.eq + h.ea - 11 * (gram_comb_p(m) + q));
    }
    put_tail(5 * kGramDims + 5, n2[2], h.e2 - 22);
    put_tail(5 * kGramDims + 6, n2[3], h.e2 - 33);
    for (int s = 5 * kGramDims + 7; s < 64; ++s) put_tail(s, 0.0, 0);
  }
  // W rows in the k order of the Phi x W product: centre 16 ct + 4 g + r <-> k = 8 g + 4 ct + r
  const int g = row >> 2, j = ct * 4 + (row & 3);
  for (int wt = 0; wt < NT; ++wt) {                          // column tiles of 16 outputs: W hi, W lo
    _Float16* wh = reinterpret_cast<_Float16*>(p + kGramOpBytes + (size_t)wt * 2 * kF16WBytes);
    _Float16* wl = wh + kF16WBytes / 2;
    const int oo = part, o = wt * 16 + oo;
    float w = 0.0f;
    if (real && o < O) w = kernel[(size_t)(n % K) * O + o] / oscale[o];
    _Float16 hh, ll;
    split_static_f16(w, hh, ll);
    wh[(g * 16 + oo) * 8 + j] = hh;
    wl[(g * 16 + oo) * 8 + j] = ll;
  }
}

// ---- kernel ----------------------------------------------------------------------------------------------
template <int DC, int BC, bool ROLL>
__device__ __forceinline__ void gram_body(const GramArgs& ga, const F16Roll& rl, int mode, unsigned char* lds) {
  static_assert(DC <= kGramDims, "eight coordinate slots");
  const F16Args& a = ga.f;
  constexpr int CBL = f16_chunk_bytes(DC);                   // K1h's chunk image (the VALU path reads its records)
  constexpr int CB = kGramChunkBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, QG = a.QG;
  const int slice = wave / QG, qg = wave % QG;               // the QG waves of a slice are adjacent and share its ring
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * 32;
  const GramHdr* hp = ga.hdr;
  long qrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    qrow[t] = q < 0 ? 0 : q;
  }
  // ---- query-side operands: B[k = slot][column = query]
  h4_t bhd[2];
  h8_t btl[2][2];
  const bool bad = gram_query_operands<DC>(a, hp, qrow, g, bhd, btl);
  const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;     // wave-uniform: the VALU distances for these 32 queries

  const int c0 = (int)((long)a.nchunks * slice / S), c1 = (int)((long)a.nchunks * (slice + 1) / S);
  const int na = c1 - c0;
  int nsteps = 0;                                            // every wave of the block walks the longest slice (barriers)
  for (int s2 = 0; s2 < S; ++s2) {
    const int m = (int)((long)a.nchunks * (s2 + 1) / S) - (int)((long)a.nchunks * s2 / S);
    nsteps = m > nsteps ? m : nsteps;
  }
  // Ring of kGramRing chunk images per slice: during step i the waves read the distance operands of chunk i + 1 and the W
  // operands of chunk i while later chunks land (end_of_step below).
  unsigned char* ring = lds + (size_t)slice * kGramRing * CB;
  constexpr int NVI = CB / 1024;                             // 7 wave-instructions per chunk image
  auto stage = [&](int k, int buf) {                         // chunk c0 + k of the slice -> ring slot buf; the QG waves share the copy
    if (k >= na) return;
    const unsigned char* gp = ga.gimg + (size_t)(c0 + k) * CB + lane * 16;
    unsigned char* dst = ring + buf * CB;
    for (int v = qg; v < NVI; v += QG)
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto next3 = [&](int b3) { return b3 == kGramRing - 1 ? 0 : b3 + 1; };
  // A step touches chunks i (W) and i + 1 (distance operands).  One barrier per kGramPer = (kGramRing - 1) / 2 steps: behind it chunks
  // i + 1 .. i + kGramPer + 1 are resident and the next kGramPer are requested into the slots of the chunks everybody has left
  // (kGramRing = 3: a barrier per chunk; 5: one per two chunks).
  constexpr int kGramPer = (kGramRing - 1) / 2;
  static_assert(kGramRing == 2 * kGramPer + 1 && kGramPer >= 1, "ring = 2 x (steps per barrier) + 1");
  auto end_of_step = [&](int i, int b0) {
    if ((i % kGramPer) != kGramPer - 1) return;
    step_barrier();
#pragma unroll
    for (int j = 0; j < kGramPer; ++j) {
      int slot = b0 - (kGramPer - 1) + j;                    // slot of chunk i - (kGramPer - 1) + j
      slot = slot < 0 ? slot + kGramRing : slot;
      stage(i + kGramPer + 2 + j, slot);
    }
  };

  f4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A1: ph * wh
  f4_t acl[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A2: pls * wh + ph * wls
  auto distances = [&](const unsigned char* buf, f4_t (&u)[2][2]) { gram_distances(buf, lane, bhd, btl, u); };
  // transcendental, hi / lo split and Phi x W of the 16 pairs in t16 with the W operands of chunk `buf`
  auto products = [&](float (&t16)[16], const unsigned char* buf, auto pre) {
    const h8_t bh = *reinterpret_cast<const h8_t*>(buf + kGramOpBytes + lane * 16);
    const h8_t bl = *reinterpret_cast<const h8_t*>(buf + kGramOpBytes + kF16WBytes + lane * 16);
    pre(t16);                                                // P = 2^kPhiExp * phi for the step's 16 pairs
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned wh[4], wl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) split_pair_mix(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
      const h8_t ah = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
      const h8_t al = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
      acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acl[t], 0, 0, 0);
      acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acl[t], 0, 0, 0);
    }
  };

#pragma unroll
  for (int k = 0; k <= kGramPer; ++k) stage(k, k);
  step_barrier();                                            // the first kGramPer + 1 chunks are there
#pragma unroll
  for (int k = kGramPer + 1; k < kGramRing; ++k) stage(k, k);
  if (!wave_bad) {
    // two steps per trip: the distances of chunk i + 1 are issued in front of the VALU work on chunk i
    f4_t ua[2][2], ub[2][2];
    if (na > 0) distances(ring, ua);
    // trans16 is inline asm: the hazard recogniser does not see it read MFMA results.  In the loop the 12 distance MFMAs of the
    // next chunk lie in front of it; a slice of ONE chunk reads them right away: explicit wait states, once
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    int b0 = 0;                                              // ring slot of chunk i
    [[maybe_unused]] unsigned long long tph[5] = {0, 0, 0, 0, 0};
    auto one_step = [&](int i, f4_t (&ucur)[2][2], f4_t (&unxt)[2][2]) {
      const int b1 = next3(b0);
      [[maybe_unused]] const unsigned long long t0 = IRBFN_GRAM_T();
      if (i + 1 < na) distances(ring + b1 * CB, unxt);       // issued in front of the VALU work on chunk i
      if (i < na) {
        float t16[16];
        products(t16, ring + b0 * CB, [&](float (&o)[16]) { trans16<BC>(ucur, o); });
      }
      [[maybe_unused]] const unsigned long long t2 = IRBFN_GRAM_T();
      end_of_step(i, b0);
      [[maybe_unused]] const unsigned long long t4 = IRBFN_GRAM_T();
#ifdef IRBFN_GRAM_STAMPS
      tph[1] += t2 - t0; tph[2] += t4 - t2; tph[4] += 1;
#endif
      b0 = b1;
    };
    for (int i = 0; i < nsteps; i += 2) {
      one_step(i, ua, ub);
      if (i + 1 < nsteps) one_step(i + 1, ub, ua);
    }
#ifdef IRBFN_GRAM_STAMPS
    if (blockIdx.x < 2 && tid == 0)
      for (int k = 0; k < 5; ++k) g_gram_stamps[blockIdx.x * 8 + k] = tph[k];
#endif
  } else {
    // a query of this wave lies outside the representable box (or is not finite): K1h's distances for its 32 queries,
    // same barriers and the same share of the copies
    int b0 = 0;
    for (int i = 0; i < nsteps; ++i) {
      if (i < na) {
        float t16[16];
        gram_valu_args<DC, BC>(a, qrow, g, reinterpret_cast<const float*>(a.img + (size_t)(c0 + i) * CBL), t16);
        products(t16, ring + b0 * CB, [&](float (&o)[16]) { trans_block<BC, 16>(o); });
      }
      end_of_step(i, b0);
      b0 = next3(b0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], kLoScale, acc[t][r]);   // A1 + 2^-11 A2

  // ---- smooth region gate of the single region (model.py:42-95), one value per query
  const GateTables gt = a.gate;
  float gam[2] = {0.0f, 0.0f};
  if (slice == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float gv = gt.n_ranges > 0 ? 1.0f : 0.0f;              // model.py:70
#pragma unroll
      for (int d = 0; d < DC; ++d)
        if (d < gt.nsplit && gt.n_ranges > 0) {
          const int e = d * gt.max_ranges + gt.dim_ranges[d];
          gv *= gate_factor(a.x[qrow[t] * a.Dreal + d], gt.lo[e], gt.hi[e], gt.delta[d]);
        }
      gam[t] = gv;
    }
  }
  narrow_epilogue<ROLL>(a, rl, mode, lds, acc, gam, S, slice, qg, q0, 1.0f / (gram_phi_scale<BC>() * kWScale));
}

template <int DC, int BC>
__global__ __launch_bounds__(1024, IRBFN_GRAM_WAVES) void rbf_fwd_f16gram(const GramArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  gram_body<DC, BC, false>(ga, F16Roll{}, -1, lds);
}

// the planning tick of a narrow net in one launch (forward + sign flip + roll-out; irbfn_planner.py:203-212)
template <int DC, int BC>
__global__ __launch_bounds__(1024, IRBFN_GRAM_WAVES) void rbf_tick_f16gram(const GramArgs ga, const F16Roll rl, const int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  gram_body<DC, BC, true>(ga, rl, mode, lds);
}

// ---- host side -------------------------------------------------------------------------------------------
static int gram_nt(const irbfn_net* net) { return (net->O + 15) / 16; }

bool gram_eligible(const irbfn_net* net) {
  return f16_eligible(net) && net->DC <= kGramDims;
}

size_t gram_image_bytes(const irbfn_net* net) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  return (size_t)nchunks * gram_chunk_bytes(gram_nt(net));
}

size_t gram_header_bytes() { return sizeof(GramHdr); }

// after K1h's pack (needs its column scales).  Reads the header back (one small synchronous copy): whether this net
// runs on K1g is a property of its parameters.
int launch_pack_gram(irbfn_net* net, const float* centers, const float* log_sigs, const float* kernel, hipStream_t s) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  GramHdr* hdr = reinterpret_cast<GramHdr*>(net->gram_hdr);
  hipLaunchKernelGGL(gram_stats_kernel, dim3(1), dim3(1024), 0, s, centers, log_sigs, hdr, net->N, net->D,
                     net->bclass, gauss_scale(net->basis));
  IRBFN_HIP_CHECK(hipGetLastError());
  const int total = nchunks * kF16Chunk * kGramPackThreads;
  const dim3 grid((total + 255) / 256), block(256);
  const float gs = gauss_scale(net->basis);
  switch (net->bclass) {
    case BC_GAUSS: hipLaunchKernelGGL((gram_pack_kernel<BC_GAUSS>), grid, block, 0, s, centers, log_sigs, kernel, net->f16_oscale, hdr, net->gram_img, net->N, net->K, net->D, net->O, gram_nt(net), gs, nchunks); break;
    case BC_IQ: hipLaunchKernelGGL((gram_pack_kernel<BC_IQ>), grid, block, 0, s, centers, log_sigs, kernel, net->f16_oscale, hdr, net->gram_img, net->N, net->K, net->D, net->O, gram_nt(net), gs, nchunks); break;
    case BC_IMQ: hipLaunchKernelGGL((gram_pack_kernel<BC_IMQ>), grid, block, 0, s, centers, log_sigs, kernel, net->f16_oscale, hdr, net->gram_img, net->N, net->K, net->D, net->O, gram_nt(net), gs, nchunks); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  if (net->opt[IRBFN_OPT_GRAM_STICKY] != 0 && net->gram_checked) return IRBFN_OK;     // the first verdict stands (training loops)
  GramHdr h;
  IRBFN_HIP_CHECK(hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, s));
  IRBFN_HIP_CHECK(hipStreamSynchronize(s));
  net->gram_checked = 1;
  net->gram_ok = h.ok;
  net->gram_exp[0] = h.ex; net->gram_exp[1] = h.ec; net->gram_exp[2] = h.eq; net->gram_exp[3] = h.ea; net->gram_exp[4] = h.e2;
  return IRBFN_OK;
}

#ifdef IRBFN_GRAM_STAMPS
extern "C" int irbfn_debug_gram_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_gram_stamps), sizeof(unsigned long long) * 32);
}
#endif

template <int DC>
static int launch_gram_bc(const GramArgs& a, int bc, int grid, int block, size_t lds, hipStream_t s) {
#define IRBFN_GCASE(BCV)                                                                                      \
  case BCV: {                                                                                                 \
    auto k = rbf_fwd_f16gram<DC, BCV>;                                                                        \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a);                                                \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_GCASE(BC_GAUSS)
    IRBFN_GCASE(BC_IQ)
    IRBFN_GCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_GCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// S centre slices x QG query groups of 32 per block (S * QG <= 8 waves)
int launch_forward_gram(irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, hipStream_t s) {
  if (!net->gram_img || !net->f16_img || !gram_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  if (net->O > 16) return launch_forward_gram_wide(net, x, out, B, S, QG, s);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  if (S < 1 || QG < 1 || S * QG > 16 || S > nchunks) return IRBFN_ERR_BAD_ARG;
  GramArgs a;
  a.f.x = x; a.f.img = net->f16_img; a.f.oscale = net->f16_oscale; a.f.bias = net->bias; a.f.out = out; a.f.gate = net->gate();
  a.f.B = (long)B; a.f.Dreal = net->D; a.f.O = net->O; a.f.nchunks = nchunks; a.f.S = S; a.f.QG = QG;
  a.gimg = net->gram_img;
  a.hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
  const int waves = S * QG;
  const size_t ring = (size_t)S * kGramRing * kGramChunkBytes;
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32) * sizeof(float);
  const size_t lds = ring > red ? ring : red;
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_gram_bc<3>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 4: rc = launch_gram_bc<4>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 7: rc = launch_gram_bc<7>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 8: rc = launch_gram_bc<8>(a, net->bclass, grid, waves * 64, lds, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_f16gram<D=%d,BC=%d,S=%d,QG=%d>", net->DC, net->bclass, S, QG);
    net->last_grid = grid;
    net->last_block = waves * 64;
  }
  return rc;
}

template <int DC>
static int launch_tick_gram_bc(const GramArgs& a, const F16Roll& rl, int mode, int bc, int grid, int block, size_t lds, hipStream_t s) {
#define IRBFN_GCASE(BCV)                                                                                      \
  case BCV: {                                                                                                 \
    auto k = rbf_tick_f16gram<DC, BCV>;                                                                       \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a, rl, mode);                                      \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_GCASE(BC_GAUSS)
    IRBFN_GCASE(BC_IQ)
    IRBFN_GCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_GCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// The one-launch planning tick of a narrow net on K1g (rbf_tick_f16gram): same conditions as K1h's (launch_tick_f16_narrow),
// and the parameters fit the expansion.  IRBFN_ERR_UNSUPPORTED: no instance -> the caller takes another path
int launch_tick_gram_narrow(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0, const DynParams& dp,
                            float* controls, float* states, int64_t B, int T, hipStream_t s) {
  if (net->opt[IRBFN_OPT_FWD_KERNEL] != IRBFN_FWD_AUTO && net->opt[IRBFN_OPT_FWD_KERNEL] != IRBFN_FWD_K1G) return IRBFN_ERR_UNSUPPORTED;
  if (!gram_preferred(net, B)) return IRBFN_ERR_UNSUPPORTED;
  if (net->opt[IRBFN_OPT_TICK_FUSED] == 0 || net->O != 2 * T || T > kTickNarrowT) return IRBFN_ERR_UNSUPPORTED;
  const bool st = mode == IRBFN_ROLLOUT_ST_SELECT || mode == IRBFN_ROLLOUT_ST_KS || mode == IRBFN_ROLLOUT_FULLINT;
  if (!((st && net->DC == 7) || (mode == IRBFN_ROLLOUT_FRENET_LS && net->DC == 8))) return IRBFN_ERR_UNSUPPORTED;
  int S, QG;
  gram_geometry(net, B, &S, &QG);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  const int waves = S * QG;
  const size_t ring = (size_t)S * kGramRing * kGramChunkBytes;
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32 + (size_t)QG * 32 * (kTickNarrowCP + kTickNarrowSP)) * sizeof(float);
  const size_t lds = ring > red ? ring : red;
  if (lds > 160 * 1024 || S > nchunks) return IRBFN_ERR_UNSUPPORTED;
  GramArgs a;
  a.f.x = x; a.f.img = net->f16_img; a.f.oscale = net->f16_oscale; a.f.bias = net->bias; a.f.out = controls; a.f.gate = net->gate();
  a.f.B = (long)B; a.f.Dreal = net->D; a.f.O = net->O; a.f.nchunks = nchunks; a.f.S = S; a.f.QG = QG;
  a.gimg = net->gram_img;
  a.hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
  F16Roll rl;
  rl.state0 = state0; rl.states = states; rl.mirror = mirror; rl.T = T; rl.wlds = 0; rl.dp = dp;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  const int rc = net->DC == 7 ? launch_tick_gram_bc<7>(a, rl, mode, net->bclass, grid, waves * 64, lds, s)
                              : launch_tick_gram_bc<8>(a, rl, mode, net->bclass, grid, waves * 64, lds, s);
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_tick_f16gram<D=%d,BC=%d,MODE=%d,S=%d,QG=%d>", net->DC, net->bclass, mode, S, QG);
    net->last_grid = grid;
    net->last_block = waves * 64;
  }
  return rc;
}

}  // namespace irbfn
