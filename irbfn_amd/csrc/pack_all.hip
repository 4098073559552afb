// K0: irbfn_net_set_params -- the parameter pytree (flax_rbf.py:258-285: centers, log_sigs; model.py:169-198: the Dense kernel and
// bias) into every image the kernels of this net read, in TWO launches.  A training loop re-binds the parameters at every step
// (train_nmpc.py:258-300: the optimiser returns new leaves), so the pack is part of the step: six launches (records, K1m records,
// column scales, K1h images, K1g statistics, K1g images) took 72 us of the 405 us of a config-3 step.
//
//   level 0 (nothing but the parameters needed), roles by block range:
//     R  records of K1 / K2 (rec, sig2, bias)          one thread per centre
//     M  records of K1m                                 one thread per centre
//     C  power-of-two column scales of W (K1h / K1g)    one block per output column
//     S  K1g statistics, partial: per block of 1024 centres the per-coordinate max and -min, the largest |alpha|, finiteness
//     P  the centre and weight tables of the region-sparse kernels K1r / K2r (rbf_sparse.hip)
//   level 1 (needs C and S), roles by block range:
//     F  K1h chunk images                               one thread per centre
//     G  K1g chunk images                               one block per chunk of 32 centres: every block folds the partial statistics
//        into the header itself (a few hundred bytes; block 0 stores it for the kernels), builds its chunk image in LDS and writes
//        it out in whole 16-byte pieces -- 2-byte stores into lines shared with a block on another XCD made the old pack 31 us.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "rbf_forward_gram.h"

namespace irbfn {

constexpr int kPackBlock = 256;
constexpr int kStatsCentres = 1024;                          // centres per S block
constexpr int kStatsVals = 18;                               // [0, 8) max_i, [8, 16) -min_i, 16 max |alpha|, 17 not finite
constexpr int kGramLdsMax = gram_chunk_bytes(8);             // O <= 128

struct PackArgs {
  const float* __restrict__ centers;
  const float* __restrict__ log_sigs;
  const float* __restrict__ kernel;
  const float* __restrict__ bias;
  float* __restrict__ rec;
  float* __restrict__ bias_out;
  float* __restrict__ sig2;
  float* __restrict__ recm;
  float* __restrict__ oscale;
  unsigned char* __restrict__ f16_img;
  unsigned char* __restrict__ gram_img;
  GramHdr* __restrict__ hdr;
  float* __restrict__ part;                                  // [nbS][kStatsVals]
  int N, Npad, K, D, DC, O, OP, S, bclass;
  float gscale;
  int CW, OW;                                                // K1m record: centre words, weight words
  int RF, NT, nchunks;                                       // K1h / K1g chunk images
  int nbR, nbM, nbC, nbS, nbP, nbF, nbG;
  float* __restrict__ sp_ctab;                               // K1r / K2r: [n_ranges][RS] centre table, [K][WP] weight rows
  float* __restrict__ sp_wtab;
  int sp_nr, sp_EW, sp_RS, sp_WP;
};

// ---- level 0 ---------------------------------------------------------------------------------------------
// rec[n] = { c[0..DC), scale, W[k][0..OP) }, n = r K + k: the weight row is replicated per region so that the hot loop reads
// ONE contiguous scalar stream
__device__ __forceinline__ void records_body(const PackArgs& a, int vb) {
  const int n = vb * kPackBlock + threadIdx.x;
  if (n < a.OP) a.bias_out[n] = n < a.O ? a.bias[n] : 0.0f;
  if (n >= a.N) return;
  const int k = n % a.K;
  float* r = a.rec + (size_t)n * a.S;
  for (int j = 0; j < a.DC; ++j) r[j] = j < a.D ? a.centers[(size_t)n * a.D + j] : 0.0f;
  const float s2 = expf(-2.0f * a.log_sigs[n]);              // 1/sigma^2, sigma = exp(log_sig) (flax_rbf.py:280)
  a.sig2[n] = s2;
  r[a.DC] = a.bclass == BC_GAUSS ? -a.gscale * 1.4426950408889634f * s2 : s2;
  for (int o = 0; o < a.OP; ++o) r[a.DC + 1 + o] = o < a.O ? a.kernel[(size_t)k * a.O + o] : 0.0f;
  for (int j = a.DC + 1 + a.OP; j < a.S; ++j) r[j] = 0.0f;
}

// recm[n] = { c[0..D), scale at [D], zeros to CW, W[k][0..OW) zero padded }; centres n >= N (padding to the MFMA chunk): zeros
__device__ __forceinline__ void mfma_records_body(const PackArgs& a, int vb) {
  const int n = vb * kPackBlock + threadIdx.x;
  if (n >= a.Npad) return;
  float* r = a.recm + (size_t)n * (a.CW + a.OW);
  if (n >= a.N) {
    for (int j = 0; j < a.CW + a.OW; ++j) r[j] = 0.0f;
    return;
  }
  const int k = n % a.K;
  for (int j = 0; j < a.CW; ++j) r[j] = j < a.D ? a.centers[(size_t)n * a.D + j] : 0.0f;
  const float s2 = expf(-2.0f * a.log_sigs[n]);
  r[a.D] = a.bclass == BC_GAUSS ? -a.gscale * 1.4426950408889634f * s2 : s2;
  for (int o = 0; o < a.OW; ++o) r[a.CW + o] = o < a.O ? a.kernel[(size_t)k * a.O + o] : 0.0f;
}

// max over the block of NV values per thread (NV <= kStatsVals): wave shuffles, one LDS pass; thread t < NV returns value t
template <int NV>
__device__ __forceinline__ float block_max(float (&v)[NV], float (&red)[kPackBlock / 64][kStatsVals]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[i] = fmaxf(v[i], __shfl_xor(v[i], off));
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave][i] = v[i];
  }
  __syncthreads();
  float m = 0.0f;
  if (tid < NV) {
    m = red[0][tid];
    for (int w = 1; w < kPackBlock / 64; ++w) m = fmaxf(m, red[w][tid]);
  }
  return m;
}

// the power of two above the largest |W[:, o]| (zero column / Inf / NaN: unscaled)
__device__ __forceinline__ void colscale_body(const PackArgs& a, int vb, float (&red)[kPackBlock / 64][kStatsVals]) {
  const int o = vb;
  float m[1] = {0.0f};
  if (o < a.O)
    for (int k = threadIdx.x; k < a.K; k += kPackBlock) m[0] = fmaxf(m[0], fabsf(a.kernel[(size_t)k * a.O + o]));
  // NaN: fmaxf drops it -- as the serial maximum before; an Inf is kept
  const float mx = block_max<1>(m, red);
  if (threadIdx.x == 0) {
    float s = 1.0f;
    if (mx > 0.0f && mx < 3.0e38f) {
      int e;
      (void)frexpf(mx, &e);                                  // mx = f 2^e, f in [0.5, 1)  ->  2^e > mx
      s = ldexpf(1.0f, e);
    }
    a.oscale[o] = s;
  }
}

__device__ inline void gram_alpha_beta(int bclass, double s2, double gscale, double& alpha, double& beta) {
  if (bclass == BC_GAUSS) { alpha = -gscale * 1.4426950408889634 * s2; beta = (double)kPhiExp; }     // P = 2^(alpha d2 + 14)
  else { alpha = s2 * (double)kPhiInv; beta = (double)kPhiInv; }   // IQ: P = 1 / (2^-14 (1 + d2 s2)); IMQ: P = rsqrt(same) = 2^7 phi
}

// ONE pass over the centres: per coordinate max and -min (origin = midpoint, half widths), the largest |alpha|, finiteness.  The
// bounds on C = -2 alpha c' and on c2 = alpha |c'|^2 + beta are products of these maxima (at most the true maxima x the spread
// of alpha over the centres: a coarser grid where it matters, never a wrong one) -- a second pass over the centres with the
// origin known would make them tight at twice the cost, and this runs at every irbfn_net_set_params.
__device__ __forceinline__ void stats_body(const PackArgs& a, int vb, float (&red)[kPackBlock / 64][kStatsVals]) {
  float v[kStatsVals];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = -3.0e38f;
  v[16] = 0.0f; v[17] = 0.0f;
  const int k1 = (vb + 1) * kStatsCentres < a.N ? (vb + 1) * kStatsCentres : a.N;
  for (int k = vb * kStatsCentres + threadIdx.x; k < k1; k += kPackBlock) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < a.D) {
        const float c = a.centers[(size_t)k * a.D + i];
        if (!(fabsf(c) < 3.0e38f)) v[17] = 1.0f;
        v[i] = fmaxf(v[i], c);
        v[8 + i] = fmaxf(v[8 + i], -c);
      }
    const double s2 = (double)expf(-2.0f * a.log_sigs[k]);   // 1/sigma^2 as K1 / K1h have it (float32, flax_rbf.py:280)
    double alpha, beta;
    gram_alpha_beta(a.bclass, s2, a.gscale, alpha, beta);
    const float fa1 = (float)fabs(alpha);
    if (!(fa1 < 1.0e30f)) v[17] = 1.0f;
    v[16] = fmaxf(v[16], fa1);
  }
  const float m = block_max<kStatsVals>(v, red);
  if (threadIdx.x < kStatsVals) a.part[(size_t)vb * kStatsVals + threadIdx.x] = m;
}

// ctab[r][k] = { c[0..DC), folded width scale (as rec[DC] of the dense kernels), 0.. }, a zero padding slot behind each region's K
// entries; wtab[k] = W[k, 0..WP)
__device__ __forceinline__ void sparse_tables_body(const PackArgs& a, int vb) {
  const int n = vb * kPackBlock + threadIdx.x;
  if (n < a.K * a.sp_WP) {
    const int k = n / a.sp_WP, o = n - k * a.sp_WP;
    a.sp_wtab[n] = o < a.O ? a.kernel[(size_t)k * a.O + o] : 0.0f;
  }
  const int pad = a.sp_RS - a.K * a.sp_EW;
  if (n < a.sp_nr * pad) {
    const int r = n / pad, j = n - r * pad;
    a.sp_ctab[(size_t)r * a.sp_RS + a.K * a.sp_EW + j] = 0.0f;
  }
  if (n >= a.sp_nr * a.K) return;
  const int r = n / a.K, k = n - r * a.K;
  float* dst = a.sp_ctab + (size_t)r * a.sp_RS + (size_t)k * a.sp_EW;
  for (int j = 0; j < a.sp_EW; ++j) dst[j] = (j < a.D) ? a.centers[(size_t)n * a.D + j] : 0.0f;
  const float s2 = expf(-2.0f * a.log_sigs[n]);
  dst[a.DC] = a.bclass == BC_GAUSS ? -a.gscale * 1.4426950408889634f * s2 : s2;
}

__global__ __launch_bounds__(kPackBlock) void pack_level0_kernel(const PackArgs a) {
  __shared__ float red[kPackBlock / 64][kStatsVals];
  int vb = blockIdx.x;
  if (vb < a.nbR) { records_body(a, vb); return; }
  vb -= a.nbR;
  if (vb < a.nbM) { mfma_records_body(a, vb); return; }
  vb -= a.nbM;
  if (vb < a.nbC) { colscale_body(a, vb, red); return; }
  vb -= a.nbC;
  if (vb < a.nbS) { stats_body(a, vb, red); return; }
  vb -= a.nbS;
  if (vb < a.nbP) sparse_tables_body(a, vb);
}

// ---- level 1 ---------------------------------------------------------------------------------------------
// one thread per (chunk, centre-in-chunk): record + this centre's 16 weights of both parts
__device__ __forceinline__ void f16_image_body(const PackArgs& a, int vb) {
  const int idx = vb * kPackBlock + threadIdx.x;
  if (idx >= a.nchunks * kF16Chunk) return;
  const int c = idx / kF16Chunk, kk = idx % kF16Chunk;
  const int n = idx;                                         // centre index (R == 1: n == k)
  const int RF = a.RF, NT = a.NT;
  const size_t cb = (size_t)kF16Chunk * RF * 4 + (size_t)NT * 2 * kF16WBytes + (NT == 1 ? kF16WBytes : 0);
  unsigned char* p = a.f16_img + (size_t)c * cb;
  float* rec = reinterpret_cast<float*>(p) + kk * RF;
  const bool real = n < a.N;
  for (int j = 0; j < RF - 1; ++j) rec[j] = (real && j < a.D) ? a.centers[(size_t)n * a.D + j] : 0.0f;
  float sc = 0.0f;                                           // padding centre: P = 2^kPhiExp exactly, W = 0
  if (real) {
    const float s2 = expf(-2.0f * a.log_sigs[n]);            // 1/sigma^2 (flax_rbf.py:280)
    if (a.bclass == BC_GAUSS) sc = -a.gscale * 1.4426950408889634f * s2;   // P = 2^(r2*sc + kPhiExp)
    else if (a.bclass == BC_IQ) sc = s2 * kPhiInv;                         // P = 1 / (2^-kPhiExp (1 + d2))
    else sc = s2 * kPhiInv * kPhiInv;                                      // P = rsqrt(2^-2kPhiExp (1 + d2))
  }
  rec[RF - 1] = sc;
  const int g = kk >> 3, j = kk & 7;
  for (int ct = 0; ct < NT; ++ct) {
    _Float16* wh = reinterpret_cast<_Float16*>(p + (size_t)kF16Chunk * RF * 4 + (size_t)ct * 2 * kF16WBytes);
    _Float16* wl = wh + kF16WBytes / 2;
    for (int oo = 0; oo < 16; ++oo) {
      const int o = ct * 16 + oo;
      float w = 0.0f;                                        // W / s_o, |.| < 1 (exact scaling)
      if (real && o < a.O) w = a.kernel[(size_t)(n % a.K) * a.O + o] / a.oscale[o];
      _Float16 h, l;
      split_static_f16(w, h, l);
      wh[(g * 16 + oo) * 8 + j] = h;
      wl[(g * 16 + oo) * 8 + j] = l;
      if (NT == 1) reinterpret_cast<__bf16*>(wl + kF16WBytes / 2)[(g * 16 + oo) * 8 + j] = (__bf16)(w * kWScale);
    }
  }
}

__device__ inline double gram_pow2(int e) {                  // 2^e, -1022 <= e <= 1023, without the library's ldexp
  return __builtin_bit_cast(double, (unsigned long long)(1023 + e) << 52);
}
__device__ inline int gram_exp_above(double v) {            // smallest e with |v| < 2^e: v = f 2^e, f in [0.5, 1) (frexp's e)
  if (!(v > 0.0)) return -40;
  const int be = (int)((__builtin_bit_cast(unsigned long long, v) >> 52) & 0x7ff);
  if (be == 0) return -1022;                                 // subnormal: far below anything the checks below accept
  if (be == 0x7ff) return 1025;                              // Inf: refused by the range checks
  return be - 1022;
}

// origin, exponents and the exactness budget of the expansion from the folded statistics tot[kStatsVals]
__device__ inline GramHdr gram_header(const float* tot, int D, int bclass, float gscale) {
  GramHdr h;
  float hw2 = 0.0f, hwm = 0.0f;                              // sum of the squared half widths, largest half width
  for (int i = 0; i < 8; ++i) {
    h.r[i] = i < D ? 0.5f * tot[i] - 0.5f * tot[8 + i] : 0.0f;
    const float hw = i < D ? fmaxf(tot[i] - h.r[i], h.r[i] + tot[8 + i]) * 1.0000005f : 0.0f;     // the rounded midpoint's two sides
    hw2 += hw * hw;
    hwm = fmaxf(hwm, hw);
  }
  double beta0, alpha0;
  gram_alpha_beta(bclass, 1.0, gscale, alpha0, beta0);
  const float fc = hwm, fa = tot[16], fC = 2.0f * tot[16] * hwm, f2 = tot[16] * hw2 + (float)fabs(beta0), bad = tot[17];
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
  const double xb = gram_pow2(h.ex);                         // -40 <= ex <= 128 here (float inputs)
  const double worst = ((double)fa * D * xb * xb + (double)D * fC * xb + (double)f2) * (1.0 + 1.0 / 256.0);
  while (h.ec < 40 && worst >= gram_pow2(h.ex + h.ec + 4 < -1000 ? -1000 : (h.ex + h.ec + 4 > 1000 ? 1000 : h.ex + h.ec + 4))) ++h.ec;
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
  return h;
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

// One block per chunk; sixteen work items per centre-in-chunk: item `part` fills head slot `part` and the tail slots 4 part .. 4 part + 3
// (rbf_forward_gram.h: the slot tables say which product sits where) and the W values of output `part` of every column tile.
__device__ __forceinline__ void gram_image_body(const PackArgs& a, int c, unsigned char* simg, GramHdr& hs, float* tot) {
  const int tid = threadIdx.x;
  if (tid < kStatsVals) {
    float m = a.part[tid];
    for (int b = 1; b < a.nbS; ++b) m = fmaxf(m, a.part[(size_t)b * kStatsVals + tid]);
    tot[tid] = m;
  }
  __syncthreads();
  if (tid == 0) {
    hs = gram_header(tot, a.D, a.bclass, a.gscale);
    if (c == 0) *a.hdr = hs;
  }
  __syncthreads();
  const GramHdr h = hs;
  const int NT = a.NT, D = a.D;
  const int CB = gram_chunk_bytes(NT);
  unsigned char* p = simg;
  for (int it = tid; it < kF16Chunk * 16; it += kPackBlock) {
    const int part = it >> 5, kk = it & 31;                  // a wave = two parts x 32 centres: (nearly) one path per wave
    const int n = c * kF16Chunk + kk;
    const bool real = n < a.N;
    const int ct = kk >> 4, row = kk & 15;                   // centre tile, A-operand row
    _Float16* head = reinterpret_cast<_Float16*>(p + ct * 512);                // lane (g, row): k = 4 g + j
    auto put_head = [&](int s, double v, int T) { head[((s >> 2) * 16 + row) * 4 + (s & 3)] = (_Float16)(float)(v * gram_pow2(T - gram_ax(T))); };
    auto put_tail = [&](int s, double v, int T) {
      const int half = s >> 5, g = (s >> 3) & 3, j = s & 7;
      _Float16* tail = reinterpret_cast<_Float16*>(p + kGramHeadBytes + (ct * 2 + half) * 1024);
      tail[(g * 16 + row) * 8 + j] = (_Float16)(float)(v * gram_pow2(T - gram_ax(T)));
    };
    double alpha = 0.0, beta = 0.0;
    if (h.ok) {
      const double s2 = real ? (double)expf(-2.0f * a.log_sigs[n]) : 1.0;   // 1/sigma^2 as K1 / K1h have it (float32, flax_rbf.py:280);
      gram_alpha_beta(a.bclass, s2, a.gscale, alpha, beta);                 // every term of the expansion uses this one value
      if (!real) alpha = 0.0;                                // a padding centre: u = beta (P finite), its W rows are 0
    }
    // centre-side factor of a slot's product: part q of C_dim = -2 alpha c'_dim, part q of alpha, part p of c2 = alpha |c'|^2 + beta
    auto centre_value = [&](const GramSlot sl) -> double {
      if (!h.ok || sl.kind == 0) return 0.0;
      if (sl.kind == 1) {
        double nC[3] = {0.0, 0.0, 0.0};
        if (real && sl.dim < D) gram_parts_d(-2.0 * alpha * ((double)a.centers[(size_t)n * D + sl.dim] - (double)h.r[sl.dim]), h.ec, nC);
        return sl.q == 0 ? nC[0] : (sl.q == 1 ? nC[1] : nC[2]);
      }
      if (sl.kind == 2) {
        double nA[3];
        gram_parts_d(alpha, h.ea, nA);
        return sl.q == 0 ? nA[0] : (sl.q == 1 ? nA[1] : nA[2]);
      }
      double c2 = beta, n2[4];
      if (real)
        for (int i = 0; i < D && i < kGramDims; ++i) {
          const double cp = (double)a.centers[(size_t)n * D + i] - (double)h.r[i];
          c2 += alpha * cp * cp;
        }
      gram_parts_c2(c2, h.e2, n2);
      return sl.p == 0 ? n2[0] : (sl.p == 1 ? n2[1] : (sl.p == 2 ? n2[2] : n2[3]));
    };
    {
      const GramSlot sl = gram_head_slot(part);
      put_head(part, centre_value(sl), gram_T(h, sl));
    }
    for (int i = 0; i < 4; ++i) {
      const GramSlot sl = gram_tail_slot(4 * part + i);
      put_tail(4 * part + i, centre_value(sl), gram_T(h, sl));
    }
    // W rows in the k order of the Phi x W product: centre 16 ct + 4 g + r <-> k = 8 g + 4 ct + r
    const int g = row >> 2, j = ct * 4 + (row & 3);
    for (int wt = 0; wt < NT; ++wt) {                        // column tiles of 16 outputs: W hi, W lo
      _Float16* wh = reinterpret_cast<_Float16*>(p + kGramOpBytes + (size_t)wt * 2 * kF16WBytes);
      _Float16* wl = wh + kF16WBytes / 2;
      const int oo = part, o = wt * 16 + oo;
      float w = 0.0f;
      if (real && o < a.O) w = a.kernel[(size_t)(n % a.K) * a.O + o] / a.oscale[o];
      _Float16 hh, ll;
      split_static_f16(w, hh, ll);
      wh[(g * 16 + oo) * 8 + j] = hh;
      wl[(g * 16 + oo) * 8 + j] = ll;
    }
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(a.gram_img + (size_t)c * CB);
  const uint4* src = reinterpret_cast<const uint4*>(simg);
  for (int v = tid; v < CB / 16; v += kPackBlock) dst[v] = src[v];
}

__global__ __launch_bounds__(kPackBlock) void pack_level1_kernel(const PackArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char simg[kGramLdsMax];
  __shared__ GramHdr hs;
  __shared__ float tot[kStatsVals];
  int vb = blockIdx.x;
  if (vb < a.nbF) { f16_image_body(a, vb); return; }
  vb -= a.nbF;
  if (vb < a.nbG) gram_image_body(a, vb, simg, hs, tot);
}

// ---- host ------------------------------------------------------------------------------------------------
size_t pack_partials_bytes(const irbfn_net* net) {
  return (size_t)((net->N + kStatsCentres - 1) / kStatsCentres) * kStatsVals * sizeof(float);
}

// Reads K1g's header back (one small synchronous copy) unless IRBFN_OPT_GRAM_STICKY lets the first verdict stand: whether a net
// runs on K1g / K2g is a property of its parameters.
int launch_pack_all(irbfn_net* net, const float* centers, const float* log_sigs, const float* kernel, const float* bias,
                    hipStream_t s) {
  PackArgs a{};
  a.centers = centers; a.log_sigs = log_sigs; a.kernel = kernel; a.bias = bias;
  a.rec = net->rec; a.bias_out = net->bias; a.sig2 = net->sig2; a.recm = net->recm; a.oscale = net->f16_oscale;
  a.f16_img = net->f16_img; a.gram_img = net->gram_img; a.hdr = reinterpret_cast<GramHdr*>(net->gram_hdr);
  a.part = net->pack_part;
  a.N = net->N; a.Npad = net->Npad; a.K = net->K; a.D = net->D; a.DC = net->DC; a.O = net->O; a.OP = net->OP; a.S = net->S;
  a.bclass = net->bclass; a.gscale = gauss_scale(net->basis);
  a.CW = mfma_cw(net->D); a.OW = 16 * ((net->O + 15) / 16);
  a.RF = f16_rf(net->DC); a.NT = (net->O + 15) / 16;
  a.nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  const int nrec = net->N > net->OP ? net->N : net->OP;
  a.nbR = (nrec + kPackBlock - 1) / kPackBlock;
  a.nbM = net->recm ? (net->Npad + kPackBlock - 1) / kPackBlock : 0;
  a.nbC = net->f16_img ? 16 * a.NT : 0;
  const bool gram = net->gram_img && net->f16_img && net->pack_part;
  a.nbS = gram ? (net->N + kStatsCentres - 1) / kStatsCentres : 0;
  a.nbF = net->f16_img ? (a.nchunks * kF16Chunk + kPackBlock - 1) / kPackBlock : 0;
  a.nbG = gram ? a.nchunks : 0;
  a.nbP = 0;
  if (net->sp_ok) {
    float *ctab = nullptr, *wtab = nullptr;
    sparse_pack_tables(net, &ctab, &wtab, &a.sp_WP);
    a.sp_ctab = ctab; a.sp_wtab = wtab;
    a.sp_nr = net->n_ranges; a.sp_EW = net->sp_EW; a.sp_RS = net->sp_RS;
    int np = a.sp_nr * net->K > net->K * a.sp_WP ? a.sp_nr * net->K : net->K * a.sp_WP;
    const int npad = a.sp_nr * (a.sp_RS - net->K * a.sp_EW);
    np = np > npad ? np : npad;
    a.nbP = (np + kPackBlock - 1) / kPackBlock;
  }
  hipLaunchKernelGGL(pack_level0_kernel, dim3(a.nbR + a.nbM + a.nbC + a.nbS + a.nbP), dim3(kPackBlock), 0, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  if (a.nbF + a.nbG > 0) {
    hipLaunchKernelGGL(pack_level1_kernel, dim3(a.nbF + a.nbG), dim3(kPackBlock), 0, s, a);
    IRBFN_HIP_CHECK(hipGetLastError());
  }
  if (!gram) return IRBFN_OK;
  if (net->opt[IRBFN_OPT_GRAM_STICKY] != 0 && net->gram_checked) return IRBFN_OK;     // the first verdict stands (training loops)
  GramHdr h;
  IRBFN_HIP_CHECK(hipMemcpyAsync(&h, net->gram_hdr, sizeof(h), hipMemcpyDeviceToHost, s));
  IRBFN_HIP_CHECK(hipStreamSynchronize(s));
  net->gram_checked = 1;
  net->gram_ok = h.ok;
  net->gram_exp[0] = h.ex; net->gram_exp[1] = h.ec; net->gram_exp[2] = h.eq; net->gram_exp[3] = h.ea; net->gram_exp[4] = h.e2;
  return IRBFN_OK;
}

}  // namespace irbfn
