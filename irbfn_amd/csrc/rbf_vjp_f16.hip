// K2h: parameter VJP of the single-region RBF net with its two GEMM-shaped pieces on the f16 matrix cores at
// float32 accuracy (hi/lo operand pairs with the lo halves pre-scaled by 2^11 and accumulated separately,
// f16_split.h -- no loss for cotangent rows far below the batch maximum or weights far below their column's):
//     hbar[q,k] = sum_o g[q,o] W[k,o]        (v_mfma_f32_16x16x16_f16: rows = queries, cols = centres, k = outputs)
//     dW[k,o]   = sum_q gamma_q phi[q,k] g[q,o]   (v_mfma_f32_16x16x16_f16: rows = outputs, cols = centres, k = queries)
// and everything else (distances, basis, d centers, d log_sigs; SURVEY App. A.2, jax.value_and_grad at
// scripts/train_nmpc.py:297-298) on the VALU.  Replaces rbf_vjp_kernel (K2) where eligible and writes the
// same slab format, so vjp_reduce_kernel and the bias column sums are shared.
//
// Layout: a wave owns CT tiles of 16 centres and a slice of the 32-query blocks.  Lane (g = l >> 4, n = l & 15)
// holds centre n of every tile (coordinates, scales and gradient accumulators in VGPRs for the whole launch) and,
// per query block, the 8 queries qid_g(p) = 16 (p >> 2) + 4 g + (p & 3): exactly the rows the 16x16x16 MFMA
// returns to this lane (hbar needs no transpose) and, read as k = 8 g + p, a valid B operand of the dW MFMA.
// Query blocks come pre-packed (vjp_pack_blocks_kernel): x and gamma as float rows for broadcast LDS reads,
// g as ready-made f16 (hi, lo) MFMA operands in both orientations.  Both MFMAs are issued per 16-query half:
// its 4 queries per lane are the k-slice 4 g + j of the 16x16x16 dW product as well.
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "f16_split.h"
#include "rbf_forward.h"
#include "rbf_vjp_f16.h"

namespace irbfn {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef __fp16 h2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// ---- pre-pass ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2_ceil_scale(float mx) {
  if (!(mx > 0.0f) || !(mx < 3.0e38f)) return 1.0f;          // zero / Inf / NaN: unscaled
  int e;
  (void)frexpf(mx, &e);
  return ldexpf(1.0f, e);
}

// one 64-lane block per 32-query block
__global__ __launch_bounds__(64) void vjp_pack_blocks_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                             const float* __restrict__ bmax, int nbmax,
                                                             const float* __restrict__ oscale,
                                                             unsigned char* __restrict__ qblk, float* __restrict__ scales,
                                                             GateTables gt, long B, int D, int RFQ, int O, const int* __restrict__ run_if, int run_gen) {
  if (run_if != nullptr && *run_if != run_gen) return;             // K2g did the work
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const long q0 = (long)blockIdx.x * 32;
  const int blkb = 32 * RFQ * 4 + 4096;
  unsigned char* p = qblk + (size_t)blockIdx.x * blkb;
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
  const float sg = pow2_ceil_scale(mx);
  float somax = 0.0f;
  for (int o = 0; o < O; ++o) somax = fmaxf(somax, oscale[o]);
  const float sh = sg * somax;
  if (blockIdx.x == 0 && lane == 0) { scales[0] = sg; scales[1] = sh; }
  // x rows + gate (model.py:42-95, single region)
  float gm = 0.0f;                               // gamma of query q0 + lane (lanes >= 32: 0)
  if (lane < 32) {
    const long q = q0 + lane;
    float* row = reinterpret_cast<float*>(p) + lane * RFQ;
    if (q < B) {
      gm = gt.n_ranges > 0 ? 1.0f : 0.0f;
      for (int d = 0; d < gt.nsplit && gt.n_ranges > 0; ++d) {
        const int e = d * gt.max_ranges + gt.dim_ranges[d];
        gm *= gate_factor(x[q * D + d], gt.lo[e], gt.hi[e], gt.delta[d]);
      }
    }
    for (int j = 0; j < RFQ - 1; ++j) row[j] = (q < B && j < D) ? x[q * D + j] : 0.0f;
    row[RFQ - 1] = gm;
  }
  h4v* gA = reinterpret_cast<h4v*>(p + 32 * RFQ * 4);                // [s][part][lane] x 4 halfs
  // A operand of the hbar MFMA: rows = queries 16 s + n, k = outputs 4 g + j, scaled by s_o / s_h
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const long q = q0 + 16 * s + n;
    h4v hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 4 * g + j;
      float v = 0.0f;
      if (q < B && o < O) v = gout[q * O + o] * (oscale[o] / sh);
      _Float16 h, l;
      split_static_f16(v, h, l);
      hi[j] = h;
      lo[j] = l;
    }
    gA[(s * 2 + 0) * 64 + lane] = hi;
    gA[(s * 2 + 1) * 64 + lane] = lo;
  }
  // A operand of the dW MFMA (16x16x16, one per 16-query half): rows = outputs n, k = 4 g + j <-> query 16 s + 4 g + j
  // (the lane's own 4 queries of that half), scaled by gamma / s_g: d W = (gamma Phi)^T g = Phi^T (gamma g), so the main kernel
  // hands the basis values to the MFMA as they leave the transcendental unit (one multiply per pair less)
  h4v* gT = reinterpret_cast<h4v*>(p + 32 * RFQ * 4 + 2048);         // [s][part][lane] x 4 halfs
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    h4v hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long q = q0 + 16 * s + 4 * g + j;
      const float gq = __shfl(gm, 16 * s + 4 * g + j);
      float v = 0.0f;
      if (q < B && n < O) v = gq * gout[q * O + n] / sg;
      _Float16 h, l;
      split_static_f16(v, h, l);
      hi[j] = h;
      lo[j] = l;
    }
    gT[(s * 2 + 0) * 64 + lane] = hi;
    gT[(s * 2 + 1) * 64 + lane] = lo;
  }
}

// ---- main kernel -------------------------------------------------------------------------------------------
struct VjpHArgs {
  const unsigned char* __restrict__ qblk;   // [nqb][block image]
  const float* __restrict__ scales;         // [0] = s_g, [1] = s_h
  const float* __restrict__ rec;            // [N][S] K1 records: c[DC], scale, W[OP]
  const float* __restrict__ sig2;           // [N]
  const float* __restrict__ oscale;         // [O]
  float* __restrict__ part;                 // [QSB][V][Npad]
  long nqb;
  int O, OP, N, S, Npad, basis, bpw;
  float gscale;
  const int* __restrict__ run_if;           // null, or: return at once unless *run_if == run_gen (K2g did the work)
  int run_gen;
};

#ifndef IRBFN_K2H_MINW
#define IRBFN_K2H_MINW 4       // waves per SIMD the CT = 2 instance is allocated for (128 VGPRs once the ring is filled by LDS-DMA: 314 -> 303 us)
#endif
template <int DC, int BC, int CT>
__global__ __launch_bounds__(256, CT == 2 ? IRBFN_K2H_MINW : 2) void rbf_vjp_f16mfma(const VjpHArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (a.run_if != nullptr && *a.run_if != a.run_gen) return;
  constexpr int RFQ = vjph_rfq(DC);
  constexpr int QXB = 32 * RFQ * 4;
  constexpr int BLKB = QXB + 4096;
  constexpr int NV = BLKB / 16;
  constexpr int NP = (NV + 63) / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int cb = blockIdx.x * 16 * CT;
  const float sg = a.scales[0], sh = a.scales[1];
  // s_h times the constant of dphi/dd2 (gaussian family: -a; inverse quadratic: -1; inverse multiquadric: -1/2)
  // (and 2^-30: both operands of the hbar product carry 2^15)
  // (and 2^-14 per power of P = 2^14 phi in dphi/dd2: the basis values stay in the scale the dW MFMA wants)
  const float shk = sh * (1.0f / (kWScale * kWScale)) * (BC == BC_GAUSS ? -a.gscale * kPhiInv
                                                          : (BC == BC_IQ ? -kPhiInv * kPhiInv : -0.5f * kPhiInv * kPhiInv * kPhiInv));

  float c[CT][DC], sc[CT], s2m2[CT];
  h4v wth[CT], wtl[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int cid = cb + ct * 16 + n;
    cid = cid < a.N ? cid : a.N - 1;
    const float* rp = a.rec + (size_t)cid * a.S;
#pragma unroll
    for (int j = 0; j < DC; ++j) c[ct][j] = rp[j];
    // argument of the transcendental for P = 2^14 phi: gaussian 2^(r2 sc + 14); 1 / (2^-14 (1 + d2)); rsqrt(2^-28 (1 + d2))
    sc[ct] = rp[DC] * (BC == BC_GAUSS ? 1.0f : (BC == BC_IQ ? kPhiInv : kPhiInv * kPhiInv));
    s2m2[ct] = -2.0f * a.sig2[cid];                          // -2 / sigma^2
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 4 * g + j;
      const float w = o < a.O ? rp[DC + 1 + o] / a.oscale[o] : 0.0f;
      _Float16 h, l;
      split_static_f16(w, h, l);
      wth[ct][j] = h;
      wtl[ct][j] = l;
    }
  }
  float gc[CT][DC], gls[CT];
  f4v dW[CT], dWl[CT];                                       // A1, A2 of the dW product (f16_split.h)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    gls[ct] = 0.0f;
    dW[ct] = f4v{0, 0, 0, 0};
    dWl[ct] = f4v{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < DC; ++j) gc[ct][j] = 0.0f;
  }

  const long wslice = (long)blockIdx.y * 4 + wave;
  const long qb0 = wslice * a.bpw;
  long qb1 = qb0 + a.bpw;
  qb1 = qb1 < a.nqb ? qb1 : a.nqb;
  unsigned char* mylds = lds + wave * (2 * BLKB);
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // the next 32-query block lands in the other half of the wave's ring by LDS-DMA (global_load_lds_dwordx4: no
  // staging VGPRs -- the register copy held 20 -- and no ds_write pass) while the current one is worked on
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto request = [&](long qb, unsigned char* dst) {
    const unsigned char* src = a.qblk + (size_t)qb * BLKB + lane * 16;
#pragma unroll
    for (int v = 0; v < NP; ++v)
      if (v * 64 + lane < NV) __builtin_amdgcn_global_load_lds((gptr_t)(src + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  if (qb0 < qb1) request(qb0, mylds);
  landed();
  for (long qb = qb0; qb < qb1; ++qb) {
    const unsigned char* cur = mylds + ((qb - qb0) & 1) * BLKB;
    unsigned char* nxt = mylds + ((qb - qb0 + 1) & 1) * BLKB;
    const bool has_next = qb + 1 < qb1;
    if (has_next) request(qb + 1, nxt);                       // `nxt` was last read one block ago (wave_sync below)
    constexpr int RQ = 4 / CT;                                // queries per inner step: CT * RQ = 4 transcendentals
    // NOT unrolled: the two halves are independent and hipcc would interleave them (286 VGPRs instead of ~140)
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      // hbar / s_h of the half's 16 queries x this wave's centres: independent of phi, so it goes first
      const h4v gah = *reinterpret_cast<const h4v*>(cur + QXB + ((s * 2 + 0) * 64 + lane) * 8);
      const h4v gal = *reinterpret_cast<const h4v*>(cur + QXB + ((s * 2 + 1) * 64 + lane) * 8);
      f4v hb[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f4v hl = f4v{0, 0, 0, 0};
        hb[ct] = f4v{0, 0, 0, 0};
        hb[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(gah, wth[ct], hb[ct], 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_16x16x16f16(gal, wth[ct], hl, 0, 0, 0);
        hl = __builtin_amdgcn_mfma_f32_16x16x16f16(gah, wtl[ct], hl, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[ct][r] = __builtin_fmaf(hl[r], kLoScale, hb[ct][r]);
      }
      float hq[CT][4];                                        // 2^kPhiExp gamma phi of this lane's 4 queries of the half
#pragma unroll
      for (int r0 = 0; r0 < 4; r0 += RQ) {
        float diff[RQ][CT][DC], r2[RQ][CT], t[RQ * CT], kq[RQ];
#pragma unroll
        for (int u = 0; u < RQ; ++u) {
          const float* xr = reinterpret_cast<const float*>(cur) + (16 * s + 4 * g + r0 + u) * RFQ;
          float xv[RFQ];
#pragma unroll
          for (int v = 0; v < RFQ / 4; ++v) {
            const f4v rr = *reinterpret_cast<const f4v*>(xr + 4 * v);
            xv[4 * v] = rr.x; xv[4 * v + 1] = rr.y; xv[4 * v + 2] = rr.z; xv[4 * v + 3] = rr.w;
          }
          kq[u] = xv[RFQ - 1] * shk;                          // gamma * s_h * (basis constant of dphi/dd2)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < DC; ++j) {
              diff[u][ct][j] = xv[j] - c[ct][j];
              acc = __builtin_fmaf(diff[u][ct][j], diff[u][ct][j], acc);
            }
            r2[u][ct] = acc;
            t[u * CT + ct] = BC == BC_GAUSS ? __builtin_fmaf(acc, sc[ct], (float)kPhiExp)
                                            : __builtin_fmaf(acc, sc[ct], BC == BC_IQ ? kPhiInv : kPhiInv * kPhiInv);
          }
        }
        trans_block<BC, RQ * CT>(t);                          // phi of the step's 4 pairs
#pragma unroll
        for (int u = 0; u < RQ; ++u)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const int r = r0 + u;
            const float phi = t[u * CT + ct];                 // P = 2^14 phi
            hq[ct][r] = phi;
            // tt = hbar * gamma * dphi/dd2:  gaussian -a phi | IQ -phi^2 | IMQ -phi^3 / 2  (constants folded into kq)
            float pw = phi;
            if constexpr (BC == BC_IQ) pw = phi * phi;
            if constexpr (BC == BC_IMQ) pw = phi * phi * phi;
            // the centre's own factor -2 / sigma^2 (s2m2) multiplies the finished sums, not every pair
            const float tt = hb[ct][r] * kq[u] * pw;
            gls[ct] = __builtin_fmaf(tt, r2[u][ct], gls[ct]);                  // x s2m2 -> tt * (-2 d2)
#pragma unroll
            for (int j = 0; j < DC; ++j) gc[ct][j] = __builtin_fmaf(tt, diff[u][ct][j], gc[ct][j]);   // x s2m2 -> -2 tt / sigma^2
          }
      }
      // dW of the half: rows = outputs, cols = centres, k = the lane's 4 queries
      const h4v gth = *reinterpret_cast<const h4v*>(cur + QXB + 2048 + ((s * 2 + 0) * 64 + lane) * 8);
      const h4v gtl = *reinterpret_cast<const h4v*>(cur + QXB + 2048 + ((s * 2 + 1) * 64 + lane) * 8);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        unsigned bh2[2], bl2[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) split_pair_f16<3>(hq[ct][2 * jj], hq[ct][2 * jj + 1], bh2[jj], bl2[jj]);
        const h4v bh = __builtin_bit_cast(h4v, u2v{bh2[0], bh2[1]});
        const h4v bl = __builtin_bit_cast(h4v, u2v{bl2[0], bl2[1]});
        dW[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(gth, bh, dW[ct], 0, 0, 0);
        dWl[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(gtl, bh, dWl[ct], 0, 0, 0);
        dWl[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(gth, bl, dWl[ct], 0, 0, 0);
      }
    }
    wave_sync();                                              // this block's LDS reads are done before `cur` is overwritten
    landed();
  }

  // ---- this wave's totals: d centers / d log_sigs summed over the 4 lane groups (different queries, same centre)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
    for (int j = 0; j < DC; ++j) {
      float v = gc[ct][j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      gc[ct][j] = v * s2m2[ct];
    }
    float v = gls[ct];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    gls[ct] = v * s2m2[ct];
  }
  // ---- 4 waves (query slices) summed in fixed order through LDS, slab row written (format of rbf_vjp_kernel)
  __syncthreads();
  const int V = DC + 1 + a.OP;
  constexpr int COLS = 16 * CT;
  float* red = reinterpret_cast<float*>(lds);                 // [4][V][COLS + 1]
  const float wscale = sg * (1.0f / (16384.0f * kWScale));    // s_g * 2^-14 (phi) * 2^-15 (cotangent)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = ct * 16 + n;
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < DC; ++j) red[(wave * V + j) * (COLS + 1) + col] = gc[ct][j];
      red[(wave * V + DC) * (COLS + 1) + col] = gls[ct];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 4 * g + r;
      if (o < a.OP) red[(wave * V + DC + 1 + o) * (COLS + 1) + col] = __builtin_fmaf(dWl[ct][r], kLoScale, dW[ct][r]) * wscale;
    }
  }
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.y * V * a.Npad + cb;
  for (int idx = tid; idx < V * COLS; idx += 256) {
    const int v = idx / COLS, col = idx - v * COLS;
    if (cb + col < a.Npad) {
      const float s = (red[(0 * V + v) * (COLS + 1) + col] + red[(1 * V + v) * (COLS + 1) + col]) +
                      (red[(2 * V + v) * (COLS + 1) + col] + red[(3 * V + v) * (COLS + 1) + col]);
      dst[(size_t)v * a.Npad + col] = s;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
bool vjph_eligible(const irbfn_net* net) {
  return net->f16_oscale != nullptr && net->R == 1 && net->bclass != BC_GENERIC && net->O <= 16 &&
         (net->DC == 3 || net->DC == 4 || net->DC == 7 || net->DC == 8);
}

size_t vjph_block_bytes(const irbfn_net* net) { return (size_t)32 * vjph_rfq(net->DC) * 4 + 4096; }

template <int DC, int CT>
static int launch_vjph_bc(const VjpHArgs& a, int bc, dim3 grid, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_vjp_f16mfma<DC, BC_GAUSS, CT>), grid, dim3(256), lds, s, a); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_vjp_f16mfma<DC, BC_IQ, CT>), grid, dim3(256), lds, s, a); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_vjp_f16mfma<DC, BC_IMQ, CT>), grid, dim3(256), lds, s, a); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int DC>
static int launch_vjph_dc(const VjpHArgs& a, int CT, int bc, dim3 grid, size_t lds, hipStream_t s) {
  return CT == 4 ? launch_vjph_bc<DC, 4>(a, bc, grid, lds, s) : launch_vjph_bc<DC, 2>(a, bc, grid, lds, s);
}

// x, gout -> slabs part[QSB][V][Npad] (QSB query-slice blocks of 4 waves).  qblk / scales: workspace; bmax: per-block
// max |g| written by colsum_partial_kernel (no separate reduction pass, no atomics).
int launch_vjp_f16(irbfn_net* net, const float* x, const float* gout, int64_t B, unsigned char* qblk, const float* bmax,
                   int nbmax, float* scales, float* part, int QSB, int Npad, int CT, hipStream_t s, const int* run_if, int run_gen) {
  if (!vjph_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  const long nqb = (B + 31) / 32;
  hipLaunchKernelGGL(vjp_pack_blocks_kernel, dim3((unsigned)nqb), dim3(64), 0, s, x, gout, bmax, nbmax, net->f16_oscale,
                     qblk, scales, net->gate(), (long)B, net->D, vjph_rfq(net->DC), net->O, run_if, run_gen);
  IRBFN_HIP_CHECK(hipGetLastError());
  VjpHArgs a;
  a.qblk = qblk; a.scales = scales; a.rec = net->rec; a.sig2 = net->sig2; a.oscale = net->f16_oscale; a.part = part;
  a.nqb = nqb; a.O = net->O; a.OP = net->OP; a.N = net->N; a.S = net->S; a.Npad = Npad; a.basis = net->basis;
  const long slices = (long)QSB * 4;
  a.bpw = (int)((nqb + slices - 1) / slices);
  a.gscale = gauss_scale(net->basis);
  a.run_if = run_if; a.run_gen = run_gen;
  const int groups = (net->N + 16 * CT - 1) / (16 * CT);
  const dim3 grid(groups, QSB);
  const int V = net->DC + 1 + net->OP;
  const size_t ring = (size_t)4 * 2 * vjph_block_bytes(net);
  const size_t red = (size_t)4 * V * (16 * CT + 1) * sizeof(float);
  size_t lds = ring > red ? ring : red;
  lds += (size_t)net->opt[IRBFN_OPT_LDS_PAD];               // diagnosis only: lowers the occupancy
  if (lds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  switch (net->DC) {
    case 3: return launch_vjph_dc<3>(a, CT, net->bclass, grid, lds, s);
    case 4: return launch_vjph_dc<4>(a, CT, net->bclass, grid, lds, s);
    case 7: return launch_vjph_dc<7>(a, CT, net->bclass, grid, lds, s);
    case 8: return launch_vjph_dc<8>(a, CT, net->bclass, grid, lds, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

}  // namespace irbfn
