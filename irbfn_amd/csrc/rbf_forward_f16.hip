// K1h: fused RBF forward for NARROW outputs (O <= 16) with the Phi x W reduction on the f16 matrix cores
// at float32 accuracy.  Same mathematics as K1 (src/irbfn_mpc/model.py:169-198; RBF stage
// flax_rbf.py:258-285): gamma * sum_k phi_k(x) W[k,:] + bias, single region (R == 1).
//
// Why: K1 sits at the VALU issue limit of its instruction mix (26 VALU instructions per (query,
// centre) pair, 10 of them the weight-row FMAs).  An MFMA inside a VALU stream costs its plain 14-20 cycles
// (no co-execution on gfx950) but does 8192 FMAs, so moving Phi x W to v_mfma_f32_16x16x32_f16 removes
// those 10 instructions for 3 (the split below) plus ~6 cycles of MFMA per 64 pairs.
//
// Accuracy: f16 has 11 significant bits, so both operands are split into an (hi, lo) pair, and the lo halves are
// stored PRE-SCALED by 2^11 and accumulated in their own accumulator, so that they stay in the f16 normal range
// wherever the hi half does (no dependence on how a weight compares with its column's maximum):
//     P  = 2^14 * phi            = ph + pl     ph = the top 11 significant bits of P (exact in f16)
//                                              pls = f16_rtz(2^11 (P - ph))                       |pls| < 2^15
//     Ws = 2^15 * W / s_o        = wh + wl     wh = f16_rn(Ws)  (s_o = power of two > max_k |W[k,o]|, |Ws| < 2^15)
//                                              wls = f16_rn(2^11 (Ws - wh))                       |wls| <= 2^14
//     A1 += ph * wh              A2 += pls * wh + ph * wls              (f16 x f16 products are exact in f32)
//     sum_k phi_k W_ko = s_o 2^-29 (A1 + 2^-11 A2)        (the pl*wl term, < 2^-22 relative per pair, is dropped)
// Both halves of a pair are normal f16 numbers down to phi = 2^-28 and |W| = 2^-29 s_o: every product carries
// ~22 significant bits whatever the dynamic range inside a weight column; below those floors the absolute error
// of a factor is 2^-49 (phi) / 2^-50 s_o (W).  The first version of this kernel kept pl and wl unscaled in the
// same accumulator: wl fell into the f16 subnormals for |W| < 2^-3 s_o and the pair lost one bit per factor of two
// below that (3e-5 relative at 2^-10 s_o) -- tests/test_gpu_f16.py::test_forward_f16_ill_conditioned_columns.
// 2^14 is folded into the argument of the transcendental (free).
// TERMS = 1 keeps only ph*wh (plain f16 operands, ~1e-4): the reduced-precision variant BASELINE config 5
// asks to report; reachable only through irbfn_net_set_option(IRBFN_OPT_FWD_F16_TERMS), never by default.
//
// Layout (v_mfma_f32_16x16x32_f16: A[row l&15][k = 8(l>>4)+j], B[k][col l&15], D[row 4(l>>4)+reg][col l&15]):
// rows = 16 queries, k = 32 centres of a chunk, cols = outputs.  A lane owns query (l & 15) of each of its
// wave's two tiles and the 8 centres 8g..8g+7 (g = l >> 4) of every chunk: centre records are broadcast
// reads from a wave-private LDS ring (double buffered, no block barriers in the loop), the W operands come
// straight from HBM/L2 (one coalesced 16-byte load per lane and part).  The transcendentals of a step (16
// per lane) are issued as ONE block (rbf_forward.h).  The MFMAs of step c are issued between the distance
// computations of step c+1.  Centres are sliced across the S waves of a query group; the slices are
// summed in fixed order through LDS (deterministic).
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rbf_forward_f16_narrow.h"

namespace irbfn {

// TERMS: 3 = (hi, lo) f16 pairs (the product path); 1 = plain f16 operands; BF (with TERMS = 1) = plain bf16 operands --
// the two reduced-precision variants BASELINE config 5 asks to report, reachable only through an explicit option.
// ROLL (rbf_tick_f16mfma, the planning tick of the narrow nets: irbfn_planner.py:203-212): the slice-0 wave of a query
// group keeps its 32 rows of controls in LDS and integrates the trajectories itself -- see the epilogue.
template <int DC, int BC, int TERMS, bool BF, bool ROLL>
__device__ __forceinline__ void narrow_body(const F16Args& a, const F16Roll& rl, int mode, unsigned char* lds) {
  static_assert(!BF || TERMS == 1, "bf16 operands: single product");
  constexpr int RF = f16_rf(DC);
  constexpr int RECB = kF16Chunk * RF * 4;                   // record bytes per chunk
  constexpr int CB = f16_chunk_bytes(DC);
  constexpr int NV = RECB / 16;                              // 16-byte pieces of a record block (32 / 64 / 96)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S;
  const int slice = wave % S, qg = wave / S;
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * a.QG + qg) * 32;       // this wave's 32 queries (two 16-row tiles)
  float xq[2][DC];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    q = q < 0 ? 0 : q;
#pragma unroll
    for (int i = 0; i < DC; ++i) xq[t][i] = i < a.Dreal ? a.x[q * a.Dreal + i] : 0.0f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < DC; ++i) asm volatile("" : "+v"(xq[t][i]));   // loads complete before the loop
  // Wave-private double-buffered ring of chunk images [records | W hi | W lo], filled by LDS-DMA
  // (global_load_lds_dwordx4: 1 KiB per wave-instruction, no staging VGPRs, no ds_write pass) one step ahead.
  constexpr int WB = (TERMS >= 3 ? 2 : 1) * kF16WBytes;      // W bytes staged per chunk
  constexpr int BUFB = RECB + 2 * kF16WBytes;
  unsigned char* mylds = lds + wave * (2 * BUFB);
  const int c0 = (int)((long)a.nchunks * slice / S), c1 = (int)((long)a.nchunks * (slice + 1) / S);
  auto stage = [&](int c, unsigned char* dst) {              // chunk c -> dst (asynchronous; retired by vmcnt)
    const unsigned char* gp = a.img + (size_t)c * CB + lane * 16;
#pragma unroll
    for (int v = 0; v < (NV + 63) / 64; ++v)
      if (v * 64 + lane < NV)
        __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024),
                                         (lptr_t)(dst + v * 1024), 16, 0, 0);
#pragma unroll
    for (int v = 0; v < WB / 1024; ++v)          // bf16 variant: the third W part lands in the "hi" slot of the ring
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + RECB + (BF ? 2 * kF16WBytes : 0) + v * 1024),
                                       (lptr_t)(dst + RECB + v * 1024), 16, 0, 0);
  };

  f4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A1: ph * wh
  f4_t acl[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A2: pls * wh + ph * wls  (2^11 x the lo terms)
  h8_t ah[2], al[2], bh, bl;                                 // operands of the PREVIOUS step (deferred MFMAs)
#pragma unroll
  for (int j = 0; j < 8; ++j) { ah[0][j] = 0; ah[1][j] = 0; al[0][j] = 0; al[1][j] = 0; bh[j] = 0; bl[j] = 0; }
  if (c0 < c1) stage(c0, mylds);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int c = c0; c < c1; ++c) {
    const unsigned char* cur = mylds + ((c - c0) & 1) * BUFB;
    unsigned char* nxt = mylds + ((c - c0 + 1) & 1) * BUFB;
    // every LDS read of the previous step (records and W of `nxt`) has returned before the DMA overwrites it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c + 1 < c1) stage(c + 1, nxt);
    float t16[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* rp = reinterpret_cast<const float*>(cur) + (8 * g + j) * RF;
      float r[RF];
#pragma unroll
      for (int v = 0; v < RF / 4; ++v) {
        const f4_t rr = *reinterpret_cast<const f4_t*>(rp + 4 * v);
        r[4 * v] = rr.x; r[4 * v + 1] = rr.y; r[4 * v + 2] = rr.z; r[4 * v + 3] = rr.w;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float r2 = 0.0f;
#pragma unroll
        for (int i = 0; i < DC; ++i) {
          const float d = xq[t][i] - r[i];                   // flax_rbf.py:280  (x - centers)
          r2 = __builtin_fmaf(d, d, r2);
        }
        t16[t * 8 + j] = f16_arg<BC>(r2, r[RF - 1]);
      }
      // (the two tiles' distances as v_pk_add_f32 / v_pk_fma_f32 with the centre broadcast by op_sel -- 7 packed instead of
      // 14 plain instructions per pair of pairs: measured 135.5 vs 131.7-132.3 us at config 2, rejected)
      // one deferred MFMA per centre: (tile, term) = (j & 1, j >> 1); terms: ph*wh -> A1; pls*wh, ph*wls -> A2
      const int t = j & 1, m = j >> 1;
      if (m < TERMS) {
        const h8_t av = (m == 1) ? al[t] : ah[t];
        const h8_t bv = (m == 2) ? bl : bh;
        if (m == 0 || (m == 1 && !IRBFN_K1H_PHI_LOS)) acc[t] = mfma_16x16x32<BF>(av, bv, acc[t]);
        else acl[t] = mfma_16x16x32<BF>(av, bv, acl[t]);
      }
    }
    trans_block<BC, 16>(t16);                                // P = 2^kPhiExp * phi for the step's 16 pairs
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned wh[4], wl[4];                                 // 4 packed f16 pairs each = one MFMA A operand
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        if constexpr (BF) {
          wh[jj] = __builtin_bit_cast(unsigned, bf2_t{(__bf16)t16[t * 8 + 2 * jj], (__bf16)t16[t * 8 + 2 * jj + 1]});
          wl[jj] = 0u;
        } else {
          split_pair_f16<TERMS, IRBFN_K1H_PHI_LOS>(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
        }
      }
      ah[t] = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
      al[t] = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
    }
    bh = *reinterpret_cast<const h8_t*>(cur + RECB + lane * 16);              // this chunk's W operands (B layout)
    if constexpr (TERMS >= 3) bl = *reinterpret_cast<const h8_t*>(cur + RECB + kF16WBytes + lane * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // chunk c + 1 has landed in `nxt`
  }
  // drain the deferred MFMAs of the last step
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc[t] = mfma_16x16x32<BF>(ah[t], bh, acc[t]);
    if constexpr (TERMS >= 2 && IRBFN_K1H_PHI_LOS) acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acl[t], 0, 0, 0);
    if constexpr (TERMS >= 2 && !IRBFN_K1H_PHI_LOS) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t], 0, 0, 0);
    if constexpr (TERMS >= 3) acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acl[t], 0, 0, 0);
    if constexpr (TERMS >= 2)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], kLoScale, acc[t][r]);   // A1 + 2^-11 A2
  }

  // ---- smooth region gate of the single region (model.py:42-95), one value per query
  const GateTables gt = a.gate;
  float gam[2] = {0.0f, 0.0f};
  if (slice == 0) {                                          // only the slice-0 wave of a query group applies it
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float gv = gt.n_ranges > 0 ? 1.0f : 0.0f;              // model.py:70
#pragma unroll
      for (int d = 0; d < DC; ++d)
        if (d < gt.nsplit && gt.n_ranges > 0) {
          const int e = d * gt.max_ranges + gt.dim_ranges[d];
          gv *= gate_factor(xq[t][d], gt.lo[e], gt.hi[e], gt.delta[d]);
        }
      gam[t] = gv;
    }
  }
  narrow_epilogue<ROLL>(a, rl, mode, lds, acc, gam, S, slice, qg, q0, 1.0f / (16384.0f * kWScale));
}

template <int DC, int BC, int TERMS, bool BF = false>
__global__ __launch_bounds__(512, IRBFN_K1H_MIN_WAVES) void rbf_fwd_f16mfma(const F16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  narrow_body<DC, BC, TERMS, BF, false>(a, F16Roll{}, -1, lds);
}

// the planning tick of a narrow net in one launch (forward + sign flip + roll-out)
template <int DC, int BC>
__global__ __launch_bounds__(512, IRBFN_K1H_MIN_WAVES) void rbf_tick_f16mfma(const F16Args a, const F16Roll rl, const int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  narrow_body<DC, BC, 3, false, true>(a, rl, mode, lds);
}

// ---- wide outputs (16 < O <= 128): NT = ceil(O/16) column tiles ----------------------------------------------
// Same mathematics and operand split; the W operands of a chunk (NT x 2 KiB) are too many to stream per wave,
// so the 8 waves of a block are SW centre slices x QG query groups and the QG waves of a slice SHARE one
// double-buffered LDS image of the slice's current chunk (cooperative 16-byte copies, one barrier per step).
// Per step a wave computes the distances / basis / split of its 32 queries x 32 centres, then for every
// column tile reads (bh, bl) from LDS and issues 2 row tiles x 3 terms MFMAs.
template <int DC, int BC, int NT>
__global__ __launch_bounds__(512) void rbf_fwd_f16mfma_wide(const F16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int RF = f16_rf(DC);
  constexpr int RECB = kF16Chunk * RF * 4;
  constexpr int CB = f16_chunk_bytes(DC, NT);
  constexpr int NV = CB / 16;                                // 16-byte pieces per chunk image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SW = a.S, QG = a.QG;
  const int slice = wave / QG, qg = wave % QG;               // the QG waves of a slice are adjacent
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * 32;
  float xq[2][DC];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    q = q < 0 ? 0 : q;
#pragma unroll
    for (int i = 0; i < DC; ++i) xq[t][i] = i < a.Dreal ? a.x[q * a.Dreal + i] : 0.0f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < DC; ++i) asm volatile("" : "+v"(xq[t][i]));
  const int nsteps = (a.nchunks + SW - 1) / SW;              // chunks per slice (the last slice may have fewer)
  const int c0 = slice * nsteps;
  const int c1 = (c0 + nsteps) < a.nchunks ? (c0 + nsteps) : a.nchunks;
  unsigned char* stream = lds + (size_t)slice * 2 * CB;      // this slice's two chunk buffers
  const int st = qg * 64 + lane;                             // thread index within the slice's copy team
  const int team = QG * 64;
  constexpr int MAXP = (NV + 127) / 128;                     // pieces per thread at the smallest team (QG = 2)
  u4_t pre[MAXP];
  auto fetch = [&](int c) {
    const u4_t* src = reinterpret_cast<const u4_t*>(a.img + (size_t)c * CB);
#pragma unroll
    for (int v = 0; v < MAXP; ++v)
      if (v * team + st < NV) pre[v] = src[v * team + st];
  };
  auto stash = [&](unsigned char* dst) {
#pragma unroll
    for (int v = 0; v < MAXP; ++v)
      if (v * team + st < NV) reinterpret_cast<u4_t*>(dst)[v * team + st] = pre[v];
  };
  f4_t acc[2][NT], acl[2][NT];                               // A1, A2 (header)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) { acc[t][ct] = f4_t{0, 0, 0, 0}; acl[t][ct] = f4_t{0, 0, 0, 0}; }
  if (c0 < c1) { fetch(c0); stash(stream); }
  __syncthreads();
  for (int i = 0; i < nsteps; ++i) {
    const int c = c0 + i;
    const unsigned char* cur = stream + (i & 1) * CB;
    unsigned char* nxt = stream + ((i + 1) & 1) * CB;
    const bool has_next = c + 1 < c1;
    if (has_next) fetch(c + 1);
    if (c < c1) {
      float t16[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* rp = reinterpret_cast<const float*>(cur) + (8 * g + j) * RF;
        float r[RF];
#pragma unroll
        for (int v = 0; v < RF / 4; ++v) {
          const f4_t rr = *reinterpret_cast<const f4_t*>(rp + 4 * v);
          r[4 * v] = rr.x; r[4 * v + 1] = rr.y; r[4 * v + 2] = rr.z; r[4 * v + 3] = rr.w;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float r2 = 0.0f;
#pragma unroll
          for (int d = 0; d < DC; ++d) {
            const float df = xq[t][d] - r[d];
            r2 = __builtin_fmaf(df, df, r2);
          }
          t16[t * 8 + j] = f16_arg<BC>(r2, r[RF - 1]);
        }
      }
      trans_block<BC, 16>(t16);
      h8_t ah[2], al[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned wh[4], wl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          split_pair_f16<3, IRBFN_K1H_PHI_LOS>(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
        ah[t] = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
        al[t] = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
      }
      // W operands of tile ct + 1 are read while the 6 MFMAs of tile ct run (the LDS latency is otherwise
      // exposed 7 times per step: hipcc issues each read right in front of its first use)
      h8_t bh = *reinterpret_cast<const h8_t*>(cur + RECB + lane * 16);
      h8_t bl = *reinterpret_cast<const h8_t*>(cur + RECB + kF16WBytes + lane * 16);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        h8_t nbh = bh, nbl = bl;
        if (ct + 1 < NT) {
          nbh = *reinterpret_cast<const h8_t*>(cur + RECB + (ct + 1) * 2 * kF16WBytes + lane * 16);
          nbl = *reinterpret_cast<const h8_t*>(cur + RECB + (ct + 1) * 2 * kF16WBytes + kF16WBytes + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][ct], 0, 0, 0);
          if constexpr (IRBFN_K1H_PHI_LOS) acl[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acl[t][ct], 0, 0, 0);
          else acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t][ct], 0, 0, 0);
          acl[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acl[t][ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        bh = nbh; bl = nbl;
      }
    }
    if (has_next) stash(nxt);
    __syncthreads();
  }

  const GateTables gt = a.gate;
  float gam[2] = {0.0f, 0.0f};
  if (slice == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float gv = gt.n_ranges > 0 ? 1.0f : 0.0f;
#pragma unroll
      for (int d = 0; d < DC; ++d)
        if (d < gt.nsplit && gt.n_ranges > 0) {
          const int e = d * gt.max_ranges + gt.dim_ranges[d];
          gv *= gate_factor(xq[t][d], gt.lo[e], gt.hi[e], gt.delta[d]);
        }
      gam[t] = gv;
    }
  }
  // slices summed per column tile in fixed order through LDS (the streams are dead: the loop ended on a barrier)
  float* red = reinterpret_cast<float*>(lds);                // [SW][QG][2][4][64]
  float* gl = red + (size_t)SW * QG * 2 * 4 * 64;            // [QG][32]
  if (slice == 0 && g == 0) { gl[qg * 32 + n] = gam[0]; gl[qg * 32 + 16 + n] = gam[1]; }
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[(((slice * QG + qg) * 2 + t) * 4 + r) * 64 + lane] = __builtin_fmaf(acl[t][ct][r], kLoScale, acc[t][ct][r]);
    __syncthreads();
    const int o = ct * 16 + n;
    if (slice == 0 && o < a.O) {
      const float sc = a.oscale[o] * (1.0f / (16384.0f * kWScale));
      const float bi = a.bias[o];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 0.0f;
          for (int s2 = 0; s2 < SW; ++s2) v += red[(((s2 * QG + qg) * 2 + t) * 4 + r) * 64 + lane];
          const int row = t * 16 + 4 * g + r;
          const long q = q0 + row;
          if (q < a.B) a.out[q * a.O + o] = __builtin_fmaf(gl[qg * 32 + row] * v, sc, bi);
        }
    }
    __syncthreads();
  }
}


// ---- wide outputs, pipelined (rbf_forward_f16_wide.h): the default --------------------------------------------------
template <int DC, int BC, int NT>
__global__ __launch_bounds__(512) void rbf_fwd_f16mfma_wide_pipe(const F16Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  wide_pipe_body<DC, BC, NT, -1>(a, F16Roll{}, lds);
}

// ---- host side -------------------------------------------------------------------------------------------
bool f16_eligible(const irbfn_net* net) {
  return net->R == 1 && net->bclass != BC_GENERIC && net->O <= 128 && (net->DC == 3 || net->DC == 4 || net->DC == 7 || net->DC == 8);
}

static int f16_nt(const irbfn_net* net) { return (net->O + 15) / 16; }

size_t f16_image_bytes(const irbfn_net* net) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  return (size_t)nchunks * f16_chunk_bytes(net->DC, f16_nt(net));
}

template <int DC, int NT>
static int launch_f16w_bc(const F16Args& a, int bc, int grid, int block, size_t lds, bool pipe, hipStream_t s) {
#define IRBFN_WCASE(BCV)                                                                                      \
  case BCV: {                                                                                                 \
    auto k = pipe ? rbf_fwd_f16mfma_wide_pipe<DC, BCV, NT> : rbf_fwd_f16mfma_wide<DC, BCV, NT>;               \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a);                                                \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_WCASE(BC_GAUSS)
    IRBFN_WCASE(BC_IQ)
    IRBFN_WCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_WCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int DC>
static int launch_f16w_dc(const F16Args& a, int NT, int bc, int grid, int block, size_t lds, bool pipe, hipStream_t s) {
  switch (NT) {
    case 2: return launch_f16w_bc<DC, 2>(a, bc, grid, block, lds, pipe, s);
    case 3: return launch_f16w_bc<DC, 3>(a, bc, grid, block, lds, pipe, s);
    case 4: return launch_f16w_bc<DC, 4>(a, bc, grid, block, lds, pipe, s);
    case 5: return launch_f16w_bc<DC, 5>(a, bc, grid, block, lds, pipe, s);
    case 6: return launch_f16w_bc<DC, 6>(a, bc, grid, block, lds, pipe, s);
    case 7: return launch_f16w_bc<DC, 7>(a, bc, grid, block, lds, pipe, s);
    case 8: return launch_f16w_bc<DC, 8>(a, bc, grid, block, lds, pipe, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

// final block geometry of the wide kernels and the choice between them: the pipelined kernel (kWideRing chunk buffers
// per slice) wherever its ring fits the 160 KB of LDS; IRBFN_OPT_FWD_WIDE_PIPE = 0 selects the two-buffer kernel
void f16_wide_normalize(const irbfn_net* net, int* SW, int* QG, bool* pipe) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  while (*SW > 1 && nchunks / *SW < 2) *SW /= 2;
  if (*QG < 2 || *SW * *QG > 8) *QG = 8 / *SW;               // the copy team needs >= 128 threads per slice
  *pipe = net->opt[IRBFN_OPT_FWD_WIDE_PIPE] != 0 &&
          (size_t)*SW * kWideRing * f16_chunk_bytes(net->DC, f16_nt(net)) <= 160 * 1024;
}

// SW centre slices (1, 2 or 4) x QG = 8 / SW query groups of 32 per 512-thread block
static int launch_forward_f16_wide(irbfn_net* net, const float* x, float* out, int64_t B, int SW, int QG, hipStream_t s) {
  const int NT = f16_nt(net);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  if (SW != 1 && SW != 2 && SW != 4) return IRBFN_ERR_BAD_ARG;
  bool pipe;
  f16_wide_normalize(net, &SW, &QG, &pipe);
  F16Args a;
  a.x = x; a.img = net->f16_img; a.oscale = net->f16_oscale; a.bias = net->bias; a.out = out; a.gate = net->gate();
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.nchunks = nchunks; a.S = SW; a.QG = QG;
  const size_t stream = (size_t)SW * (pipe ? kWideRing : 2) * f16_chunk_bytes(net->DC, NT);
  const size_t red = ((size_t)SW * QG * 2 * 4 * 64 + (size_t)QG * 32) * sizeof(float);
  const size_t lds = stream > red ? stream : red;
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_f16w_dc<3>(a, NT, net->bclass, grid, SW * QG * 64, lds, pipe, s); break;
    case 4: rc = launch_f16w_dc<4>(a, NT, net->bclass, grid, SW * QG * 64, lds, pipe, s); break;
    case 7: rc = launch_f16w_dc<7>(a, NT, net->bclass, grid, SW * QG * 64, lds, pipe, s); break;
    case 8: rc = launch_f16w_dc<8>(a, NT, net->bclass, grid, SW * QG * 64, lds, pipe, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_f16mfma_wide%s<D=%d,BC=%d,NT=%d,SW=%d,QG=%d>", pipe ? "_pipe" : "",
             net->DC, net->bclass, NT, SW, QG);
    net->last_grid = grid;
    net->last_block = SW * QG * 64;
  }
  return rc;
}

template <int DC, int TERMS, bool BF = false>
static int launch_f16_bc(const F16Args& a, int bc, int grid, int block, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_fwd_f16mfma<DC, BC_GAUSS, TERMS, BF>), dim3(grid), dim3(block), lds, s, a); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_fwd_f16mfma<DC, BC_IQ, TERMS, BF>), dim3(grid), dim3(block), lds, s, a); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_fwd_f16mfma<DC, BC_IMQ, TERMS, BF>), dim3(grid), dim3(block), lds, s, a); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int DC>
static int launch_f16_dc(const F16Args& a, int terms, int bc, int grid, int block, size_t lds, hipStream_t s) {
  if (terms == 2) return launch_f16_bc<DC, 1, true>(a, bc, grid, block, lds, s);       // plain bf16 operands
  return terms == 1 ? launch_f16_bc<DC, 1>(a, bc, grid, block, lds, s) : launch_f16_bc<DC, 3>(a, bc, grid, block, lds, s);
}

template <int DC>
static int launch_tick_narrow_bc(const F16Args& a, const F16Roll& rl, int mode, int bc, int grid, int block, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_tick_f16mfma<DC, BC_GAUSS>), dim3(grid), dim3(block), lds, s, a, rl, mode); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_tick_f16mfma<DC, BC_IQ>), dim3(grid), dim3(block), lds, s, a, rl, mode); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_tick_f16mfma<DC, BC_IMQ>), dim3(grid), dim3(block), lds, s, a, rl, mode); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// does the one-launch tick of a narrow net exist (and is it enabled) for this net / mode / batch / horizon?
static bool tick_narrow_plan(const irbfn_net* net, int mode, int64_t B, int T, int* S_out, int* QG_out, size_t* lds_out) {
  if (net->opt[IRBFN_OPT_TICK_FUSED] == 0 || net->O > 16 || net->O != 2 * T || T > kTickNarrowT) return false;
  const bool st = mode == IRBFN_ROLLOUT_ST_SELECT || mode == IRBFN_ROLLOUT_ST_KS || mode == IRBFN_ROLLOUT_FULLINT;
  if (!((st && net->DC == 7) || (mode == IRBFN_ROLLOUT_FRENET_LS && net->DC == 8))) return false;
  if (net->opt[IRBFN_OPT_FWD_F16_TERMS] != 3 && net->opt[IRBFN_OPT_FWD_F16_TERMS] != 0) return false;
  int S, QG;
  if (!f16_narrow_geometry(net, B, &S, &QG)) return false;   // the forward would not run K1h
  const int waves = S * QG;
  const size_t ring = (size_t)waves * 2 * (kF16Chunk * f16_rf(net->DC) * 4 + 2 * kF16WBytes);
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32 + (size_t)QG * 32 * (kTickNarrowCP + kTickNarrowSP)) * sizeof(float);
  const size_t lds = ring > red ? ring : red;
  if (lds > 64 * 1024) return false;
  *S_out = S; *QG_out = QG; *lds_out = lds;
  return true;
}

bool tick_f16_narrow_available(const irbfn_net* net, int mode, int64_t B, int T) {
  int S, QG;
  size_t lds;
  return tick_narrow_plan(net, mode, B, T, &S, &QG, &lds);
}

// IRBFN_ERR_UNSUPPORTED: no instance -> the caller takes another path
int launch_tick_f16_narrow(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                           const DynParams& dp, float* controls, float* states, int64_t B, int T, hipStream_t s) {
  int S, QG;
  size_t lds;
  if (!tick_narrow_plan(net, mode, B, T, &S, &QG, &lds)) return IRBFN_ERR_UNSUPPORTED;
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  F16Args a;
  a.x = x; a.img = net->f16_img; a.oscale = net->f16_oscale; a.bias = net->bias; a.out = controls; a.gate = net->gate();
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.nchunks = nchunks; a.S = S; a.QG = QG;
  F16Roll rl;
  rl.state0 = state0; rl.states = states; rl.mirror = mirror; rl.T = T; rl.wlds = 0; rl.dp = dp;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  const int rc = net->DC == 7 ? launch_tick_narrow_bc<7>(a, rl, mode, net->bclass, grid, S * QG * 64, lds, s)
                              : launch_tick_narrow_bc<8>(a, rl, mode, net->bclass, grid, S * QG * 64, lds, s);
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_tick_f16mfma<D=%d,BC=%d,MODE=%d,S=%d,QG=%d>", net->DC, net->bclass, mode, S, QG);
    net->last_grid = grid;
    net->last_block = S * QG * 64;
  }
  return rc;
}

// S = centre slices per query group, QG = query groups (of 32) per block; S * QG <= 8 waves
int launch_forward_f16(irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, int terms, hipStream_t s) {
  if (!net->f16_img || !f16_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  if (net->O > 16) return launch_forward_f16_wide(net, x, out, B, S, QG, s);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  if (S < 1 || QG < 1 || S * QG > 8 || S > nchunks) return IRBFN_ERR_BAD_ARG;
  F16Args a;
  a.x = x; a.img = net->f16_img; a.oscale = net->f16_oscale; a.bias = net->bias; a.out = out; a.gate = net->gate();
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.nchunks = nchunks; a.S = S; a.QG = QG;
  const int waves = S * QG;
  const size_t ring = (size_t)waves * 2 * (kF16Chunk * f16_rf(net->DC) * 4 + 2 * kF16WBytes);
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32) * sizeof(float);
  size_t lds = ring > red ? ring : red;
  lds += (size_t)net->opt[IRBFN_OPT_LDS_PAD];               // diagnosis only: lowers the occupancy
  if (lds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_f16_dc<3>(a, terms, net->bclass, grid, waves * 64, lds, s); break;
    case 4: rc = launch_f16_dc<4>(a, terms, net->bclass, grid, waves * 64, lds, s); break;
    case 7: rc = launch_f16_dc<7>(a, terms, net->bclass, grid, waves * 64, lds, s); break;
    case 8: rc = launch_f16_dc<8>(a, terms, net->bclass, grid, waves * 64, lds, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_f16mfma<D=%d,BC=%d,TERMS=%s,S=%d,QG=%d>", net->DC,
             net->bclass, terms == 1 ? "1" : (terms == 2 ? "1,BF16" : "3"), S, QG);
    net->last_grid = grid;
    net->last_block = waves * 64;
  }
  return rc;
}

}  // namespace irbfn
