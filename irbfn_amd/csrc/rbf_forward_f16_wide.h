// Shared definitions of the f16-MFMA forward kernels (K1h) and the pipelined wide kernel's body, which is instantiated
// both as the plain forward (rbf_forward_f16.hip) and as the fused planning tick (plan_tick_wide.hip).
#pragma once

#include <hip/hip_fp16.h>
#include <stdint.h>

#include "f16_split.h"
#include "rbf_forward.h"
#include "rollout_pair.h"

namespace irbfn {

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;   // operands of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kF16Chunk = 32;                    // centres per chunk = MFMA K
constexpr int kF16WBytes = 4 * 16 * 8 * 2;       // one W part of a chunk: [g][n][8] halfs = 1 KiB
constexpr int f16_rf(int DC) { return DC <= 3 ? 4 : (DC <= 7 ? 8 : 12); }    // floats per centre record
constexpr int f16_chunk_bytes(int DC, int NT = 1) { return kF16Chunk * f16_rf(DC) * 4 + NT * 2 * kF16WBytes + (NT == 1 ? kF16WBytes : 0); }
// chunk image: rec[32][RF] floats, then per column tile ct < NT: Whi[ct] (1 KiB), Wlo[ct] (1 KiB); narrow nets (NT = 1)
// carry a third part Wbf (1 KiB): the same scaled weights rounded to bf16, for the plain-bf16 variant of config 5
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));

struct F16Args {
  const float* __restrict__ x;            // [B][Dreal]
  const unsigned char* __restrict__ img;  // [nchunks][chunk image]
  const float* __restrict__ oscale;       // [16 * NT] s_o
  const float* __restrict__ bias;         // [OP]
  float* __restrict__ out;                // [B][O]
  GateTables gate;
  long B;
  int Dreal, O, nchunks, S, QG;
};

// ---- kernel ----------------------------------------------------------------------------------------------
template <int BC>
__device__ __forceinline__ float f16_arg(float r2, float sc) {
  if constexpr (BC == BC_GAUSS) return __builtin_fmaf(r2, sc, (float)kPhiExp);
  else if constexpr (BC == BC_IQ) return __builtin_fmaf(r2, sc, kPhiInv);
  else return __builtin_fmaf(r2, sc, kPhiInv * kPhiInv);
}

#ifndef IRBFN_K1H_PHI_LOS
#define IRBFN_K1H_PHI_LOS 1        // 1: residual of the basis value pre-scaled by 2^11 (A2); 0: unscaled (A1)
#endif
#ifndef IRBFN_K1H_MIN_WAVES
#define IRBFN_K1H_MIN_WAVES 2      // waves per SIMD the register allocation must allow (512 threads = 2 per block)
#endif
template <bool BF>
__device__ __forceinline__ f4_t mfma_16x16x32(h8_t av, h8_t bv, f4_t c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8_t, av), __builtin_bit_cast(bf8_t, bv), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
}

// ---- wide outputs, pipelined: the default ------------------------------------------------------------------------
// The kernel above issues a step's 6 NT MFMAs as one burst behind the step's VALU work, stages the chunk images through
// VGPRs (32 of them) and a ds_write pass, and keeps every wave of the block in the same phase.  Here
//  * the MFMAs of step c are DEFERRED into step c + 1: column tile j's six are issued behind the distances of centre j;
//  * the chunk images are staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction: no staging VGPRs, no
//    ds_write pass) into a ring of kWideRing buffers per slice: W of chunk c - 1 (the deferred MFMAs), records of chunk
//    c, chunk c + 1 landing.  One barrier per step, at its top.
// Measured at BASELINE config 4 (O = 100; tools/sweep_wide_geometry.py, same results bit for bit): B = 32768 per GPU
// 173 -> 135-141 us, B = 262144 1050-1150 -> 910-980 us.  What did NOT help (measured, tools/build_variant.py): a ring
// of four with chunks requested two steps ahead (+4 %: the top-of-step wait was never the DMA), spreading the MFMAs
// evenly over the step with sched_group_barrier (+6 %).  Ablation at B = 262144: distances / basis / split only 495 us,
// MFMAs + W reads only 538 us, both 1165 us -- the sum: the chip is clock-limited under this load (effective clock
// GRBM_GUI_ACTIVE / 8 / wall = 2.09 GHz here, 1.91 GHz for the narrow kernel at config 2; MI355X_MICROARCH.md "DVFS
// give-back"), so issue-level overlap of the two pipes returns as lower clock, and what counts is the work per pair.
template <bool B>
struct BoolC { static constexpr bool value = B; };
#ifndef IRBFN_WIDE_RING
#define IRBFN_WIDE_RING 3
#endif
#ifndef IRBFN_WIDE_AHEAD
#define IRBFN_WIDE_AHEAD (IRBFN_WIDE_RING - 2)
#endif
constexpr int kWideRing = IRBFN_WIDE_RING;     // chunk buffers per slice: W of chunk c - 1, chunk c, kWideAhead chunks landing
constexpr int kWideAhead = IRBFN_WIDE_AHEAD;   // steps a chunk is requested ahead of its use
static_assert(kWideAhead >= 1 && kWideAhead <= kWideRing - 2, "slot of chunk i + ahead must not be chunk i - 1's or chunk i's");

constexpr int kTickTch = 50;         // control knots per lane the fused tick is compiled for (T <= 50)
struct F16Roll {                        // the roll-out behind the forward (fused planning tick, MODE >= 0)
  const float* __restrict__ state0;     // [B][S0]
  float* __restrict__ states;           // [B][T][S]
  const int* __restrict__ mirror;       // [B] or null: rows with mirror != 0 get outputs [T, 2T) negated (irbfn_planner.py:203-204)
  int T;
  int wlds;                             // floats of LDS per roll-out wave
  DynParams dp;
};

// The epilogue of the pipelined wide kernels (K1h rbf_fwd_f16mfma_wide_pipe / rbf_tick_f16mfma_wide, K1g's wide form): gate of the
// single region (model.py:42-95), the SW centre slices summed per column tile in fixed order through LDS, scale, bias, store;
// MODE >= 0: the block's controls stay in LDS and its slice-0 waves roll the trajectories out (rollout_pair.h).  acc / acl: A1 / A2
// of f16_split.h per query tile and column tile; inv_scale: 1 / (scale of the basis values x 2^15).  Every wave of the block
// arrives here; the ring is dead.
template <int DC, int NT, int MODE>
__device__ __forceinline__ void wide_epilogue(const F16Args& a, const F16Roll& rl, unsigned char* lds, const f4_t (&acc)[2][NT],
                                              const f4_t (&acl)[2][NT], int slice, int qg, long q0, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n = lane & 15;
  const int SW = a.S, QG = a.QG;
  long qrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    qrow[t] = q < 0 ? 0 : q;
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
          gv *= gate_factor(a.x[qrow[t] * a.Dreal + d], gt.lo[e], gt.hi[e], gt.delta[d]);
        }
      gam[t] = gv;
    }
  }
  __syncthreads();                                           // every wave is done with the ring
  float* red = reinterpret_cast<float*>(lds);                // [SW][QG][2][4][64]
  float* gl = red + (size_t)SW * QG * 2 * 4 * 64;            // [QG][32]
  [[maybe_unused]] float* ctile = gl + QG * 32;              // MODE >= 0: the block's controls [QG * 32][CP]
  [[maybe_unused]] const int CP = a.O | 1;                   // odd pitch: a lane's row reads spread over the banks
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
      const float sc = a.oscale[o] * inv_scale;
      const float bi = a.bias[o];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = 0.0f;
          for (int s2 = 0; s2 < SW; ++s2) v += red[(((s2 * QG + qg) * 2 + t) * 4 + r) * 64 + lane];
          const int row = t * 16 + 4 * g + r;
          const long q = q0 + row;
          float y = __builtin_fmaf(gl[qg * 32 + row] * v, sc, bi);
          if constexpr (MODE >= 0) {
            // planner mirror trick (irbfn_planner.py:203-204): steering-rate knots of mirrored rows change sign
            if (rl.mirror != nullptr && o >= rl.T && q < a.B && rl.mirror[q] != 0) y = -y;
            ctile[(qg * 32 + row) * CP + o] = y;             // rows past the batch hold the clamped last query: finite, unused
          }
          if (q < a.B && a.out != nullptr) a.out[q * a.O + o] = y;
        }
    }
    __syncthreads();
  }

  if constexpr (MODE >= 0) {
    // ---- the roll-out of the block's QG x 32 trajectories (K3p core, rollout_pair.h) by its slice-0 waves: the controls
    // never leave the CU between the two stages.  Each lane pulls ONE control stream of its row into registers; after
    // the barrier the tile is dead and the LDS becomes the waves' output staging tiles.
    static_assert(MODE == IRBFN_ROLLOUT_ST_SELECT || MODE == IRBFN_ROLLOUT_ST_KS, "instantiated for the single-track tick");
    constexpr int S = ModeTraits<MODE>::S;
    constexpr int TS = pair_ts(S);
    const int T = rl.T;
    const int odd = lane & 1, prow = lane >> 1;
    const long left = a.B - q0;
    const int nvalid = left < kPairRows ? (left > 0 ? (int)left : 0) : kPairRows;
    const bool roller = slice == 0 && nvalid > 0;
    float st[S], ctl[kTickTch];
#pragma unroll
    for (int t = 0; t < kTickTch; ++t) ctl[t] = 0.0f;
    if (roller) {
      const long bb = q0 + (prow < nvalid ? prow : nvalid - 1);
#pragma unroll
      for (int i = 0; i < S; ++i) st[i] = rl.state0[bb * S + i];
      const float* ur = ctile + (qg * 32 + prow) * CP + (odd ? T : 0);
#pragma unroll
      for (int t = 0; t < kTickTch; ++t)
        if (t < T) ctl[t] = ur[t];
    }
    __syncthreads();                                         // the controls tile is dead
    if (!roller) return;
    float* tile = reinterpret_cast<float*>(lds) + (size_t)qg * rl.wlds;
    pair_rollout_run<MODE, kTickTch, TS>(st, ctl, rl.dp, tile, rl.states + q0 * (long)T * S, T, nvalid, lane);
  }
}

// MODE < 0: forward only.  MODE >= 0 (plan_tick_wide.hip): the block's controls stay in LDS and its slice-0 waves roll
// the trajectories out (rollout_pair.h) -- the planning tick in ONE launch.
template <int DC, int BC, int NT, int MODE>
__device__ __forceinline__ void wide_pipe_body(const F16Args& a, const F16Roll& rl, unsigned char* lds) {
  constexpr int RF = f16_rf(DC);
  constexpr int RECB = kF16Chunk * RF * 4;
  constexpr int CB = f16_chunk_bytes(DC, NT);
  constexpr int NV = CB / 16;                                // 16-byte pieces per chunk image
  constexpr int NVI = (NV + 63) / 64;                        // wave-instructions per chunk image
  static_assert(NT <= 8, "column tile j rides behind centre j of the next step");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SW = a.S, QG = a.QG;
  const int slice = wave / QG, qg = wave % QG;
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
  const int na = c1 > c0 ? c1 - c0 : 0;                      // this slice's chunks
  unsigned char* ring = lds + (size_t)slice * kWideRing * CB;
  auto stage = [&](int c, unsigned char* dst) {              // the slice's QG waves share the copy: wave qg takes every QG-th KiB
    const unsigned char* gp = a.img + (size_t)c * CB + lane * 16;
    for (int v = qg; v < NVI; v += QG)
      if (v * 64 + lane < NV)
        __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  f4_t acc[2][NT], acl[2][NT];                               // A1, A2 (header)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) { acc[t][ct] = f4_t{0, 0, 0, 0}; acl[t][ct] = f4_t{0, 0, 0, 0}; }
  h8_t ah[2], al[2];                                         // A operands of the previous step (deferred MFMAs)
#pragma unroll
  for (int j = 0; j < 8; ++j) { ah[0][j] = 0; ah[1][j] = 0; al[0][j] = 0; al[1][j] = 0; }

  // one step: VALU = distances / basis / split of chunk `cur`; MFMA = the deferred products of the previous chunk `prv`
  auto body = [&](auto DV, auto DM, const unsigned char* cur, const unsigned char* prv) {
    constexpr bool V = decltype(DV)::value, M = decltype(DM)::value;
    float t16[16];
    h8_t bh, bl;
    if constexpr (M) {
      bh = *reinterpret_cast<const h8_t*>(prv + RECB + lane * 16);
      bl = *reinterpret_cast<const h8_t*>(prv + RECB + kF16WBytes + lane * 16);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (V) {
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
            const float df = xq[t][d] - r[d];                // flax_rbf.py:280
            r2 = __builtin_fmaf(df, df, r2);
          }
          t16[t * 8 + j] = f16_arg<BC>(r2, r[RF - 1]);
        }
      }
      if constexpr (M) {
        if (j < NT) {                                        // column tile j of the previous step
          h8_t nbh = bh, nbl = bl;
          if (j + 1 < NT) {
            nbh = *reinterpret_cast<const h8_t*>(prv + RECB + (j + 1) * 2 * kF16WBytes + lane * 16);
            nbl = *reinterpret_cast<const h8_t*>(prv + RECB + (j + 1) * 2 * kF16WBytes + kF16WBytes + lane * 16);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][j], 0, 0, 0);
            acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acl[t][j], 0, 0, 0);
            acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acl[t][j], 0, 0, 0);
          }
          bh = nbh; bl = nbl;
        }
      }
    }
    if constexpr (V) {
      trans_block<BC, 16>(t16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned wh[4], wl[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) split_pair_f16<3, true>(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
        ah[t] = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
        al[t] = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
      }
    }
  };

  // The accumulating code exists ONCE inside the loop (a second accumulating body in the loop makes hipcc give the 112
  // accumulator registers a different assignment per body and copy / spill them at the joins): step 0 (no products
  // pending) is peeled in front, the drain (no chunk left) behind; a slice with fewer chunks idles through the barriers.
  // Chunks are requested kWideAhead steps ahead: at the top of step i only chunk i has to be there, the requests of the
  // later chunks (`per` per chunk and wave) may still be in flight.
  const int per = qg < NVI ? (NVI - qg + QG - 1) / QG : 0;   // DMA instructions this wave issues per chunk
  auto wait_all_but = [&](int k) {                           // s_waitcnt vmcnt(k), k wave-uniform
    switch (k) {
#define IRBFN_VMCNT(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory"); break;
      IRBFN_VMCNT(1) IRBFN_VMCNT(2) IRBFN_VMCNT(3) IRBFN_VMCNT(4) IRBFN_VMCNT(5) IRBFN_VMCNT(6) IRBFN_VMCNT(7) IRBFN_VMCNT(8)
      IRBFN_VMCNT(9) IRBFN_VMCNT(10) IRBFN_VMCNT(12) IRBFN_VMCNT(14) IRBFN_VMCNT(15) IRBFN_VMCNT(16) IRBFN_VMCNT(18)
#undef IRBFN_VMCNT
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto in_flight = [&](int i) {                              // chunks i + 1 .. i + kWideAhead - 1 that exist
    int m = na - 1 - i;
    m = m < kWideAhead - 1 ? m : kWideAhead - 1;
    return m > 0 ? m * per : 0;
  };
  // (a bare s_barrier: __syncthreads() would put a vmcnt(0) in front of it and wait for the chunks still in flight)
  auto ring_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto next_buf = [&](int b) { return b + 1 == kWideRing ? 0 : b + 1; };
  for (int k = 0; k < kWideAhead; ++k)
    if (k < na) stage(c0 + k, ring + k * CB);
  wait_all_but(in_flight(0));
  ring_barrier();
  if (kWideAhead < na) stage(c0 + kWideAhead, ring + kWideAhead * CB);
  if (na > 0) body(BoolC<true>{}, BoolC<false>{}, ring, ring);
  int bprv = 0, bcur = 1 % kWideRing, bnew = (1 + kWideAhead) % kWideRing;   // ring slots of chunk i - 1, i, i + kWideAhead
  for (int i = 1; i < nsteps; ++i) {
    wait_all_but(in_flight(i));                              // this wave's share of chunk i has landed ...
    ring_barrier();                                          // ... and everybody's; step i - 1 is over in every wave
    if (i + kWideAhead < na) stage(c0 + i + kWideAhead, ring + bnew * CB);   // the slot last read during step i - 1 at the latest
    if (i < na) {
      body(BoolC<true>{}, BoolC<true>{}, ring + bcur * CB, ring + bprv * CB);
      bprv = bcur; bcur = next_buf(bcur); bnew = next_buf(bnew);
    }
  }
  if (na > 0) body(BoolC<false>{}, BoolC<true>{}, ring, ring + bprv * CB);      // the products of the slice's last chunk

  wide_epilogue<DC, NT, MODE>(a, rl, lds, acc, acl, slice, qg, q0, 1.0f / (16384.0f * kWScale));
}


}  // namespace irbfn
