// K3: stand-alone roll-out kernels (forward).  One lane integrates one trajectory, the state lives
// in VGPRs for all T steps.  Traffic is 1.8 KB per trajectory at T = 50 against ~10 kFLOP, so HBM is
// the roof -- but only if enough waves are resident to hide the long dependent trig chains of a step.
//
// Replaces integrate_st_mult (src/irbfn_mpc/dynamics.py:94-100), dynamic_st_onestep_aux (:103-187),
// integrate_frenet_mult (:284-290), the inline bicycle of scripts/train_nmpc.py:329-374 and
// integrate_path_mult (src/irbfn_mpc/planner_utils.py:62-77).
#include <stdlib.h>

#include "common.h"
#include "rollout_step.h"
#include "rollout_pair.h"

namespace irbfn {

int rollout_state_dim(int mode) {
  switch (mode) {
    case IRBFN_ROLLOUT_ST_SELECT:
    case IRBFN_ROLLOUT_ST_KS: return 7;
    case IRBFN_ROLLOUT_FULLINT: return 5;
    case IRBFN_ROLLOUT_FRENET_LS: return 8;
    case IRBFN_ROLLOUT_SPIRAL: return 6;
    default: return -1;
  }
}

int rollout_input_dim(int mode, int T) {
  switch (mode) {
    case IRBFN_ROLLOUT_ST_SELECT:
    case IRBFN_ROLLOUT_ST_KS: return 7 + 2 * T;
    case IRBFN_ROLLOUT_FULLINT: return 1 + 2 * T;
    case IRBFN_ROLLOUT_FRENET_LS: return 8 + 2 * T;
    case IRBFN_ROLLOUT_SPIRAL: return 5;
    default: return -1;
  }
}

struct RollArgs {
  const float* __restrict__ x0;    // initial-state rows: row b at x0 + b*L0   (combined layout: x0u, L0 = L)
  const float* __restrict__ u;     // control rows [a_0.., sv_0..]: row b at u + b*LU (combined: x0u + S0, LU = L)
  long L0, LU;
  float* __restrict__ states;      // [B][T][S]
  long B;
  int T, L;
  int wlds;                        // floats of LDS per wave (regs kernel)
  int dma_ok;                      // input buffers 16-byte aligned: whole-tile LDS-DMA prologue allowed
  DynParams dp;
};

// 16-byte vector with 4-byte alignment: the input rows are only dword aligned (L is odd), gfx950
// handles the unaligned global_load_dwordx4
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kRollWaves = 4;      // waves per workgroup (independent; wave-private LDS)
constexpr int kRollTS = 4;         // lean kernel: steps staged per flush = controls fetched per 16-byte load
constexpr int kRollRPP = 16;       // regs kernel: rows per LDS-DMA input pass
constexpr int kRegsWaves = 2;      // regs kernel: waves per workgroup (17 KB of LDS each)

constexpr int roll_pitch(int S, int TS) { return (31 + TS * S) | 1; }   // floats per row window (leftover < 32 + one chunk), odd
constexpr int roll_ts(int S) { return (32 + S - 1) / S; }               // steps per chunk: the fewest with TS * S >= 32
typedef const __attribute__((address_space(1))) void* gptr_t;          // operands of __builtin_amdgcn_global_load_lds
typedef __attribute__((address_space(3))) void* lptr_t;

// K3 `rollout_fwd_regs_kernel<MODE, TCH, TS>`: the roll-out for horizons T <= TCH (BASELINE config 4: T = 50).
// HBM-bound by its bytes (1828 B per trajectory at T = 50 against ~8 kFLOP); what it takes to get there:
//  * every input row is touched ONCE.  A wave's 64 rows are one contiguous block of HBM (64 L floats); it is copied
//    in RPP-row passes by LDS-DMA (global_load_lds_dwordx4: whole aligned kilobytes, no staging VGPRs) and each
//    lane then pulls its own row into REGISTERS: the 2T controls of a trajectory live in VGPRs for the whole
//    roll-out (the register file is the one on-chip store large enough: 428 B per trajectory in flight; LDS is
//    not).  Round 1 fetched per-row control chunks twice per stream: FETCH_SIZE 2.6x the algorithmic bytes.
//  * the step loop is unrolled in groups (static register indices, rotated between groups), everything that
//    depends on T is wave-uniform;
//  * states leave as WHOLE 128-byte lines.  Each row has a window in a wave-private LDS tile that starts on a line
//    boundary of HBM (h_r floats before the row's first state); a chunk of TS steps appends TS*S >= 32 floats, the
//    complete lines (1 or 2 per row and chunk) are stored by 8 lanes x 16 bytes each, the < 32 leftover floats
//    move to the window front.  Only the first and the last line of a row (shared with its neighbours) are
//    written in part.  Writing 112-byte runs instead (round 1, and the first version of this kernel) left every
//    line open in L2 until the next flush 4 steps later: with ~3 MB of open lines per XCD they were evicted half
//    written -- WRITE_SIZE 1.31x the algorithmic bytes and the L1 stalled on the write path 90 % of the time
//    (profiles/r02_rollout_pmc_regs_kernel_112B_runs.txt);
//  * nothing in the loop waits for memory: the stores are fire-and-forget (no load is outstanding after the
//    prologue, so no s_waitcnt vmcnt is ever needed), LDS hazards are the wave's own in-order queue.
// Bit-identical to the other roll-out kernels: the step functions are shared (rollout_step.h, no contraction).
#ifndef IRBFN_ROLL_MINW
#define IRBFN_ROLL_MINW 3          // waves per SIMD the long-horizon instance is allocated for (100 control VGPRs)
#endif
template <int MODE, int TCH, int TS>
__global__ __launch_bounds__(64 * kRegsWaves, TCH > 8 ? IRBFN_ROLL_MINW : 4) void rollout_fwd_regs_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int S0 = ModeTraits<MODE>::S0;
  constexpr int CF = TS * S;                     // floats a full chunk appends to every row window
  constexpr int PITCH = roll_pitch(S, TS);       // odd: conflict-free per-lane row access
  constexpr int RPP = kRollRPP;                  // rows per input pass
  constexpr bool HAS_U = MODE != IRBFN_ROLLOUT_SPIRAL;
  static_assert(CF >= 32 && CF <= 64, "a chunk completes at least one 128-byte line per row and at most two (+1 with the leftover)");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long b0 = ((long)blockIdx.x * kRegsWaves + wave) * kWave;
  if (b0 >= a.B) return;                         // whole wave out of range (no block-level barriers used)
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const int T = a.T;
  float* tile = lds + (size_t)wave * a.wlds;
  float* mine = tile + lane * PITCH;
  auto lds_drain = [&]() {                       // wave-private LDS: the wave's own reads / writes have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  // ---- prologue: this lane's row -> registers -------------------------------------------------------------
  float s[S];
  float ua[TCH], us[TCH];
#pragma unroll
  for (int t = 0; t < TCH; ++t) { ua[t] = 0.0f; us[t] = 0.0f; }
  float q0[S0];                                  // raw leading floats of the row (state / spiral parameters)
  const bool dma = nvalid == kWave && a.dma_ok;  // tail tile / unaligned buffers: per-lane loads
  if (dma) {
    const bool split = a.x0 != a.u - S0 || a.L0 != a.LU;     // state rows and control rows in separate buffers
#pragma unroll 1
    for (int p = 0; p < kWave / RPP; ++p) {
      // pass p: rows [p*RPP, (p+1)*RPP) of the tile, contiguous in HBM, as whole 16-byte pieces
      const long r0 = b0 + (long)p * RPP;
      int nf0, nf1 = 0;                          // floats of region 0 (rows incl. state) and region 1 (split: controls)
      const float* src0 = a.x0 + r0 * a.L0;
      const float* src1 = nullptr;
      if (split) { nf0 = RPP * (int)a.L0; nf1 = RPP * (int)a.LU; src1 = a.u + r0 * a.LU; }
      else nf0 = RPP * (int)a.L0;
      for (int v = lane * 4; v < nf0; v += 256)
        __builtin_amdgcn_global_load_lds((gptr_t)(src0 + v), (lptr_t)(tile + (v - lane * 4)), 16, 0, 0);
      for (int v = lane * 4; v < nf1; v += 256)
        __builtin_amdgcn_global_load_lds((gptr_t)(src1 + v), (lptr_t)(tile + nf0 + (v - lane * 4)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if ((lane / RPP) == p) {
        const float* rr = tile + (lane % RPP) * (int)a.L0;
#pragma unroll
        for (int i = 0; i < S0; ++i) q0[i] = rr[i];
        if constexpr (HAS_U) {
          const float* ur = split ? tile + nf0 + (lane % RPP) * (int)a.LU : rr + S0;
#pragma unroll
          for (int t = 0; t < TCH; ++t) {        // u = [a_0.., sv_0..] (dynamics.py:98); slots t >= T are never used
            ua[t] = ur[t];                       // (an LDS read past the wave's tile returns junk or 0, never faults)
            us[t] = ur[T + t];
          }
        }
      }
      lds_drain();
    }
  } else {
    const long bb = b0 + (lane < nvalid ? lane : nvalid - 1);
    const float* row = a.x0 + bb * a.L0;
    const float* urow = a.u + bb * a.LU;
#pragma unroll
    for (int i = 0; i < S0; ++i) q0[i] = row[i];
    if constexpr (HAS_U) {
#pragma unroll
      for (int t = 0; t < TCH; ++t)
        if (t < T) { ua[t] = urow[t]; us[t] = urow[T + t]; }
    }
  }
  [[maybe_unused]] float coef[4];
  [[maybe_unused]] float slen = 0.0f;
  [[maybe_unused]] float spc[2] = {0.0f, 1.0f};      // spiral: (sin, cos) of the current heading, carried step to step
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(q0[0], 0.0f, 7.0f);             // train_nmpc.py:319
  } else if constexpr (MODE == IRBFN_ROLLOUT_SPIRAL) {
    spiral_coefs(q0, coef);
    slen = q0[4];
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[3] = coef[0]; s[4] = 0.0f; s[5] = 0.0f;   // planner_utils.py:67-70
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = q0[i];
  }

  // ---- steps + whole-line flush ------------------------------------------------------------------------------
  float* gout = a.states + b0 * (long)T * S;     // tile base in HBM; row stride T*S
  const long rs = (long)T * S;
  const int g32 = (int)((reinterpret_cast<uintptr_t>(gout) >> 2) & 31);
  auto head_of = [&](int r) { return (g32 + (int)((r * rs) & 31)) & 31; };   // floats between the line start and row r
  const int myH = head_of(lane);
  int fill = myH;                                // floats in my window (the first myH of them are not mine)
  // lines [0, nl_r) of every row window -> HBM.  p0 = floats each row had produced before this chunk.
  auto flush = [&](long p0, int nfl, bool last) {
    const int sub = lane >> 3, k = lane & 7;     // 8 lanes per line
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll 1
      for (int j = 0; j < kWave / 8; ++j) {
        const int r = j * 8 + sub;
        const int h = head_of(r);
        const int f0 = (int)((h + p0) & 31);     // floats in row r's window before the chunk
        const int tot = f0 + nfl;
        const int nl = tot >> 5;                 // complete lines now in the window
        const bool tail = last && l == nl && (tot & 31) > 0;             // the row's last, partial line
        if (r < nvalid && (l < nl || tail)) {
          const long q = (long)h + p0 - f0 + 32 * l;                     // line start, floats from the window origin W_r
          const float* src = tile + r * PITCH + 32 * l + 4 * k;
          float* dst = gout + r * rs - h + q + 4 * k;                    // 16-byte aligned
          const float v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
          int lo = (q == 0) ? h - 4 * k : 0;                             // first line of the row: floats [0, h) are not ours
          int hi = tail ? (tot & 31) - 4 * k : 4;                        // last line: only (tot & 31) floats exist
          if (lo <= 0 && hi >= 4) {
            *reinterpret_cast<float4*>(dst) = float4{v0, v1, v2, v3};
          } else {
            if (lo <= 0 && hi > 0) dst[0] = v0;
            if (lo <= 1 && hi > 1) dst[1] = v1;
            if (lo <= 2 && hi > 2) dst[2] = v2;
            if (lo <= 3 && hi > 3) dst[3] = v3;
          }
        }
      }
    }
  };
  // The step loop is ROLLED over groups of G steps (a straight-line 50-step body is ~70 KB of code: measured
  // instruction-fetch bound, 68 us per wave); inside a group the controls sit at static register indices, between
  // groups the control registers rotate down by G (2 (TCH - G) v_mov per group, ~5 % of a group's instructions).
  constexpr int G = TCH < 2 * TS ? ((TCH + TS - 1) / TS) * TS : 2 * TS;
  const int ngroups = (T + G - 1) / G;
#pragma unroll 1
  for (int gI = 0; gI < ngroups; ++gI) {
    const int tg = gI * G;
#pragma unroll
    for (int t0 = 0; t0 < G; t0 += TS) {
      if (tg + t0 < T) {                         // wave-uniform
        const int n = (T - tg - t0) < TS ? (T - tg - t0) : TS;
        float* wr = mine + fill;
#pragma unroll
        for (int tt = 0; tt < TS; ++tt) {
          if (t0 + tt < TCH && tt < n) {
            if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua[t0 + tt], us[t0 + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua[t0 + tt], us[t0 + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua[t0 + tt], us[t0 + tt]);
            else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ua[t0 + tt], us[t0 + tt], a.dp);
            else spiral_step(s, coef, slen, tg + t0 + tt, T, spc);
#pragma unroll
            for (int i = 0; i < S; ++i) wr[tt * S + i] = s[i];
          }
        }
        lds_drain();
        flush((long)(tg + t0) * S, n * S, tg + t0 + n >= T);
        // the floats behind the last complete line move to the window front
        const int tot = fill + n * S;
        const float* from = mine + (tot & ~31);
        float keep[31];
#pragma unroll
        for (int i = 0; i < 31; ++i) keep[i] = from[i];
        lds_drain();
        if (tot >= 32) {
#pragma unroll
          for (int i = 0; i < 31; ++i) mine[i] = keep[i];
        }
        fill = tot & 31;
      }
    }
    if constexpr (HAS_U && TCH > G) {
#pragma unroll
      for (int i = 0; i + G < TCH; ++i) { ua[i] = ua[i + G]; us[i] = us[i + G]; }
    }
  }
}

// K3p `rollout_fwd_pair_kernel<MODE, TCH, TS>`: TWO adjacent lanes per trajectory (32 trajectories per wave).
// The step's two transcendental evaluations -- sincos of the heading, tan of the steering angle -- run in ONE
// instruction stream (even lane: heading, odd lane: steering angle; TrigPair swaps the results by DPP), everything
// else is evaluated redundantly by both lanes from the same operands: bit-identical to the one-lane step, ~40 %
// fewer instructions per step, twice the waves.  What that buys:
//  * small batches (the 32768-trajectory per-GPU share of config 4 is 512 one-lane waves for 1024 SIMDs): every
//    SIMD gets a wave and the wave's serial chain is shorter;
//  * large batches: half the trajectories in flight per resident wave, so half as many output rows are open at any
//    time (DRAM page locality of the write stream) and a row window can hold TS = 10 steps: every flush writes
//    2-3 whole lines per row;
//  * the even lane keeps the acceleration knots, the odd lane the steering-rate knots (50 control VGPRs per lane
//    instead of 100; each step broadcasts its pair by DPP).
// Input / output machinery as in rollout_fwd_regs_kernel: whole-tile LDS-DMA prologue, row windows on 128-byte
// line boundaries, whole-line stores, leftover compaction (split between the two lanes).
template <int MODE, int TCH, int TS>
__global__ __launch_bounds__(64 * kPairWaves, IRBFN_PAIR_MINW) void rollout_fwd_pair_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int S0 = ModeTraits<MODE>::S0;
  constexpr int SE = (S + 1) / 2;                // state components the even lane writes; the odd lane writes S - SE
  constexpr int CF = TS * S;
  constexpr int PITCH = pair_pitch(S, TS);
  constexpr int RPP = kRollRPP;                  // two 16-row input passes: one pass of 32 rows needs 13.7 KB of LDS per wave and
                                                 // costs the third resident workgroup per CU (126 vs 107 us at B = 262144)
  constexpr int LMAX = ((31 + CF) >> 5) + 1;     // lines a row window can complete in one chunk, + the tail line
  static_assert(MODE != IRBFN_ROLLOUT_SPIRAL, "the spiral has no control knots and no tan: one lane per path");
  static_assert(CF >= 32, "a chunk completes at least one 128-byte line per row");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int odd = lane & 1, prow = lane >> 1;
  const long b0 = ((long)blockIdx.x * kPairWaves + wave) * kPairRows;
  if (b0 >= a.B) return;
  const long left = a.B - b0;
  const int nvalid = left < kPairRows ? (int)left : kPairRows;
  const int T = a.T;
  float* tile = lds + (size_t)wave * a.wlds;
  float* mine = tile + prow * PITCH;
  auto lds_drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  // ---- prologue: the row's state (both lanes) and ONE control stream per lane -> registers -------------------
  float s[S], q0[S0], ctl[TCH];
#pragma unroll
  for (int t = 0; t < TCH; ++t) ctl[t] = 0.0f;
  const bool dma = nvalid == kPairRows && a.dma_ok;
  if (dma) {
    const bool split = a.x0 != a.u - S0 || a.L0 != a.LU;
#pragma unroll 1
    for (int p = 0; p < kPairRows / RPP; ++p) {
      const long r0 = b0 + (long)p * RPP;
      int nf0 = RPP * (int)a.L0, nf1 = 0;
      const float* src0 = a.x0 + r0 * a.L0;
      const float* src1 = nullptr;
      if (split) { nf1 = RPP * (int)a.LU; src1 = a.u + r0 * a.LU; }
      for (int v = lane * 4; v < nf0; v += 256)
        __builtin_amdgcn_global_load_lds((gptr_t)(src0 + v), (lptr_t)(tile + (v - lane * 4)), 16, 0, 0);
      for (int v = lane * 4; v < nf1; v += 256)
        __builtin_amdgcn_global_load_lds((gptr_t)(src1 + v), (lptr_t)(tile + nf0 + (v - lane * 4)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if ((prow / RPP) == p) {
        const float* rr = tile + (prow % RPP) * (int)a.L0;
#pragma unroll
        for (int i = 0; i < S0; ++i) q0[i] = rr[i];
        const float* ur = (split ? tile + nf0 + (prow % RPP) * (int)a.LU : rr + S0) + (odd ? T : 0);
#pragma unroll
        for (int t = 0; t < TCH; ++t) ctl[t] = ur[t];   // slots t >= T are never used (an LDS read past the tile is harmless)
      }
      lds_drain();
    }
  } else {
    const long bb = b0 + (prow < nvalid ? prow : nvalid - 1);
    const float* row = a.x0 + bb * a.L0;
    const float* urow = a.u + bb * a.LU + (odd ? T : 0);
#pragma unroll
    for (int i = 0; i < S0; ++i) q0[i] = row[i];
#pragma unroll
    for (int t = 0; t < TCH; ++t)
      if (t < T) ctl[t] = urow[t];
  }
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(q0[0], 0.0f, 7.0f);             // train_nmpc.py:319
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = q0[i];
  }

  pair_rollout_run<MODE, TCH, TS>(s, ctl, a.dp, tile, a.states + b0 * (long)T * S, T, nvalid, lane);
}

// K3b: the roll-out with a ROLLED step loop, for horizons beyond the unroll depth of rollout_fwd_regs_kernel.
// A group of 4 steps is the unit: the group's controls (2 streams x 4 values per row) are staged
// through the SAME tile that stages the group's output states (aligned 16-byte pieces fetched one group ahead
// into 4 VGPRs per lane), so a wave needs ~90 VGPRs and one 9.5 KB tile: 4 waves per SIMD.
template <int MODE>
__global__ __launch_bounds__(64 * kRollWaves, 3) void rollout_fwd_lean_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int TS = kRollTS;
  constexpr int CF = TS * S;
  constexpr int NP = CF / 4;
  constexpr int PITCH = 37;
  static_assert(CF % 4 == 0 && CF + 3 <= PITCH && NP <= 8 && TS == 4, "lean roll-out: 4-step groups");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long b0 = ((long)blockIdx.x * kRollWaves + wave) * kWave;
  if (b0 >= a.B) return;
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const bool last_tile = left <= kWave;
  const int T = a.T;
  const long bb = b0 + (lane < nvalid ? lane : nvalid - 1);
  const float* row = a.x0 + bb * a.L0;
  const float* urow = a.u + bb * a.LU;
  float* tile = lds + wave * (kWave * PITCH);
  float* mine = tile + lane * PITCH;
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float* gout = a.states + b0 * (long)T * S;
  auto carry_of = [&](int r) { return (int)((reinterpret_cast<uintptr_t>(gout + (long)r * T * S) >> 2) & 3); };
  const int myC = carry_of(lane < nvalid ? lane : 0);

  float s[S];
  [[maybe_unused]] float coef[4];
  [[maybe_unused]] float slen = 0.0f;
  [[maybe_unused]] float spc[2] = {0.0f, 1.0f};      // spiral: (sin, cos) of the current heading, carried step to step
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(row[0], 0.0f, 7.0f);
  } else if constexpr (MODE == IRBFN_ROLLOUT_SPIRAL) {
    float q[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] = row[i];
    spiral_coefs(q, coef);
    slen = q[4];
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[3] = coef[0]; s[4] = 0.0f; s[5] = 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = row[i];
  }

  // control pieces: per row 2 streams x 2 aligned float4 (the 4 wanted floats lie inside 8 aligned ones);
  // lane l covers rows l/4 + 16 j (j < 4), stream (l >> 1) & 1, half l & 1
  const int crow = lane >> 2, cstream = (lane >> 1) & 1, chalf = lane & 1;
  float4 pre[4];
  auto fetch_ctrl = [&](int t0) {                // controls of steps [t0, t0 + 4) of every row -> pre
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r = crow + 16 * j;
      r = r < nvalid ? r : nvalid - 1;
      const float* g0 = a.u + (b0 + r) * a.LU + (cstream ? T : 0) + t0;
      const float* al = g0 - (int)((reinterpret_cast<uintptr_t>(g0) >> 2) & 3);
      pre[j] = *reinterpret_cast<const float4*>(al + 4 * chalf);
    }
  };
  auto stash_ctrl = [&]() {                      // pre -> tile[row][stream * 8 + 4 half ..]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = crow + 16 * j;
      float* t = tile + r * PITCH + 3 + cstream * 8 + 4 * chalf;       // [0, 3) stays free for the carry
      t[0] = pre[j].x; t[1] = pre[j].y; t[2] = pre[j].z; t[3] = pre[j].w;
    }
  };
  auto flush = [&](int t0, bool first) {
    const int rsub = lane >> 3, part = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = j * 8 + rsub;
      if (r < nvalid && part < NP) {
        const int C = carry_of(r);
        const float* src = tile + r * PITCH + 4 * part;
        float* dst = gout + (long)r * T * S + (long)t0 * S - C + 4 * part;
        if (first && part == 0 && C > 0) {
          for (int i = C; i < 4; ++i) dst[i] = src[i];
        } else {
          *reinterpret_cast<float4*>(dst) = float4{src[0], src[1], src[2], src[3]};
        }
      }
    }
  };

  const bool has_ctrl = MODE != IRBFN_ROLLOUT_SPIRAL;
  const int ngroups = T / TS;                    // full 4-step groups; the < 4 remaining steps go out as dwords
  const bool vec = has_ctrl && !last_tile;       // the aligned superset may reach past the buffer end on the last tile
  if (vec && ngroups > 0) fetch_ctrl(0);
#pragma unroll 1
  for (int gI = 0; gI < ngroups; ++gI) {
    const int t0 = gI * TS;
    float ua[TS], us[TS];
    if (has_ctrl) {
      if (vec) {
        float keep[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = mine[i];      // carry lives in [0, 3): untouched by stash_ctrl
        (void)keep;
        stash_ctrl();
        wave_sync();
        const int sha = (int)((reinterpret_cast<uintptr_t>(urow + t0) >> 2) & 3);
        const int shs = (int)((reinterpret_cast<uintptr_t>(urow + T + t0) >> 2) & 3);
#pragma unroll
        for (int i = 0; i < TS; ++i) { ua[i] = mine[3 + sha + i]; us[i] = mine[3 + 8 + shs + i]; }
        wave_sync();
        if (gI + 1 < ngroups) fetch_ctrl(t0 + TS);          // one group ahead
      } else {
#pragma unroll
        for (int i = 0; i < TS; ++i) { ua[i] = urow[t0 + i]; us[i] = urow[T + t0 + i]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TS; ++i) { ua[i] = 0.0f; us[i] = 0.0f; }
    }
#pragma unroll
    for (int tt = 0; tt < TS; ++tt) {
      if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua[tt], us[tt], a.dp);
      else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua[tt], us[tt], a.dp);
      else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua[tt], us[tt]);
      else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ua[tt], us[tt], a.dp);
      else spiral_step(s, coef, slen, t0 + tt, T, spc);
#pragma unroll
      for (int i = 0; i < S; ++i) mine[myC + tt * S + i] = s[i];
    }
    wave_sync();
    flush(t0, t0 == 0);
    wave_sync();
#pragma unroll
    for (int i = 0; i < 3; ++i) {                // carry the last C floats to the front of the window
      const float v = mine[CF + i];
      if (i < myC) mine[i] = v;
    }
    wave_sync();
  }
  // tail: T mod 4 steps, then whatever is left in the window as dwords
  const int done = ngroups * TS;
  int fill = myC;
  for (int t = done; t < T; ++t) {
    const float ca = has_ctrl ? urow[t] : 0.0f, cs = has_ctrl ? urow[T + t] : 0.0f;
    if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ca, cs, a.dp);
    else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ca, cs, a.dp);
    else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ca, cs);
    else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ca, cs, a.dp);
    else spiral_step(s, coef, slen, t, T, spc);
#pragma unroll
    for (int i = 0; i < S; ++i) mine[fill + i] = s[i];
    fill += S;
  }
  if (lane < nvalid) {
    const int first_unflushed = done * S - (done > 0 ? myC : 0);
    const int skip = done > 0 ? 0 : myC;
    float* dst = gout + (long)lane * T * S + first_unflushed;
    for (int i = skip; i < fill; ++i) dst[i - skip] = mine[i];
  }
}

constexpr int kRollTchLong = 50, kRollTchShort = 8;    // compiled unroll depths: T <= 8 (the reference's 5), T <= 50

template <int MODE>
static int launch_mode(const RollArgs& a0, hipStream_t s) {
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int TS = roll_ts(S);
  RollArgs a = a0;
  const long waves = (a.B + kWave - 1) / kWave;
  const long grid = (waves + kRollWaves - 1) / kRollWaves;
  const long rgrid = (waves + kRegsWaves - 1) / kRegsWaves;
  if constexpr (MODE != IRBFN_ROLLOUT_SPIRAL) {
    if (a.T <= kRollTchLong && a.T > kRollTchShort) {
      // K3p: two lanes per trajectory
      constexpr int PTS = pair_ts(S);
      const bool split = a.x0 != a.u - ModeTraits<MODE>::S0 || a.L0 != a.LU;
      long w = (long)kPairRows * pair_pitch(S, PTS);
      const long in_floats = (long)kRollRPP * (a.L0 + (split ? a.LU : 0));
      if (in_floats > w) w = in_floats;
      a.wlds = (int)((w + 3) & ~3L);
      a.dma_ok = ((reinterpret_cast<uintptr_t>(a.x0) | reinterpret_cast<uintptr_t>(a.u)) & 15) == 0 ||
                 (!split && (reinterpret_cast<uintptr_t>(a.x0) & 15) == 0);
      const size_t lds = (size_t)kPairWaves * a.wlds * sizeof(float);
      if (lds <= 64 * 1024) {
        const long pw = (a.B + kPairRows - 1) / kPairRows;
        hipLaunchKernelGGL((rollout_fwd_pair_kernel<MODE, kRollTchLong, PTS>), dim3((unsigned)((pw + kPairWaves - 1) / kPairWaves)),
                           dim3(kWave * kPairWaves), lds, s, a);
        IRBFN_HIP_CHECK(hipGetLastError());
        return IRBFN_OK;
      }
    }
  }
  if (a.T <= kRollTchLong) {
    // K3: controls in registers, whole-tile LDS-DMA input, fire-and-forget aligned stores
    const bool has_u = MODE != IRBFN_ROLLOUT_SPIRAL;
    const bool split = has_u && (a.x0 != a.u - ModeTraits<MODE>::S0 || a.L0 != a.LU);
    long in_floats = (long)kRollRPP * (a.L0 + (split ? a.LU : 0));
    long w = (long)kWave * roll_pitch(S, TS);
    if (in_floats > w) w = in_floats;
    a.wlds = (int)((w + 3) & ~3L);
    a.dma_ok = ((reinterpret_cast<uintptr_t>(a.x0) | reinterpret_cast<uintptr_t>(a.u)) & 15) == 0 ||
               (!split && (reinterpret_cast<uintptr_t>(a.x0) & 15) == 0);
    const size_t lds = (size_t)kRegsWaves * a.wlds * sizeof(float);
    if (lds <= 64 * 1024) {
      if (a.T <= kRollTchShort)
        hipLaunchKernelGGL((rollout_fwd_regs_kernel<MODE, kRollTchShort, TS>), dim3((unsigned)rgrid), dim3(kWave * kRegsWaves),
                           lds, s, a);
      else
        hipLaunchKernelGGL((rollout_fwd_regs_kernel<MODE, kRollTchLong, TS>), dim3((unsigned)rgrid), dim3(kWave * kRegsWaves),
                           lds, s, a);
      IRBFN_HIP_CHECK(hipGetLastError());
      return IRBFN_OK;
    }
  }
  // longer horizons: rolled 4-step groups, controls staged through LDS group by group
  const size_t lds = (size_t)kRollWaves * kWave * 37 * sizeof(float);
  hipLaunchKernelGGL(rollout_fwd_lean_kernel<MODE>, dim3((unsigned)grid), dim3(kWave * kRollWaves), lds, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

static int dispatch_mode(int mode, const RollArgs& a, hipStream_t s) {
  switch (mode) {
    case IRBFN_ROLLOUT_ST_SELECT: return launch_mode<IRBFN_ROLLOUT_ST_SELECT>(a, s);
    case IRBFN_ROLLOUT_ST_KS: return launch_mode<IRBFN_ROLLOUT_ST_KS>(a, s);
    case IRBFN_ROLLOUT_FULLINT: return launch_mode<IRBFN_ROLLOUT_FULLINT>(a, s);
    case IRBFN_ROLLOUT_FRENET_LS: return launch_mode<IRBFN_ROLLOUT_FRENET_LS>(a, s);
    case IRBFN_ROLLOUT_SPIRAL: return launch_mode<IRBFN_ROLLOUT_SPIRAL>(a, s);
    default: return IRBFN_ERR_BAD_ARG;
  }
}

// Measured and rejected: splitting B > 131072 into several launches (hoping the 256 MB memory-side cache would
// merge the partial lines of one launch): 216 vs 190 us at B = 262144 -- the 71 us seen for a single 131072
// launch is an artefact of re-running on cache-resident buffers, not a property of the size.
static int s0_of(int mode) {
  switch (mode) {
    case IRBFN_ROLLOUT_FULLINT: return 1;
    case IRBFN_ROLLOUT_FRENET_LS: return 8;
    case IRBFN_ROLLOUT_SPIRAL: return 5;
    default: return 7;
  }
}

// combined layout of the reference: row = [state, a_0..a_{T-1}, sv_0..sv_{T-1}]
int launch_rollout_forward(int mode, const float* x0u, const DynParams& dp, float* states, int64_t B,
                           int T, hipStream_t s) {
  if (B == 0 || T == 0) return IRBFN_OK;
  RollArgs a;
  a.L = rollout_input_dim(mode, T);
  a.x0 = x0u;
  a.u = x0u + s0_of(mode);
  a.L0 = a.L;
  a.LU = a.L;
  a.states = states;
  a.B = (long)B;
  a.T = T;
  a.wlds = 0;
  a.dma_ok = 0;
  a.dp = dp;
  return dispatch_mode(mode, a, s);
}

// split layout: initial states [B][S0] and controls [B][2T] in separate buffers (the planning tick:
// no hstack((states, pred_u)) copy, src/irbfn_mpc/irbfn_planner.py:209-210)
int launch_rollout_forward_split(int mode, const float* state0, const float* controls, const DynParams& dp,
                                 float* states, int64_t B, int T, hipStream_t s) {
  if (B == 0 || T == 0) return IRBFN_OK;
  if (mode == IRBFN_ROLLOUT_SPIRAL) return IRBFN_ERR_BAD_ARG;
  RollArgs a;
  a.L = rollout_input_dim(mode, T);
  a.x0 = state0;
  a.u = controls;
  a.L0 = s0_of(mode);
  a.LU = 2L * T;
  a.states = states;
  a.B = (long)B;
  a.T = T;
  a.wlds = 0;
  a.dma_ok = 0;
  a.dp = dp;
  return dispatch_mode(mode, a, s);
}

}  // namespace irbfn
