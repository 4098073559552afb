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

template <int MODE>
struct ModeTraits;
template <>
struct ModeTraits<IRBFN_ROLLOUT_ST_SELECT> { static constexpr int S = 7, S0 = 7; };
template <>
struct ModeTraits<IRBFN_ROLLOUT_ST_KS> { static constexpr int S = 7, S0 = 7; };
template <>
struct ModeTraits<IRBFN_ROLLOUT_FULLINT> { static constexpr int S = 5, S0 = 1; };
template <>
struct ModeTraits<IRBFN_ROLLOUT_FRENET_LS> { static constexpr int S = 8, S0 = 8; };
template <>
struct ModeTraits<IRBFN_ROLLOUT_SPIRAL> { static constexpr int S = 6, S0 = 5; };

struct RollArgs {
  const float* __restrict__ x0;    // initial-state rows: row b at x0 + b*L0   (combined layout: x0u, L0 = L)
  const float* __restrict__ u;     // control rows [a_0.., sv_0..]: row b at u + b*LU (combined: x0u + S0, LU = L)
  long L0, LU;
  float* __restrict__ states;      // [B][T][S]
  long B;
  int T, L;
  int dbg;                         // diagnosis only (IRBFN_ROLL_DBG): 1 = skip stores, 2 = skip control loads
  DynParams dp;
};

// 16-byte vector with 4-byte alignment: the input rows are only dword aligned (L is odd), gfx950
// handles the unaligned global_load_dwordx4
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kRollWaves = 4;      // waves per workgroup (independent; wave-private LDS)
constexpr int kRollTS = 4;         // steps staged per flush = controls fetched per 16-byte load

// What makes this kernel fast (each item measured, profiles/r01_rollout_*, tools/ubench_mem.hip):
//  * occupancy: one step is a long DEPENDENT chain, so a SIMD needs several resident waves; LDS is
//    one 64 x 37-float tile per wave (9.5 KB), shared by the input and the output staging;
//  * straight-line sin/cos/tan (rollout_step.h) instead of ocml's branchy range reduction;
//  * only ALIGNED 16-byte HBM accesses.  Rows are merely dword aligned and gfx950 splits unaligned
//    16-byte accesses (112-byte row runs: 2.2 TB/s unaligned float4 vs 4.2 TB/s aligned float2).
//    - stores: sliding window.  Row r's buffer starts C_r floats before the chunk so that it begins
//      on a 16-byte boundary of HBM; TS*S = 0 (mod 4), hence every flush emits exactly TS*S/4
//      aligned float4 per row and carries the same C_r trailing floats into the next chunk; only
//      the first C_r-complement and the last < 4 floats of a row are written as dwords.
//    - loads: per 8-step chunk the aligned float4 superset of a[t0..t0+8) (<= 3 pieces per row) is
//      fetched cooperatively, 21 rows per instruction, all loads of a stream issued back to back.
//  * measured and rejected: a whole-128-byte-line flush (64-float ring per row, 20 KB LDS per wave):
//    187-195 us vs 205 us at B = 262144 but 49-54 us vs 40 us at B = 32768 -- the store phase costs
//    ~100 us whatever the pattern (a 367 MB fill alone takes 64 us), so the simpler window stays.
template <int MODE>
__global__ __launch_bounds__(64 * kRollWaves) void rollout_fwd_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int S0 = ModeTraits<MODE>::S0;
  constexpr int TS = kRollTS;
  constexpr int CF = TS * S;                     // floats per full output chunk per row
  constexpr int NP = CF / 4;                     // aligned float4 pieces per row per flush
  constexpr int TCH = 32;                        // steps per control chunk
  constexpr int PITCH = 37;                      // >= max(CF + 3, 4*PPR), odd -> conflict-free row access
  static_assert(CF % 4 == 0 && CF + 3 <= PITCH && NP <= 8, "sliding-window flush needs TS*S = 0 (mod 4)");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long b0 = ((long)blockIdx.x * kRollWaves + wave) * kWave;
  if (b0 >= a.B) return;                         // whole wave out of range (no block-level barriers used)
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const bool last_tile = left <= kWave;          // reads near the end of the input buffer stay scalar
  const int T = a.T;
  const long bb = b0 + (lane < nvalid ? lane : nvalid - 1);
  const float* row = a.x0 + bb * a.L0;           // initial state (or spiral parameters)
  const float* urow = a.u + bb * a.LU;           // this lane's controls
  float* tile = lds + wave * (kWave * PITCH);
  float* mine = tile + lane * PITCH;
  auto wave_sync = [&]() {                       // wave-private LDS: in-order queue, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float* gout = a.states + b0 * (long)T * S;     // tile base in HBM; row stride T*S
  // leading floats of a row's window that belong to the previous 16-byte block: C = (addr / 4) mod 4
  auto carry_of = [&](int r) { return (int)((reinterpret_cast<uintptr_t>(gout + (long)r * T * S) >> 2) & 3); };
  const int myC = carry_of(lane < nvalid ? lane : 0);

  float s[S];
  [[maybe_unused]] float coef[4];
  [[maybe_unused]] float slen = 0.0f;
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(row[0], 0.0f, 7.0f);            // train_nmpc.py:319
  } else if constexpr (MODE == IRBFN_ROLLOUT_SPIRAL) {
    float q[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] = row[i];
    spiral_coefs(q, coef);
    slen = q[4];
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[3] = coef[0]; s[4] = 0.0f; s[5] = 0.0f;   // planner_utils.py:67-70
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = row[i];
  }

  // cooperative aligned fetch of one control stream chunk: floats [g0, g0 + n) of every row, g0 = r*L + off
  auto fetch_stream = [&](int off, int n, float (&dst)[TCH]) {
    if (last_tile || (a.dbg & 2)) {              // scalar path (tail tile / diagnosis)
#pragma unroll
      for (int i = 0; i < TCH; ++i) dst[i] = (a.dbg & 2) ? 0.25f : (i < n ? urow[off + i] : 0.0f);
      return;
    }
    const float* tin = a.u + b0 * a.LU;
    constexpr int PPR = (TCH + 3 + 3) / 4, RPI = kWave / PPR, NI = (kWave + RPI - 1) / RPI;   // pieces/row, rows/instr
    const int rsub = lane / PPR, part = lane - rsub * PPR;
    float4 v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {               // all loads first: one wait for the whole stream chunk
      int r = j * RPI + rsub;
      r = (r < nvalid && rsub < RPI) ? r : 0;
      const float* g0 = tin + (long)r * a.LU + off;
      // aligned-down pointer by ARITHMETIC on g0 (an integer round trip loses the address space: hipcc then emits
      // flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) after every single load, which also drains the pending stores)
      const float* al = g0 - (int)((reinterpret_cast<uintptr_t>(g0) >> 2) & 3);
      v[j] = *reinterpret_cast<const float4*>(al + 4 * part);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = j * RPI + rsub;
      if (r < nvalid && rsub < RPI) {
        float* t = tile + r * PITCH + 4 * part;
        t[0] = v[j].x; t[1] = v[j].y; t[2] = v[j].z; t[3] = v[j].w;
      }
    }
    wave_sync();
    const int sh = (int)((reinterpret_cast<uintptr_t>(urow + off) >> 2) & 3);  // my row's offset inside piece 0
#pragma unroll
    for (int i = 0; i < TCH; ++i) dst[i] = i < n ? mine[sh + i] : 0.0f;
    wave_sync();
  };

  // cooperative flush of the row windows: pieces [4p, 4p+4) of each row's buffer -> aligned float4 in HBM
  auto flush = [&](int t0, bool first) {
    const int rsub = lane >> 3, part = lane & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                // 8 x (4 LDS reads + 1 aligned 16-byte store)
      const int r = j * 8 + rsub;
      if (r < nvalid && part < NP) {
        const int C = carry_of(r);
        const float* src = tile + r * PITCH + 4 * part;
        float* dst = gout + (long)r * T * S + (long)t0 * S - C + 4 * part;     // 16-byte aligned
        if (a.dbg & 1) continue;
        if (first && part == 0 && C > 0) {       // the window starts before the row: only floats [C, 4) exist
          for (int i = C; i < 4; ++i) dst[i] = src[i];
        } else {
          // (non-temporal stores measured: 294 vs 196 us at B = 262144 -- the 112-byte runs rely on L2 merging)
          *reinterpret_cast<float4*>(dst) = float4{src[0], src[1], src[2], src[3]};
        }
      }
    }
  };

  int fill = myC;                                // floats in my window (the first C are carry / padding)
#pragma unroll 1
  for (int tc = 0; tc < T; tc += TCH) {
    float ua[TCH], us[TCH];
    if constexpr (MODE != IRBFN_ROLLOUT_SPIRAL) {
      const int n = (T - tc) < TCH ? (T - tc) : TCH;
      // the carry lives in the tile: park it in registers while the tile stages the controls
      float keep[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) keep[i] = mine[i];
      wave_sync();
      fetch_stream(tc, n, ua);                   // u = [a_0..a_{T-1}, sv_0..sv_{T-1}] (dynamics.py:98)
      fetch_stream(T + tc, n, us);
#pragma unroll
      for (int i = 0; i < 3; ++i) mine[i] = keep[i];
    }
#pragma unroll
    for (int ts = 0; ts < TCH; ts += TS) {
      const int t0 = tc + ts;
      if (t0 + TS <= T) {                        // full chunk
#pragma unroll
        for (int tt = 0; tt < TS; ++tt) {
          if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua[ts + tt], us[ts + tt], a.dp);
          else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua[ts + tt], us[ts + tt], a.dp);
          else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua[ts + tt], us[ts + tt]);
          else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ua[ts + tt], us[ts + tt], a.dp);
          else spiral_step(s, coef, slen, t0 + tt, T);
#pragma unroll
          for (int i = 0; i < S; ++i) mine[myC + tt * S + i] = s[i];
        }
        wave_sync();
        flush(t0, t0 == 0);
        wave_sync();
#pragma unroll
        for (int i = 0; i < 3; ++i) {            // carry the last C floats to the front of the window
          const float v = mine[CF + i];
          if (i < myC) mine[i] = v;
        }
        fill = myC;
      } else if (t0 < T) {                       // partial last chunk: dword stores
        const int tn = T - t0;
#pragma unroll
        for (int tt = 0; tt < TS - 1; ++tt) {      // static register indices (no scratch)
          if (tt < tn) {
            if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua[ts + tt], us[ts + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua[ts + tt], us[ts + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua[ts + tt], us[ts + tt]);
            else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ua[ts + tt], us[ts + tt], a.dp);
            else spiral_step(s, coef, slen, t0 + tt, T);
#pragma unroll
            for (int i = 0; i < S; ++i) mine[myC + tt * S + i] = s[i];
          }
        }
        fill = myC + tn * S;
      }
    }
  }
  // epilogue: whatever is left in my window (carry, or carry + partial chunk) goes out as dwords
  if (lane < nvalid && !(a.dbg & 1)) {
    const int done = (T / TS) * TS;              // steps covered by full chunks
    const int first_unflushed = done * S - (done > 0 ? myC : 0);
    const int skip = done > 0 ? 0 : myC;         // no full chunk was flushed: window still has its padding
    float* dst = gout + (long)lane * T * S + first_unflushed;
    for (int i = skip; i < fill; ++i) dst[i - skip] = mine[i];
  }
}

// K3b: the same roll-out with a ROLLED step loop.  The 32-step unroll of rollout_fwd_kernel keeps 64 control
// values in VGPRs (239 VGPRs -> 2 waves per SIMD), which leaves the LDS-read -> store latency of every flush
// exposed.  Here a group of 4 steps is the unit: the group's controls (2 streams x 4 values per row) are staged
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
      else spiral_step(s, coef, slen, t0 + tt, T);
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
    else spiral_step(s, coef, slen, t, T);
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

template <int MODE>
static int launch_mode(const RollArgs& a, hipStream_t s) {
  constexpr int S = ModeTraits<MODE>::S;
  (void)S;
  const long waves = (a.B + kWave - 1) / kWave;
  const long grid = (waves + kRollWaves - 1) / kRollWaves;
  const size_t lds = (size_t)kRollWaves * kWave * 37 * sizeof(float);
  // measured (ST kinematic, T = 50): lean 35.6 vs 39.9 us at B = 32768, 45.7 vs 46.5 at 65536, but 112 vs 70 at
  // 131072 and 234 vs 208 at 262144 (its per-group control fetch over-fetches 2x; the big batches are bound by
  // memory transactions, the small ones by one wave's serial latency) -> lean up to 65536 trajectories
  const char* le = getenv("IRBFN_ROLL_LEAN");
  const int lean = le ? atoi(le) : (a.B <= 65536 ? 1 : 0);
  if (lean)
    hipLaunchKernelGGL(rollout_fwd_lean_kernel<MODE>, dim3((unsigned)grid), dim3(kWave * kRollWaves), lds, s, a);
  else
    hipLaunchKernelGGL(rollout_fwd_kernel<MODE>, dim3((unsigned)grid), dim3(kWave * kRollWaves), lds, s, a);
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
  a.dbg = getenv("IRBFN_ROLL_DBG") ? atoi(getenv("IRBFN_ROLL_DBG")) : 0;
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
  a.dbg = 0;
  a.dp = dp;
  return dispatch_mode(mode, a, s);
}

}  // namespace irbfn
