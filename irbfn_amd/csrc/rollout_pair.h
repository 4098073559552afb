// K3p core: the step loop and the whole-line flush of the two-lanes-per-trajectory roll-out, shared by the stand-alone
// kernel (rollout.hip: rollout_fwd_pair_kernel) and the fused planning tick (plan_tick_wide.hip), so that both produce
// the same bits.  The caller provides the row's state (both lanes of a pair), ONE control stream per lane (even lane:
// acceleration knots, odd lane: steering-rate knots) and a wave-private LDS tile of kPairRows x pair_pitch(S, TS) floats.
#pragma once

#include <stdint.h>

#include "common.h"
#include "rollout_step.h"

namespace irbfn {

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

typedef float f4v __attribute__((ext_vector_type(4)));
#ifndef IRBFN_ROLL_NT
#define IRBFN_ROLL_NT 1            // whole-line stores non-temporal: 107 vs 149 us at B = 262144 (they need no merging in L2 / the
                                   // memory-side cache, and the write-back of a 367 MB stream through it is what limited the rate)
#endif

constexpr int kPairRows = 32;      // trajectories per wave
struct TrigPair {
  int odd;
  static __device__ __forceinline__ float swap(float v) {           // quad_perm [1,0,3,2]: the partner lane's value
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  }
  __device__ __forceinline__ void sincos_tan(float A, float B, bool small, float& sn, float& cs, float& tn) const {
    (void)small;                                 // the range check of sincos_fast covers both angles
    float s1, c1;
    sincos_fast(odd ? B : A, s1, c1);
    const float t1 = fdiv_fast(s1, c1);
    const float s2 = swap(s1), c2 = swap(c1), t2 = swap(t1);
    sn = odd ? s2 : s1;
    cs = odd ? c2 : c1;
    tn = odd ? t1 : t2;
  }
  __device__ __forceinline__ void sincos(float A, float& sn, float& cs) const { sincos_fast(A, sn, cs); }
};

#ifndef IRBFN_PAIR_FLOATS
#define IRBFN_PAIR_FLOATS 70       // floats a chunk of the pair kernel appends to a row window (>= 32): S = 7 -> 10 steps
#endif
constexpr int pair_ts(int S) { return (IRBFN_PAIR_FLOATS + S - 1) / S; }
#ifndef IRBFN_PAIR_MINW
#define IRBFN_PAIR_MINW 4
#endif
constexpr int kPairWaves = 4;      // waves per workgroup
constexpr int pair_pitch(int S, int TS) { return (32 + TS * S) | 1; }      // leftover < 32 + one chunk + the odd lane's spare slot


// s[S]: the state after the prologue; ctl[TCH]: this lane's control knots (slots >= T unused); tile: the wave's LDS tile;
// gout: states of the wave's first trajectory ([nvalid][T][S], rows contiguous); lane: lane id in the wave.
template <int MODE, int TCH, int TS>
__device__ __forceinline__ void pair_rollout_run(float (&s)[ModeTraits<MODE>::S], float (&ctl)[TCH], const DynParams& dp,
                                                 float* tile, float* gout, int T, int nvalid, int lane) {
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int SE = (S + 1) / 2;                // state components the even lane writes; the odd lane writes S - SE
  constexpr int CF = TS * S;
  constexpr int PITCH = pair_pitch(S, TS);
  constexpr int LMAX = ((31 + CF) >> 5) + 1;     // lines a row window can complete in one chunk, + the tail line
  static_assert(MODE != IRBFN_ROLLOUT_SPIRAL, "the spiral has no control knots and no tan: one lane per path");
  static_assert(CF >= 32, "a chunk completes at least one 128-byte line per row");
  const int odd = lane & 1, prow = lane >> 1;
  float* mine = tile + prow * PITCH;
  auto lds_drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  // ---- steps + whole-line flush ------------------------------------------------------------------------------
  const long rs = (long)T * S;
  const int g32 = (int)((reinterpret_cast<uintptr_t>(gout) >> 2) & 31);
  auto head_of = [&](int r) { return (g32 + (int)((r * rs) & 31)) & 31; };
  int fill = head_of(prow);
  const TrigPair trig{odd};
  auto flush = [&](long p0, int nfl, bool last) {
    const int sub = lane >> 3, k = lane & 7;     // 8 lanes per line, 8 rows per instruction
#pragma unroll 1
    for (int j = 0; j < kPairRows / 8; ++j) {
      const int r = j * 8 + sub;
      const int h = head_of(r);
      const int f0 = (int)((h + p0) & 31);       // floats in row r's window before the chunk
      const int tot = f0 + nfl;
      const int nl = tot >> 5;
      const long q0l = (long)h + p0 - f0;        // first line of the window, floats from the window origin
      const float* src = tile + r * PITCH + 4 * k;
      float* dst = gout + r * rs - h + q0l + 4 * k;
      if (r < nvalid) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
          const bool tail = last && l == nl && (tot & 31) > 0;
          if (l < nl || tail) {
            const float v0 = src[32 * l], v1 = src[32 * l + 1], v2 = src[32 * l + 2], v3 = src[32 * l + 3];
            const int lo = (q0l == 0 && l == 0) ? h - 4 * k : 0;       // the row's first line: floats [0, h) are not ours
            const int hi = tail ? (tot & 31) - 4 * k : 4;              // the row's last line: (tot & 31) floats exist
            float* d = dst + 32 * l;
            if (lo <= 0 && hi >= 4) {
#if IRBFN_ROLL_NT
              __builtin_nontemporal_store(f4v{v0, v1, v2, v3}, reinterpret_cast<f4v*>(d));
#else
              *reinterpret_cast<float4*>(d) = float4{v0, v1, v2, v3};
#endif
            } else {
              if (lo <= 0 && hi > 0) d[0] = v0;
              if (lo <= 1 && hi > 1) d[1] = v1;
              if (lo <= 2 && hi > 2) d[2] = v2;
              if (lo <= 3 && hi > 3) d[3] = v3;
            }
          }
        }
      }
    }
  };
  const int nchunks = (T + TS - 1) / TS;
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const int tc = c * TS;
    const int n = (T - tc) < TS ? (T - tc) : TS;
    float* wr = mine + fill + (odd ? SE : 0);
#pragma unroll
    for (int tt = 0; tt < TS; ++tt) {
      if (tt < TCH && tt < n) {                  // wave-uniform
        const int cv = __builtin_bit_cast(int, ctl[tt]);
        const float ua = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(cv, 0xA0, 0xF, 0xF, true));   // even lane's knot
        const float us = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(cv, 0xF5, 0xF, 0xF, true));   // odd lane's knot
        if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua, us, dp, trig);
        else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua, us, dp, trig);
        else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua, us, trig);
        else frenet_step(s, ua, us, dp, trig);
        // even lane: components [0, SE), odd lane: [SE, S) (+ one spare slot that the next step overwrites)
#pragma unroll
        for (int i = 0; i < SE; ++i) {
          float ev = s[i], od = s[(SE + i) < S ? SE + i : S - 1];
          asm volatile("" : "+v"(ev), "+v"(od));             // two plain values: ONE v_cndmask (not an indexed select chain)
          wr[tt * S + i] = odd ? od : ev;
        }
      }
    }
    lds_drain();
    flush((long)tc * S, n * S, tc + n >= T);
    // the floats behind the last complete line move to the window front (each lane moves every other one)
    const int tot = fill + n * S;
    const float* from = mine + (tot & ~31) + odd;
    float keep[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) keep[i] = from[2 * i];
    lds_drain();
    if (tot >= 32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[2 * i + odd] = keep[i];
    }
    fill = tot & 31;
    if constexpr (TCH > TS) {
#pragma unroll
      for (int i = 0; i + TS < TCH; ++i) ctl[i] = ctl[i + TS];
    }
  }
}

}  // namespace irbfn
