// The epilogue of the narrow matrix-core forward kernels (K1h rbf_fwd_f16mfma / rbf_tick_f16mfma, K1g rbf_fwd_f16gram /
// rbf_tick_f16gram): the S centre slices of a query group summed in fixed order through LDS, gate, scale and bias
// (src/irbfn_mpc/model.py:193-196), store; with ROLL the slice-0 wave of a query group integrates the trajectories of its 32
// rows itself (the planning tick of the narrow nets in one launch: irbfn_planner.py:203-212).
#pragma once

#include "rbf_forward_f16_wide.h"

namespace irbfn {

constexpr int kTickNarrowT = 8;       // horizons of the narrow tick: O = 2T <= 16
constexpr int kTickNarrowCP = 17;     // controls tile pitch (floats), odd
constexpr int kTickNarrowSP = 65;     // states staging pitch: T * S <= 64 floats per row, odd

// acc[t][r]: A1 + 2^-11 A2 of query tile t, D layout (row 4 (lane >> 4) + r); gam[t]: the gate of query (lane & 15) of tile t
// (slice-0 waves); inv_scale: 1 / (scale of the basis values x 2^15); every wave of the block arrives here
template <bool ROLL>
__device__ __forceinline__ void narrow_epilogue(const F16Args& a, const F16Roll& rl, int mode, unsigned char* lds, const f4_t (&acc)[2],
                                                const float (&gam)[2], int S, int slice, int qg, long q0, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, n = lane & 15;
  // ---- sum the S centre slices in fixed order, scale, bias, store
  __syncthreads();                                           // every wave is done with its ring
  float* red = reinterpret_cast<float*>(lds);                // [QG][S][2][4][64]
  float* gl = red + (size_t)a.QG * S * 2 * 4 * 64;           // [QG][32]
  [[maybe_unused]] float* ctile = gl + a.QG * 32;            // ROLL: [QG * 32][kTickNarrowCP] controls, then the states staging tiles
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(((qg * S + slice) * 2 + t) * 4 + r) * 64 + lane] = acc[t][r];
  if (slice == 0 && g == 0) { gl[qg * 32 + n] = gam[0]; gl[qg * 32 + 16 + n] = gam[1]; }
  __syncthreads();
  if (slice == 0 && n < a.O) {
    const float sc = a.oscale[n] * inv_scale;                                   // s_o * 2^-29 (K1h)
    const float bi = a.bias[n];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
        for (int s2 = 0; s2 < S; ++s2) v += red[(((qg * S + s2) * 2 + t) * 4 + r) * 64 + lane];
        const int row = t * 16 + 4 * g + r;                  // D layout: row = 4 (lane >> 4) + reg
        const long q = q0 + row;
        float y = __builtin_fmaf(gl[qg * 32 + row] * v, sc, bi);                             // model.py:193-196
        if constexpr (ROLL) {
          if (rl.mirror != nullptr && n >= rl.T && q < a.B && rl.mirror[q] != 0) y = -y;     // irbfn_planner.py:203-204
          ctile[(qg * 32 + row) * kTickNarrowCP + n] = y;
        }
        if (q < a.B && a.out != nullptr) a.out[q * a.O + n] = y;
      }
  }
  if constexpr (ROLL) {
    // ---- the wave that produced the 32 rows rolls them out: lane l < 32 integrates row l (the step functions of the
    // stand-alone kernels, rollout_step.h: same bits), the T x S states of the wave's rows -- one contiguous block of
    // HBM -- are staged row by row in LDS and leave as coalesced dwords
    if (slice != 0) return;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int T = rl.T;
    const long left = a.B - q0;
    const int nvalid = left < 32 ? (left > 0 ? (int)left : 0) : 32;
    const int S = (mode == IRBFN_ROLLOUT_FULLINT) ? 5 : (mode == IRBFN_ROLLOUT_FRENET_LS ? 8 : 7);
    float u[2 * kTickNarrowT];
    if (lane < 32) {
      const float* ur = ctile + (qg * 32 + lane) * kTickNarrowCP;
#pragma unroll
      for (int i = 0; i < 2 * kTickNarrowT; ++i) u[i] = ur[i];                               // slots >= O hold stale LDS: unused
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                                                         // the controls tile becomes the staging tile
    float* stage = ctile + (size_t)a.QG * 32 * kTickNarrowCP + (size_t)qg * 32 * kTickNarrowSP;
    if (lane < nvalid) {
      const long b = q0 + lane;
      float* o = stage + lane * kTickNarrowSP;
      auto ctl = [&](int i) {                                                                // u[i], i wave-uniform: static register index
        float v = u[0];
#pragma unroll
        for (int k = 1; k < 2 * kTickNarrowT; ++k) v = (i == k) ? u[k] : v;
        return v;
      };
      if (mode == IRBFN_ROLLOUT_ST_SELECT || mode == IRBFN_ROLLOUT_ST_KS) {
        float st[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) st[i] = rl.state0[b * 7 + i];
        for (int t = 0; t < T; ++t) {
          if (mode == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(st, ctl(t), ctl(T + t), rl.dp);
          else st_step<false>(st, ctl(t), ctl(T + t), rl.dp);
#pragma unroll
          for (int i = 0; i < 7; ++i) o[t * 7 + i] = st[i];
        }
      } else if (mode == IRBFN_ROLLOUT_FRENET_LS) {
        float st[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = rl.state0[b * 8 + i];
        for (int t = 0; t < T; ++t) {
          frenet_step(st, ctl(t), ctl(T + t), rl.dp);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[t * 8 + i] = st[i];
        }
      } else {
        float st[5] = {0.0f, 0.0f, 0.0f, clipf(rl.state0[b], 0.0f, 7.0f), 0.0f};             // train_nmpc.py:319
        for (int t = 0; t < T; ++t) {
          fullint_step(st, ctl(t), ctl(T + t));
#pragma unroll
          for (int i = 0; i < 5; ++i) o[t * 5 + i] = st[i];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int rowf = T * S;                                                                  // floats per trajectory (<= 64)
    float* gout = rl.states + q0 * (long)rowf;
    for (int idx = lane; idx < nvalid * rowf; idx += 64) {
      const int r = idx / rowf, c = idx - r * rowf;
      gout[idx] = stage[r * kTickNarrowSP + c];
    }
  }
}

}  // namespace irbfn
