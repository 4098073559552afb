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
  const float* __restrict__ x0u;   // [B][L]
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

// What makes this kernel fast (each item measured, profiles/r01_rollout_*):
//  * occupancy: one step is a long DEPENDENT chain, so a SIMD needs several resident waves; LDS is
//    kept to one 64 x 33-float tile per wave (8.4 KB), shared by the input and output staging;
//  * straight-line sin/cos/tan (rollout_step.h) instead of ocml's branchy range reduction;
//  * 16-byte accesses: controls 128 contiguous bytes per stream per lane at a time, states flushed
//    through the LDS tile as row runs of TS*S floats.
// Known limit (profiles/r01_rollout_notes.md): the rows are only dword aligned and gfx950 splits
// unaligned 16-byte accesses, so loads, compute and stores add up instead of overlapping
// (compute 67 us + loads 72 us + stores 110 us at B = 262144, T = 50); the next step is a
// sliding-window flush that writes only aligned float4 pieces.
template <int MODE>
__global__ __launch_bounds__(64 * kRollWaves) void rollout_fwd_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int S0 = ModeTraits<MODE>::S0;
  constexpr int TS = kRollTS;
  constexpr int TCH = 32;                        // steps per control chunk (128 bytes per stream per row)
  constexpr int PITCH = 33;                      // >= max(TS*S, TCH), odd: conflict-free row-wise access
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long b0 = ((long)blockIdx.x * kRollWaves + wave) * kWave;
  if (b0 >= a.B) return;                         // whole wave out of range (no block-level barriers used)
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const int L = a.L, T = a.T;
  const long bb = b0 + (lane < nvalid ? lane : nvalid - 1);
  const float* row = a.x0u + bb * L;
  static_assert(TS * S <= 32, "output chunk must fit the tile");
  float* outt = lds + wave * (kWave * PITCH);
  float* myout = outt + lane * PITCH;
  auto wave_sync = [&]() {                       // wave-private LDS: in-order queue, no workgroup barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float* gout = a.states + b0 * (long)T * S;     // tile base in HBM; row stride T*S

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

  // Controls: each lane streams its own row, 128 contiguous bytes per stream at a time (TCH = 32
  // steps = 8 x 16-byte loads per stream, all issued back to back -> one wait per chunk).
  for (int tc = 0; tc < T; tc += TCH) {
    float ua[TCH], us[TCH];
    if constexpr (MODE != IRBFN_ROLLOUT_SPIRAL) {
#pragma unroll
      for (int i4 = 0; i4 < TCH / 4; ++i4) {
        const int t = tc + 4 * i4;
        if (a.dbg & 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { ua[4 * i4 + i] = 0.5f; us[4 * i4 + i] = 0.1f; }
        } else if (t + 3 < T) {                  // u = [a_0..a_{T-1}, sv_0..sv_{T-1}] (dynamics.py:98)
          const f4u va = *reinterpret_cast<const f4u*>(row + S0 + t);
          const f4u vs = *reinterpret_cast<const f4u*>(row + S0 + T + t);
#pragma unroll
          for (int i = 0; i < 4; ++i) { ua[4 * i4 + i] = va[i]; us[4 * i4 + i] = vs[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            ua[4 * i4 + i] = (t + i) < T ? row[S0 + t + i] : 0.0f;
            us[4 * i4 + i] = (t + i) < T ? row[S0 + T + t + i] : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int ts = 0; ts < TCH; ts += TS) {
      const int t0 = tc + ts;
      if (t0 < T) {
        const int tn = (T - t0) < TS ? (T - t0) : TS;
#pragma unroll
        for (int tt = 0; tt < TS; ++tt) {
          if (tt < tn) {
            if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua[ts + tt], us[ts + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, ua[ts + tt], us[ts + tt], a.dp);
            else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, ua[ts + tt], us[ts + tt]);
            else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, ua[ts + tt], us[ts + tt], a.dp);
            else spiral_step(s, coef, slen, t0 + tt, T);
#pragma unroll
            for (int i = 0; i < S; ++i) myout[tt * S + i] = s[i];
          }
        }
        wave_sync();
        // flush: row r, run [t0*S, t0*S + tn*S) -> contiguous tn*S floats at gout + r*T*S + t0*S, as
        // 16-byte pieces (TS*S/4 lanes per row, RPI rows per store instruction).
        constexpr int P4 = (TS * S) / 4;
        constexpr int RPI = kWave / P4;
        if (tn == TS) {
          const int rsub = lane / P4, part = lane - rsub * P4;
          if (rsub < RPI) {
#pragma unroll
            for (int j = 0; j < (kWave + RPI - 1) / RPI; ++j) {
              const int r = j * RPI + rsub;
              if (r < nvalid) {
                const float* src = outt + r * PITCH + 4 * part;
                f4u v;
                v[0] = src[0]; v[1] = src[1]; v[2] = src[2]; v[3] = src[3];
                if (!(a.dbg & 1) || v[0] == 12345.678f)
                  *reinterpret_cast<f4u*>(gout + (long)r * T * S + t0 * S + 4 * part) = v;
              }
            }
          }
        } else {
          const int seg = tn * S;
          for (int idx = lane; idx < nvalid * seg; idx += kWave) {
            const int r = idx / seg, c = idx - r * seg;
            gout[(long)r * T * S + t0 * S + c] = outt[r * PITCH + c];
          }
        }
        wave_sync();
      }
    }
  }
}

template <int MODE>
static int launch_mode(const RollArgs& a, hipStream_t s) {
  constexpr int S = ModeTraits<MODE>::S;
  (void)S;
  const size_t lds = (size_t)kRollWaves * kWave * 33 * sizeof(float);
  const long waves = (a.B + kWave - 1) / kWave;
  const long grid = (waves + kRollWaves - 1) / kRollWaves;
  hipLaunchKernelGGL(rollout_fwd_kernel<MODE>, dim3((unsigned)grid), dim3(kWave * kRollWaves), lds, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int launch_rollout_forward(int mode, const float* x0u, const DynParams& dp, float* states, int64_t B,
                           int T, hipStream_t s) {
  if (B == 0 || T == 0) return IRBFN_OK;
  RollArgs a;
  a.x0u = x0u;
  a.states = states;
  a.B = (long)B;
  a.T = T;
  a.L = rollout_input_dim(mode, T);
  a.dbg = getenv("IRBFN_ROLL_DBG") ? atoi(getenv("IRBFN_ROLL_DBG")) : 0;
  a.dp = dp;
  switch (mode) {
    case IRBFN_ROLLOUT_ST_SELECT: return launch_mode<IRBFN_ROLLOUT_ST_SELECT>(a, s);
    case IRBFN_ROLLOUT_ST_KS: return launch_mode<IRBFN_ROLLOUT_ST_KS>(a, s);
    case IRBFN_ROLLOUT_FULLINT: return launch_mode<IRBFN_ROLLOUT_FULLINT>(a, s);
    case IRBFN_ROLLOUT_FRENET_LS: return launch_mode<IRBFN_ROLLOUT_FRENET_LS>(a, s);
    case IRBFN_ROLLOUT_SPIRAL: return launch_mode<IRBFN_ROLLOUT_SPIRAL>(a, s);
    default: return IRBFN_ERR_BAD_ARG;
  }
}

}  // namespace irbfn
