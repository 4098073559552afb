// K3: stand-alone roll-out kernels (forward).  One lane integrates one trajectory, the state lives
// in VGPRs for all T steps.  The kernels are HBM-bound (1.8 KB of traffic per trajectory at T = 50
// against ~10 kFLOP): a wave's 64 input rows form ONE contiguous chunk of HBM, so it is copied into
// LDS with coalesced 16-byte loads and read back row-wise; the states are staged per time-chunk in
// LDS and flushed as contiguous row segments.
//
// Replaces integrate_st_mult (src/irbfn_mpc/dynamics.py:94-100), dynamic_st_onestep_aux (:103-187),
// integrate_frenet_mult (:284-290), the inline bicycle of scripts/train_nmpc.py:329-374 and
// integrate_path_mult (src/irbfn_mpc/planner_utils.py:62-77).
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
  int T, L, TS;                    // TS = steps staged per LDS flush
  int stage_in;                    // 1: input tile staged in LDS
  DynParams dp;
};

// LDS: [in tile: 64*L floats (if stage_in)] [out tile: 64 * (TS*S + 1) floats]
template <int MODE>
__global__ __launch_bounds__(64) void rollout_fwd_kernel(const RollArgs a) {
  extern __shared__ float lds[];
  constexpr int S = ModeTraits<MODE>::S;
  constexpr int S0 = ModeTraits<MODE>::S0;
  const int lane = threadIdx.x;
  const long b0 = (long)blockIdx.x * kWave;
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const int L = a.L, T = a.T;

  const float* row;                              // this lane's input row (LDS or HBM)
  float* outt;
  if (a.stage_in) {
    const float* src = a.x0u + b0 * L;
    const int total = nvalid * L;
    // the tile base b0*L*4 bytes is 16-byte aligned (b0 is a multiple of 64)
    const int n4 = total >> 2;
    const float4* src4 = reinterpret_cast<const float4*>(src);
    float4* dst4 = reinterpret_cast<float4*>(lds);
    for (int i = lane; i < n4; i += kWave) dst4[i] = src4[i];
    for (int i = (n4 << 2) + lane; i < total; i += kWave) lds[i] = src[i];
    __syncthreads();                             // one wave per workgroup: cheap
    const int rr = lane < nvalid ? lane : nvalid - 1;
    row = lds + rr * L;
    outt = lds + ((kWave * L + 3) & ~3);
  } else {
    const long bb = (b0 + lane) < a.B ? (b0 + lane) : a.B - 1;
    row = a.x0u + bb * L;
    outt = lds;
  }
  const int TSS = a.TS * S;                      // floats per row per flush
  const int pitch = TSS | 1;                     // odd pitch: conflict-free row-wise writes
  float* myout = outt + lane * pitch;
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

  for (int t0 = 0; t0 < T; t0 += a.TS) {
    const int tn = (T - t0) < a.TS ? (T - t0) : a.TS;
    for (int tt = 0; tt < tn; ++tt) {
      const int t = t0 + tt;
      if constexpr (MODE == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, row[S0 + t], row[S0 + T + t], a.dp);
      else if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(s, row[S0 + t], row[S0 + T + t], a.dp);
      else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(s, row[S0 + t], row[S0 + T + t]);
      else if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) frenet_step(s, row[S0 + t], row[S0 + T + t], a.dp);
      else spiral_step(s, coef, slen, t, T);
#pragma unroll
      for (int i = 0; i < S; ++i) myout[tt * S + i] = s[i];
    }
    __syncthreads();
    // flush: row r, segment [t0*S, t0*S + tn*S) -> contiguous tn*S floats at gout + r*T*S + t0*S
    const int seg = tn * S;
    for (int idx = lane; idx < nvalid * seg; idx += kWave) {
      const int r = idx / seg, c = idx - r * seg;
      gout[(long)r * T * S + t0 * S + c] = outt[r * pitch + c];
    }
    __syncthreads();
  }
}

template <int MODE>
static int launch_mode(const RollArgs& a0, hipStream_t s) {
  RollArgs a = a0;
  constexpr int S = ModeTraits<MODE>::S;
  // stage the input tile if it fits a modest LDS budget; chunk the output to <= ~16 KB per wave
  a.stage_in = ((size_t)kWave * a.L * 4 <= 40 * 1024) ? 1 : 0;
  int TS = (16 * 1024) / (kWave * S * 4);
  if (TS < 1) TS = 1;
  if (TS > a.T) TS = a.T;
  a.TS = TS;
  const size_t in_f = a.stage_in ? (((size_t)kWave * a.L + 3) & ~(size_t)3) : 0;
  const size_t lds = (in_f + (size_t)kWave * ((TS * S) | 1)) * sizeof(float);
  auto kern = rollout_fwd_kernel<MODE>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  }
  const long grid = (a.B + kWave - 1) / kWave;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kWave), lds, s, a);
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
  a.TS = 1;
  a.stage_in = 0;
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
