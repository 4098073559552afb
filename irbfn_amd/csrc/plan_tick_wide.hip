// The planning tick of BASELINE config 4 in ONE launch: fused RBF forward with wide outputs (O = 2T = 100 control
// knots, rbf_forward_f16_wide.h) -> the block's 32-query x 100-control tiles stay in LDS -> its slice-0 waves integrate
// the trajectories (K3p core, rollout_pair.h) and write the states as whole 128-byte lines.  Replaces the
// forward + sign flip + roll-out launches of launch_forward_rollout (src/irbfn_mpc/irbfn_planner.py:203-210) where
// the instance exists (d = 7, 96 < O <= 112, single-track modes); same bits as the separate launches (the forward's
// arithmetic and geometry, the step functions and the flush are shared code).
// Compiled with the roll-out's flags (the 50-knot register arrays need the full unroll).
#include <stdio.h>

#include "rbf_forward_gram_wide.h"

namespace irbfn {

template <int DC, int BC, int NT, int MODE>
__global__ __launch_bounds__(512) void rbf_tick_f16mfma_wide(const F16Args a, const F16Roll r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  wide_pipe_body<DC, BC, NT, MODE>(a, r, lds);
}

// the same tick on K1g's wide body (rbf_forward_gram_wide.h): the distances on the matrix cores as well
template <int DC, int BC, int NT, int MODE>
__global__ __launch_bounds__(512) void rbf_tick_f16gram_wide(const GramArgs a, const F16Roll r) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  wide_gram_body<DC, BC, NT, MODE>(a, r, lds);
}

template <int BC, int MODE>
static int launch_gtick_inst(const GramArgs& a, const F16Roll& r, int grid, int block, size_t lds, hipStream_t s) {
  auto k = rbf_tick_f16gram_wide<7, BC, 7, MODE>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a, r);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int MODE>
static int launch_gtick_bc(int bc, const GramArgs& a, const F16Roll& r, int grid, int block, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: return launch_gtick_inst<BC_GAUSS, MODE>(a, r, grid, block, lds, s);
    case BC_IQ: return launch_gtick_inst<BC_IQ, MODE>(a, r, grid, block, lds, s);
    case BC_IMQ: return launch_gtick_inst<BC_IMQ, MODE>(a, r, grid, block, lds, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

template <int BC, int MODE>
static int launch_tick_inst(const F16Args& a, const F16Roll& r, int grid, int block, size_t lds, hipStream_t s) {
  auto k = rbf_tick_f16mfma_wide<7, BC, 7, MODE>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a, r);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int MODE>
static int launch_tick_bc(int bc, const F16Args& a, const F16Roll& r, int grid, int block, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: return launch_tick_inst<BC_GAUSS, MODE>(a, r, grid, block, lds, s);
    case BC_IQ: return launch_tick_inst<BC_IQ, MODE>(a, r, grid, block, lds, s);
    case BC_IMQ: return launch_tick_inst<BC_IMQ, MODE>(a, r, grid, block, lds, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

struct TickPlan { int SW, QG; };

// does the one-launch tick exist for this net / horizon / mode / batch (and is it enabled)?
static bool tick_plan(const irbfn_net* net, int mode, int64_t B, int T, TickPlan* p) {
  if (net->opt[IRBFN_OPT_TICK_FUSED] == 0 || B <= 64) return false;
  if (mode != IRBFN_ROLLOUT_ST_SELECT && mode != IRBFN_ROLLOUT_ST_KS) return false;
  if (net->DC != 7 || net->O != 2 * T || T > kTickTch || (net->O + 15) / 16 != 7) return false;
  if (!f16_wide_geometry(net, B, &p->SW, &p->QG)) return false;                 // the forward would not run K1h-wide
  bool pipe;
  f16_wide_normalize(net, &p->SW, &p->QG, &pipe);
  return pipe;
}

bool tick_f16_wide_available(const irbfn_net* net, int mode, int64_t B, int T) {
  TickPlan p;
  return tick_plan(net, mode, B, T, &p);
}

// IRBFN_ERR_UNSUPPORTED: no fused instance for this net / horizon / mode -> the caller takes the separate launches
int launch_tick_f16_wide(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                         const DynParams& dp, float* controls, float* states, int64_t B, int T, hipStream_t s) {
  TickPlan tp;
  if (!tick_plan(net, mode, B, T, &tp)) return IRBFN_ERR_UNSUPPORTED;
  if (net->opt[IRBFN_OPT_FWD_KERNEL] == IRBFN_FWD_AUTO && gram_wide_preferred(net, B)) {
    // K1g's wide body: same conditions, the parameters fit the expansion
    int SW, QG;
    gram_wide_geometry(net, B, &SW, &QG);
    GramArgs a;
    gram_fill_args(net, x, controls, B, SW, QG, &a);
    F16Roll r;
    r.state0 = state0; r.states = states; r.mirror = mirror; r.T = T; r.dp = dp;
    constexpr int S = 7;
    r.wlds = (kPairRows * pair_pitch(S, pair_ts(S)) + 3) & ~3;
    size_t lds = gram_wide_lds_bytes(net, SW, QG, (size_t)QG * 32 * (net->O | 1));
    const size_t out = (size_t)QG * r.wlds * sizeof(float);
    lds = lds > out ? lds : out;
    if (lds <= 160 * 1024) {
      const long groups = (B + 31) / 32;
      const int grid = (int)((groups + QG - 1) / QG);
      const int rc = mode == IRBFN_ROLLOUT_ST_KS ? launch_gtick_bc<IRBFN_ROLLOUT_ST_KS>(net->bclass, a, r, grid, SW * QG * 64, lds, s)
                                                 : launch_gtick_bc<IRBFN_ROLLOUT_ST_SELECT>(net->bclass, a, r, grid, SW * QG * 64, lds, s);
      if (rc == IRBFN_OK) {
        snprintf(net->last_name, sizeof(net->last_name), "rbf_tick_f16gram_wide<D=7,BC=%d,NT=7,MODE=%d,SW=%d,QG=%d>", net->bclass, mode, SW, QG);
        net->last_grid = grid;
        net->last_block = SW * QG * 64;
      }
      return rc;
    }
  }
  const int SW = tp.SW, QG = tp.QG;
  const int NT = 7;
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  F16Args a;
  a.x = x; a.img = net->f16_img; a.oscale = net->f16_oscale; a.bias = net->bias; a.out = controls; a.gate = net->gate();
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.nchunks = nchunks; a.S = SW; a.QG = QG;
  F16Roll r;
  r.state0 = state0; r.states = states; r.mirror = mirror; r.T = T; r.dp = dp;
  constexpr int S = 7;
  r.wlds = (kPairRows * pair_pitch(S, pair_ts(S)) + 3) & ~3;
  const size_t ring = (size_t)SW * kWideRing * f16_chunk_bytes(net->DC, NT);
  const size_t red = ((size_t)SW * QG * 2 * 4 * 64 + (size_t)QG * 32 + (size_t)QG * 32 * (net->O | 1)) * sizeof(float);
  const size_t out = (size_t)QG * r.wlds * sizeof(float);
  size_t lds = ring > red ? ring : red;
  lds = lds > out ? lds : out;
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  const int rc = mode == IRBFN_ROLLOUT_ST_KS ? launch_tick_bc<IRBFN_ROLLOUT_ST_KS>(net->bclass, a, r, grid, SW * QG * 64, lds, s)
                                             : launch_tick_bc<IRBFN_ROLLOUT_ST_SELECT>(net->bclass, a, r, grid, SW * QG * 64, lds, s);
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_tick_f16mfma_wide<D=7,BC=%d,NT=7,MODE=%d,SW=%d,QG=%d>", net->bclass, mode, SW, QG);
    net->last_grid = grid;
    net->last_block = SW * QG * 64;
  }
  return rc;
}

}  // namespace irbfn
