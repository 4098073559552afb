// K0 (pack), gate kernel and the K1 dispatcher -- see rbf_forward.h for the design of K1; the K1
// instantiations live in rbf_forward_kernels.hip (one object per compiled D).
#include <math.h>
#include "rbf_forward.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace irbfn {

int launch_forward_d3(const FwdArgs&, int, int, int, bool, bool, int, size_t, hipStream_t, int*);
int launch_forward_d4(const FwdArgs&, int, int, int, bool, bool, int, size_t, hipStream_t, int*);
int launch_forward_d7(const FwdArgs&, int, int, int, bool, bool, int, size_t, hipStream_t, int*);
int launch_forward_d8(const FwdArgs&, int, int, int, bool, bool, int, size_t, hipStream_t, int*);

static const int kCompiledOP[] = {2, 4, 5, 8, 10, 16, 32, 64, 100, 128};

int padded_O(int O) {
  for (int op : kCompiledOP)
    if (O <= op) return op;
  return -1;
}

int padded_D(int D) {
  if (D <= 3) return 3;
  if (D == 4) return 4;
  if (D <= 7) return 7;
  if (D == 8) return 8;
  return -1;
}

// ------------------------------------------------------------------------------------------------
// gate kernel: _region_activation (model.py:42-95) -> gamma[B][R].  One wave per 64 queries; the
// per-dimension factors are tabulated in LDS once, then every region is a product of nsplit
// look-ups.  Used by irbfn_net_gate and as the pre-pass of the VJP.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gate_kernel(const float* __restrict__ x, float* __restrict__ gamma,
                                                  GateTables gt, long B, int D, int R) {
  extern __shared__ float gtab[];                 // [E][64]
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  const long bb = b < B ? b : B - 1;
  const int E = gt.nsplit * gt.max_ranges;
  for (int e = 0; e < E; ++e) {
    const int d = e / gt.max_ranges;
    gtab[e * kWave + lane] = gate_factor(x[bb * D + d], gt.lo[e], gt.hi[e], gt.delta[d]);
  }
  // each lane only reads its own column: no barrier needed
  if (b >= B) return;
  for (int r = 0; r < R; ++r) {
    float g = 0.0f;
    if (r < gt.n_ranges) {
      g = 1.0f;
      for (int d = 0; d < gt.nsplit; ++d)
        g *= gtab[(d * gt.max_ranges + gt.dim_ranges[r * gt.nsplit + d]) * kWave + lane];
    }
    gamma[b * R + r] = g;
  }
}

int launch_gate(irbfn_net* net, const float* x, float* gamma, int64_t B, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  const size_t lds = (size_t)net->nsplit * net->max_ranges * kWave * sizeof(float);
  if (lds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long grid = (B + kWave - 1) / kWave;
  hipLaunchKernelGGL(gate_kernel, dim3((unsigned)grid), dim3(kWave), lds, s, x, gamma, net->gate(), (long)B,
                     net->D, net->R);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// ------------------------------------------------------------------------------------------------
// K1 dispatcher: picks queries-per-lane Q and waves-per-workgroup NW, sizes LDS, launches.
// ------------------------------------------------------------------------------------------------
// Kernel selection and launch geometry are decided here from the descriptor's shape and its options
// (irbfn_net_set_option, include/irbfn_hip.h); nothing on the launch path reads the environment.
static int opt_or(const irbfn_net* net, int key, int dflt) {
  const int v = net->opt[key];
  return v > 0 ? v : dflt;           // geometry options: 0 = automatic
}

static int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

static int run_forward(irbfn_net* net, FwdArgs& a, bool roll, hipStream_t s) {
  const int OP = net->OP;
  const bool gated = net->R > 1 || a.gamma_ext != nullptr;
  // --- Q: two queries per lane halve the scalar-stream traffic per pair, but the kernel is VALU-issue
  // bound and more resident waves hide the scalar-load latency better (measured: Q=1,NW=16 152 us vs
  // Q=2,NW=16 175 us at cfg-2), so Q = 2 only once Q = 1 alone over-subscribes the chip.
  int Q = 1;
  const bool q2_compiled = (OP == 2 || OP == 5 || OP == 10);
  if (q2_compiled && a.B >= (long)kWave * 32768) Q = 2;
  Q = opt_or(net, IRBFN_OPT_FWD_Q, Q);
  if (Q != 1 && !(Q == 2 && q2_compiled)) Q = 1;
  const int ROWS = kWave * Q;
  const long tiles = (a.B + ROWS - 1) / ROWS;
  // --- NW: enough waves to cover the chip (1024 SIMDs) several times, >= 32 centres per wave
  const int max_threads = (OP * Q > 48) ? 512 : 1024;
  long want = (16384 + tiles - 1) / tiles;
  int nw = want < 1 ? 1 : (want > 16 ? 16 : (int)want);
  nw = pow2_floor(nw);
  while (nw > 1 && net->N / nw < 32) nw /= 2;
  // small nets (the reference's trained ones have 1000-1280 centres): fewer, longer waves once the launch still has
  // 8 waves per SIMD -- the per-block prologue / 16-wave reduction costs as much as 80 centres per wave
  // (128-region net, B = 65536: 95 -> 75 us; 12-region Frenet net 64 -> 49 us)
  while (nw > 1 && net->N / nw < 128 && tiles * (nw / 2) >= 8192) nw /= 2;
  while (nw * kWave > max_threads) nw /= 2;
  nw = opt_or(net, IRBFN_OPT_FWD_NW, nw);
  if (nw * kWave > max_threads) nw = max_threads / kWave;
  // --- LDS: max(stage, gate table) aliased with the reduction buffers
  const int OC = OP < 16 ? OP : 16;
  size_t stage = (size_t)ROWS * net->D;
  if (gated) stage += (size_t)net->nsplit * net->max_ranges * ROWS;
  size_t red = (size_t)nw * Q * OC * (kWave + 1) + ROWS;
  const size_t roll_floats = roll ? (size_t)ROWS * (net->O + 65) : 0;      // controls + the states staging tile (pitch 65)
  red += roll_floats;
  size_t lds = (stage > red ? stage : red) * sizeof(float);
  while (lds > 160 * 1024 && nw > 1) {
    nw /= 2;
    red = (size_t)nw * Q * OC * (kWave + 1) + ROWS + roll_floats;
    lds = (stage > red ? stage : red) * sizeof(float);
  }
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;

  int grid = 0, rc;
  switch (net->DC) {
    case 3: rc = launch_forward_d3(a, OP, Q, net->bclass, gated, roll, nw, lds, s, &grid); break;
    case 4: rc = launch_forward_d4(a, OP, Q, net->bclass, gated, roll, nw, lds, s, &grid); break;
    case 7: rc = launch_forward_d7(a, OP, Q, net->bclass, gated, roll, nw, lds, s, &grid); break;
    case 8: rc = launch_forward_d8(a, OP, Q, net->bclass, gated, roll, nw, lds, s, &grid); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_qlane<D=%d,OP=%d,Q=%d,BC=%d,GATED=%d,ROLL=%d>",
             net->DC, OP, Q, net->bclass, (int)gated, (int)roll);
    net->last_grid = grid;
    net->last_block = nw * kWave;
  }
  return rc;
}

static void fill_args(irbfn_net* net, FwdArgs& a, const float* x, float* out, int64_t B) {
  memset(&a, 0, sizeof(a));
  a.x = x;
  a.rec = net->rec;
  a.bias = net->bias;
  a.out = out;
  a.gate = net->gate();
  a.B = (long)B;
  a.Dreal = net->D;
  a.O = net->O;
  a.N = net->N;
  a.K = net->K;
  a.S = net->S;
  a.basis = net->basis;
  a.R = net->R;
}

// K1m (Phi x W on the f32 matrix cores).  Opt-in (IRBFN_FWD_MFMA=1): measured on MI355X at cfg-2 it is
// SLOWER than K1 (169-217 us vs 152 us): the f32-input MFMA runs at the fp32 vector rate and does not
// overlap with the VALU distance/basis work, so with O = 10 padded to a 16-wide tile it buys nothing.
// Kept as the "reduction expressed as a dense GEMM" variant that BASELINE config 5 asks to report.
// For WIDE outputs (O > 16: e.g. 50-step control sequences, O = 100) the picture flips: the weight FMAs
// dominate and the MFMA issues them ~1.8x more densely than SGPR-operand v_fmac (cfg-4 forward 342 ->
// 301 us), so K1m is preferred over K1 there (behind K1h where that is eligible).
// IRBFN_OPT_FWD_KERNEL = IRBFN_FWD_K1 / IRBFN_FWD_K1M forces K1 / K1m.
bool prefer_mfma(const irbfn_net* net) {
  const int e = net->opt[IRBFN_OPT_FWD_KERNEL];
  if (!net->recm || e == IRBFN_FWD_K1) return false;
  return e == IRBFN_FWD_K1M || net->O > 16;
}

static int try_forward_mfma(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s) {
  if (!prefer_mfma(net)) return IRBFN_ERR_UNSUPPORTED;
  const bool wide = net->O > 16;
  int QJ = opt_or(net, IRBFN_OPT_FWD_QJ, wide ? 2 : 4);
  if (QJ != 1 && QJ != 2 && QJ != 4) QJ = wide ? 2 : 4;
  if (wide && QJ == 4) QJ = 2;                   // QJ = 4 is compiled for NT <= 4 only
  const long tiles = (B + 16 * QJ - 1) / (16 * QJ);
  long want = ((wide ? 2048 : 8192) + tiles - 1) / tiles;
  int nw = want < 1 ? 1 : (want > 16 ? 16 : (int)want);
  nw = pow2_floor(nw);
  const int chunks = net->Npad / 16;
  while (nw > 1 && chunks / nw < 2) nw /= 2;
  nw = opt_or(net, IRBFN_OPT_FWD_NW, nw);
  if (nw > 16) nw = 16;
  return launch_forward_mfma(net, x, out, B, QJ, nw, s);
}

// ClusterWCRBFNet gate (model.py:402-404): logits = x Wc + bc, gamma = softmax(logits) (max-subtracted, as
// jax.nn.softmax).  One thread per query, two passes over the R regions; logits are part of the model's output.
__global__ __launch_bounds__(256) void cluster_gate_kernel(const float* __restrict__ x, const float* __restrict__ wc,
                                                           const float* __restrict__ bc, float* __restrict__ logits,
                                                           float* __restrict__ gamma, long B, int D, int R) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float xv[16];
  for (int d = 0; d < D; ++d) xv[d] = x[b * D + d];
  float mx = -INFINITY;
  for (int r = 0; r < R; ++r) {
    float l = bc[r];
    for (int d = 0; d < D; ++d) l = __builtin_fmaf(xv[d], wc[d * R + r], l);
    logits[b * R + r] = l;
    mx = fmaxf(mx, l);
  }
  float sum = 0.0f;
  for (int r = 0; r < R; ++r) {
    const float e = expf(logits[b * R + r] - mx);
    gamma[b * R + r] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  for (int r = 0; r < R; ++r) gamma[b * R + r] *= inv;
}

int launch_cluster_gate(const float* x, const float* wc, const float* bc, float* logits, float* gamma, int64_t B, int D,
                        int R, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  if (D < 1 || D > 16 || R < 1) return IRBFN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(cluster_gate_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, x, wc, bc, logits, gamma,
                     (long)B, D, R);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// forward with caller-provided region weights gamma[B][R] (ClusterWCRBFNet): always the gated K1
int launch_forward_gamma(irbfn_net* net, const float* x, const float* gamma, float* out, int64_t B, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  FwdArgs a;
  fill_args(net, a, x, out, B);
  a.gamma_ext = gamma;
  return run_forward(net, a, false, s);
}

// K1h (Phi x W on the f16 matrix cores at float32 accuracy, rbf_forward_f16.hip): one region, fast basis,
// O <= 128.  The default where eligible (cfg-2 131 vs 142 us, cfg-5 7.5 vs 8.9 ms against K1);
// IRBFN_OPT_FWD_KERNEL forces another kernel, IRBFN_OPT_FWD_F16_TERMS = 1 selects the reduced-precision
// single-product variant (reporting only; never reachable without that explicit option).  Geometry: S centre
// slices x QG query groups of 32 per 8-wave block, S chosen so that the launch has >= 16384 waves.
// wide outputs (16 < O <= 128): would the automatic dispatch run K1h, and with which block geometry?  Block-shared W
// stream; SW = centre slices per block so that the grid covers the 256 CUs.  Shared with the fused planning tick
// (plan_tick_wide.hip), which must reduce the slices in the same order as the plain forward.
bool f16_wide_geometry(const irbfn_net* net, int64_t B, int* SW_out, int* QG_out) {
  const int e = net->opt[IRBFN_OPT_FWD_KERNEL];
  if (e != IRBFN_FWD_AUTO && e != IRBFN_FWD_K1H) return false;
  if (!net->f16_img || !f16_eligible(net) || net->O <= 16) return false;
  if (B < opt_or(net, IRBFN_OPT_FWD_F16_MINB, 65)) return false;
  const long groups = (B + 31) / 32;
  int SW = 1;
  while (SW < 4 && (groups * SW + 7) / 8 < 256) SW *= 2;
  SW = opt_or(net, IRBFN_OPT_FWD_F16_S, SW);
  if (SW != 1 && SW != 2 && SW != 4) SW = 1;
  *SW_out = SW;
  *QG_out = opt_or(net, IRBFN_OPT_FWD_F16_QG, 8 / SW);
  return true;
}

// narrow outputs (O <= 16): would the automatic dispatch run K1h, and with which block geometry?  S centre slices x QG
// query groups of 32 per 8-wave block, S chosen so that the launch has >= 16384 waves.  Shared with the one-launch tick.
bool f16_narrow_geometry(const irbfn_net* net, int64_t B, int* S_out, int* QG_out) {
  const int e = net->opt[IRBFN_OPT_FWD_KERNEL];
  if (e != IRBFN_FWD_AUTO && e != IRBFN_FWD_K1H) return false;
  if (!net->f16_img || !f16_eligible(net) || net->O > 16) return false;
  if (B < opt_or(net, IRBFN_OPT_FWD_F16_MINB, 65)) return false;
  const long groups = (B + 31) / 32;
  long want = (16384 + groups - 1) / groups;           // measured at cfg-2: S = 8 (16384 waves) 131 us, S = 4 134 us
  int S = want < 1 ? 1 : (want > 8 ? 8 : (int)want);
  S = pow2_floor(S);
  if (S < want && S < 8) S *= 2;
  const int nchunks = (net->N + 31) / 32;
  while (S > 1 && nchunks / S < 2) S /= 2;
  // small nets: at least 8 chunks per wave once the launch still has 4 waves per SIMD (the slice reduction and the
  // prologue cost as much as a few chunks; N = 1000: 47.5 -> 44.5 us, N = 256: 23.6 -> 19.1 us at B = 65536)
  while (S > 1 && nchunks / S < 8 && groups * (S / 2) >= 4096) S /= 2;
  S = opt_or(net, IRBFN_OPT_FWD_F16_S, S);
  if (S > 8 || S > nchunks) S = 1;
  int QG = opt_or(net, IRBFN_OPT_FWD_F16_QG, 8 / S);
  if (S * QG > 8) QG = 1;
  *S_out = S; *QG_out = QG;
  return true;
}

// K1g (rbf_forward_gram.hip): where K1h's narrow kernel would run with its default operand pairs and the parameters fit the
// expansion (gram_ok, set by the pack)
bool gram_preferred(const irbfn_net* net, int64_t B) {
  const int ot = net->opt[IRBFN_OPT_FWD_F16_TERMS];
  // below ~12k queries K1h's eight centre slices per query group finish sooner (config-2 net: B = 8192 27 vs 35 us,
  // B = 16384 44 vs 37 us)
  return net->gram_img && net->gram_ok && net->O <= 16 && (ot == 3 || ot == 0) && B >= 12288;
}

// K1g for wide outputs (rbf_forward_gram_wide.hip): d = 7 or 8, the parameters fit the expansion
bool gram_wide_preferred(const irbfn_net* net, int64_t B) {
  return net->gram_img && net->gram_ok && net->O > 16 && net->O <= 128 && (net->DC == 7 || net->DC == 8) && B >= 2048 && net->N >= 256;
}

// S centre slices x QG query groups of 32 per block; the QG waves of a slice share one stream of chunk images (a ring of five,
// 35 KiB of LDS per slice).  Measured at the config-2 net (profiles/r03_gram_batch_sweep.txt; us at B = 16384 / 24576 / 32768 /
// 65536 / 131072 / 262144): S = 4, QG = 2: 38 / 70 / 72 / 139 / 275 / 547; S = 2, QG = 4: 53 / 53 / 55 / 82 / 155 / 298; S = 1, QG = 8:
// 88 / 88 / 88 / 88 / 142 / 273 -- more slices while the launch is short of waves, never more blocks than are resident at once.
// Between one and one and a half rounds of the S = 2 form (2048 < groups <= 3072: the reference's training batch of 80000) a
// second, thin round of blocks costs a wave's whole latency-bound pass over its slice; blocks of FOUR waves with all the centres
// (S = 1, QG = 4: four resident per CU, 4096 groups at once) spread the same work over the chip in one round
// (profiles/r03_gram_geometry_sweep.txt: N = 1000, B = 80000: 40.7 vs 47.4 us; N = 2048: 67 vs 73; N = 4096: 121 vs 122).
void gram_geometry(const irbfn_net* net, int64_t B, int* S_out, int* QG_out) {
  const long groups = (B + 31) / 32;
  const int nchunks = (net->N + 31) / 32;
  int S = groups <= 512 ? 4 : (groups <= 2048 ? 2 : 1);
  while (S > 1 && nchunks / (2 * S) < 4) S /= 2;            // at least 8 chunks per wave
  S = opt_or(net, IRBFN_OPT_FWD_F16_S, S);
  if (S > 7 || S > nchunks) S = 1;
  int QG = opt_or(net, IRBFN_OPT_FWD_F16_QG, (S == 1 && groups > 2048 && groups <= 3072) ? 4 : 8 / S);
  if (S * QG > 16) QG = 1;
  *S_out = S; *QG_out = QG;
}

static int try_forward_f16(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s) {
  const int e = net->opt[IRBFN_OPT_FWD_KERNEL];
  if (e != IRBFN_FWD_AUTO && e != IRBFN_FWD_K1H) return IRBFN_ERR_UNSUPPORTED;
  if (!net->f16_img) return IRBFN_ERR_UNSUPPORTED;
  if (B < opt_or(net, IRBFN_OPT_FWD_F16_MINB, 65)) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  if (net->O > 16) {
    int SW, QGw;
    if (!f16_wide_geometry(net, B, &SW, &QGw)) return IRBFN_ERR_UNSUPPORTED;
    if (e == IRBFN_FWD_AUTO && gram_wide_preferred(net, B)) {  // K1g's wide form: the distances on the matrix cores as well
      const int rc = launch_forward_gram(net, x, out, B, 1, 1, s);
      if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
    }
    return launch_forward_f16(net, x, out, B, SW, QGw, 3, s);
  }
  int S, QG;
  if (!f16_narrow_geometry(net, B, &S, &QG)) return IRBFN_ERR_UNSUPPORTED;
  if (e == IRBFN_FWD_AUTO && gram_preferred(net, B)) {         // K1g: the distances on the matrix cores as well
    int Sg, QGg;
    gram_geometry(net, B, &Sg, &QGg);
    const int rc = launch_forward_gram(net, x, out, B, Sg, QGg, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  }
  const int ot = net->opt[IRBFN_OPT_FWD_F16_TERMS];
  const int terms = (ot == 1 || ot == 2) ? ot : 3;           // 1: plain f16, 2: plain bf16 (both reporting only), 3: pairs
  return launch_forward_f16(net, x, out, B, S, QG, terms, s);
}

int launch_forward(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  // K1s: small batches (planner ticks) -> centre-lane latency kernel (IRBFN_OPT_FWD_SMALL = 0 disables it)
  const int forced = net->opt[IRBFN_OPT_FWD_KERNEL];
  if (forced == IRBFN_FWD_AUTO && small_eligible(net, B) && net->opt[IRBFN_OPT_FWD_SMALL] != 0) {
    const int rc = launch_forward_small(net, x, out, B, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  }
  // K1r: several regions with a sparse gate -> every query visits only its non-zero regions (rbf_sparse.hip)
  if (forced == IRBFN_FWD_K1R || (forced == IRBFN_FWD_AUTO && sparse_preferred(net, B))) {
    const int rc = launch_forward_sparse(net, x, out, B, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, s);
    if (rc != IRBFN_ERR_UNSUPPORTED || forced == IRBFN_FWD_K1R) return rc;
  }
  // a kernel that is "not eligible" answers IRBFN_ERR_UNSUPPORTED; every other status (a HIP launch failure
  // of the preferred kernel in particular) is returned, never papered over by the next kernel in line
  if (forced == IRBFN_FWD_K1G) {
    if (!net->gram_img || !net->gram_ok) return IRBFN_ERR_UNSUPPORTED;
    int Sg, QGg;
    gram_geometry(net, B, &Sg, &QGg);
    return launch_forward_gram(net, x, out, B, Sg, QGg, s);
  }
  int rc = try_forward_f16(net, x, out, B, s);
  if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  if (forced == IRBFN_FWD_K1H) return IRBFN_ERR_UNSUPPORTED;
  rc = try_forward_mfma(net, x, out, B, s);
  if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  if (forced == IRBFN_FWD_K1M) return IRBFN_ERR_UNSUPPORTED;
  FwdArgs a;
  fill_args(net, a, x, out, B);
  return run_forward(net, a, false, s);
}

// the tick runs forward -> roll-out through a controls buffer (when the caller provides one)
bool tick_through_controls(const irbfn_net* net, int64_t B) {
  int S, QG;
  return B > 64 && (prefer_mfma(net) || f16_narrow_geometry(net, B, &S, &QG));
}

int launch_forward_rollout(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                           const DynParams& dp, float* controls, float* states, int64_t B, int T,
                           hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  if (states == nullptr) {                       // controls only: any forward kernel, then the sign flip
    if (!controls) return IRBFN_ERR_BAD_ARG;
    int rc = launch_forward(net, x, controls, B, s);
    if (rc != IRBFN_OK || !mirror) return rc;
    return launch_unmirror(controls, mirror, B, net->O, net->O / 2, s);
  }
  if (mode != IRBFN_ROLLOUT_ST_SELECT && mode != IRBFN_ROLLOUT_ST_KS && mode != IRBFN_ROLLOUT_FULLINT &&
      mode != IRBFN_ROLLOUT_FRENET_LS)
    return IRBFN_ERR_UNSUPPORTED;
  if (net->O != 2 * T) return IRBFN_ERR_BAD_ARG;
  {
    // K1h nets: the whole tick in one launch where the instance exists (wide: plan_tick_wide.hip; narrow: rbf_tick_f16mfma)
    int rc = launch_tick_f16_wide(net, mode, x, mirror, state0, dp, controls, states, B, T, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
    rc = launch_tick_gram_narrow(net, mode, x, mirror, state0, dp, controls, states, B, T, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
    rc = launch_tick_f16_narrow(net, mode, x, mirror, state0, dp, controls, states, B, T, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  }
  {
    // several regions, sparse gate: forward + sign flip + roll-out in one launch of the region-sparse kernel
    const int forced = net->opt[IRBFN_OPT_FWD_KERNEL];
    if (forced == IRBFN_FWD_K1R || (forced == IRBFN_FWD_AUTO && sparse_preferred(net, B))) {
      const int rc = launch_forward_sparse(net, x, controls, B, mirror, T, mode, state0, &dp, states, T, s);
      if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
    }
  }
  if (controls && tick_through_controls(net, B)) {
    // wide outputs, and narrow ones on K1h without a one-launch instance: forward into the caller's controls buffer,
    // then the roll-out on split rows (K1h forward + K3 beat K1 with the roll-out in its epilogue: 134 + 17 vs 173 us
    // at config 2's size; the 2 x B x O x 4 bytes of control traffic are noise next to the B x N pair work)
    int rc = launch_forward(net, x, controls, B, s);
    if (rc == IRBFN_OK && mirror) rc = launch_unmirror(controls, mirror, B, net->O, T, s);
    if (rc != IRBFN_OK) return rc;
    return launch_rollout_forward_split(mode, state0, controls, dp, states, B, T, s);
  }
  if (net->bclass == BC_GENERIC) return IRBFN_ERR_UNSUPPORTED;
  if (T * rollout_state_dim(mode) > 64) return IRBFN_ERR_UNSUPPORTED;      // the epilogue's staging tile holds 64 floats per row
  FwdArgs a;
  fill_args(net, a, x, controls, B);
  a.state0 = state0;
  a.states = states;
  a.T = T;
  a.mode = mode;
  a.dp = dp;
  a.mirror = mirror;
  a.sv0 = T;
  return run_forward(net, a, true, s);
}

}  // namespace irbfn
