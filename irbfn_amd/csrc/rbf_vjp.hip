// K2: parameter VJP of the WCRBFNet forward (hand-derived).
//
// Replaces jax.value_and_grad(loss_fn)(state.params) restricted to the network
// (scripts/train_nmpc.py:297-298, scripts/train_nmpc_frenet.py:388-389,416-417): given the
// cotangent g[B,O] of `out`, produce d/d{centers, log_sigs, kernel, bias}.
//
//   hbar[b,k]  = sum_o g[b,o] W[k,o]                      G[b,r,k] = hbar[b,k] * gamma[b,r]
//   d centers[r,k,:] = sum_b G * phi'(d2) * sig^-2 * (-2)(x_b - c_rk)
//   d log_sigs[r,k]  = sum_b G * phi'(d2) * (-2 d2)
//   d kernel[k,o]    = sum_b sum_r gamma[b,r] phi[b,r,k] g[b,o]         d bias[o] = sum_b g[b,o]
//
// Mapping ("centre-stationary"): one LANE owns one centre n = r*K + k: c, sigma^-2, W[k,:] and the
// D+1+O gradient accumulators stay in VGPRs; the queries (x_b, g_b, gamma_b) are wave-uniform and
// stream through the scalar cache into SGPRs.  A workgroup = 4 waves on the same 64 centres and
// different query sub-slices; grid = centre groups x query slices.  Partial sums are combined
// without atomics: LDS across the 4 waves, a [slice][value][centre] slab in the workspace, then a
// fixed-order reduce kernel -> bitwise reproducible gradients.
#include "rbf_forward.h"
#include "rbf_vjp_f16.h"


namespace irbfn {

struct VjpArgs {
  const float* __restrict__ qrec;   // [B][QS] packed query records { x[0..D), gamma (R == 1), g[0..OP) }
  const float* __restrict__ gamma;  // [B][R]  (R > 1: per-lane region weight)
  const float* __restrict__ rec;    // [N][S]
  const float* __restrict__ sig2;   // [N]
  float* __restrict__ part;         // [QSB][V][Npad]
  long B;
  int Dreal, O, N, K, R, S, basis, Npad, per_wave;
  float gscale;
};

// d phi / d(d2) from phi and d2.  The fast classes need phi only; the generic class covers the bases that depend on
// d = sqrt(d2) itself (flax_rbf.py:55-111): d phi / d(d2) = phi'(d) * (0.5 / d), evaluated literally so that a query ON a
// centre (d = 0) gives what jax.grad gives there -- sqrt has no derivative at 0: inf, and inf * 0 = NaN.
// `ls_term` is the factor of the log_sig sum, in the units of `dphi_dd2 * r2` (the caller multiplies the finished sums by
// -2 / sigma^2): for the d-dependent bases it is phi'(d) * d / (2 s2), i.e. d log_sig = phi'(d) * (-d) -- the width does
// NOT go through the sqrt in the reference (flax_rbf.py:280: sqrt(sum sq) / exp(log_sig)), so its gradient is finite (0)
// for a query on a centre, where only the centre path is NaN.
template <int BC>
__device__ __forceinline__ float dphi_dd2(float phi, float d2, float r2, float s2, float gscale, int basis, float& ls_term) {
  float t;
  if constexpr (BC == BC_GAUSS) t = -gscale * phi;             // exp(-a d2)
  else if constexpr (BC == BC_IQ) t = -(phi * phi);            // 1/(1+d2)
  else if constexpr (BC == BC_IMQ) t = -0.5f * phi * phi * phi;      // (1+d2)^-1/2
  else {
    if (basis == IRBFN_MULTIQUADRIC) { t = 0.5f / phi; ls_term = t * r2; return t; }   // sqrt(1+d2)
    if (basis == IRBFN_QUADRATIC) { ls_term = r2; return 1.0f; }                          // d2
    const float d = sqrtf(d2);
    float dp;                                                  // phi'(d)
    switch (basis) {
      case IRBFN_LINEAR: dp = 1.0f; break;                                                    // d
      case IRBFN_SPLINE: dp = 2.0f * d * logf(d + 1.0f) + d2 / (d + 1.0f); break;             // d^2 log(d + 1)
      case IRBFN_POISSON_ONE: dp = (2.0f - d) * expf(-d); break;                              // (d - 1) exp(-d)
      case IRBFN_POISSON_TWO: dp = (2.0f * d - 1.0f - 0.5f * d2) * expf(-d); break;           // ((d - 2) / 2) d exp(-d)
      case IRBFN_MATERN32: dp = -3.0f * d * expf(-1.7320508075688772f * d); break;            // (1 + sqrt3 d) exp(-sqrt3 d)
      case IRBFN_MATERN52:                                                                    // (1 + sqrt5 d + 5/3 d^2) exp(-sqrt5 d)
        dp = -(5.0f / 3.0f) * d * (1.0f + 2.23606797749979f * d) * expf(-2.23606797749979f * d);
        break;
      default: dp = 0.0f; break;
    }
    ls_term = 0.5f * dp * d / s2;
    return dp * (0.5f / d);
  }
  ls_term = t * r2;
  return t;
}

// Pre-pass: one packed, zero-padded record per query so that the hot loop reads ONE contiguous scalar
// stream (s_load_dwordx16 + x4) without per-element bounds branches:
//   qrec[b] = { x[b, 0..D) (padded to DC), gamma[b] (R == 1; model.py:42-95), g[b, 0..O) (padded to OP) }.
// For R > 1 the full gamma[B][R] matrix is written as well (per-lane region weights).
__global__ __launch_bounds__(256) void vjp_pack_queries_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               float* __restrict__ qrec, float* __restrict__ gamma,
                                                               GateTables gt, long B, int D, int DC, int O, int OP,
                                                               int QS, int R) {
  // 64 queries per block, 256 threads; every global store runs over consecutive addresses (the first version gave a lane
  // one query: its R region weights went out as R dwords at a stride of R -- 87 us of the 128-region net's VJP at B = 80000)
  extern __shared__ float gtab[];                 // [E][64] gate factors, then [n_ranges][nsplit] table rows (as E * 64 offsets)
  const int tid = threadIdx.x;
  const long b0 = (long)blockIdx.x * kWave;
  const long left = B - b0;
  const int nv = left < kWave ? (int)left : kWave;
  const int E = gt.nsplit * gt.max_ranges;
  int* rows = reinterpret_cast<int*>(gtab + E * kWave);
  for (int idx = tid; idx < gt.n_ranges * gt.nsplit; idx += 256) {
    const int d = idx % gt.nsplit;
    rows[idx] = (d * gt.max_ranges + gt.dim_ranges[idx]) * kWave;
  }
  for (int idx = tid; idx < E * kWave; idx += 256) {
    const int e = idx >> 6, q = idx & (kWave - 1);
    const int d = e / gt.max_ranges;
    const long bb = b0 + (q < nv ? q : nv - 1);
    gtab[idx] = gate_factor(x[bb * D + d], gt.lo[e], gt.hi[e], gt.delta[d]);
  }
  __syncthreads();
  auto region_weight = [&](int q, int r) {        // model.py:70, 88-93
    if (r >= gt.n_ranges) return 0.0f;
    float gm = 1.0f;
    for (int d = 0; d < gt.nsplit; ++d) gm *= gtab[rows[r * gt.nsplit + d] + q];
    return gm;
  };
  // packed records: { x (padded to DC), gamma of region 0, g (padded to OP), zeros up to QS }
  float* qb = qrec + b0 * QS;
  for (int idx = tid; idx < nv * QS; idx += 256) {
    const int q = idx / QS, j = idx - q * QS;
    float v = 0.0f;
    if (j < D) v = x[(b0 + q) * D + j];
    else if (j == DC) v = region_weight(q, 0);
    else if (j > DC && j - DC - 1 < O) v = g[(b0 + q) * O + (j - DC - 1)];
    qb[idx] = v;
  }
  if (R > 1) {                                    // the full gamma[B][R] matrix (per-lane region weights of K2)
    float* gb = gamma + b0 * R;
    for (int idx = tid; idx < nv * R; idx += 256) {
      const int q = idx / R, r = idx - q * R;
      gb[idx] = region_weight(q, r);
    }
  }
}

template <int D, int OP, int BC, bool GATED>
__global__ __launch_bounds__(256) void rbf_vjp_kernel(const VjpArgs a) {
  extern __shared__ float lds[];
  constexpr int V = D + 1 + OP;
  constexpr int LP = kWave + 1;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x * kWave + lane;
  const int nn = n < a.N ? n : a.N - 1;

  // this lane's centre
  const float* rp = a.rec + (size_t)nn * a.S;
  float c[D], w[OP];
#pragma unroll
  for (int j = 0; j < D; ++j) c[j] = rp[j];
  const float sc = rp[D];
  const float s2 = a.sig2[nn];
#pragma unroll
  for (int o = 0; o < OP; ++o) w[o] = rp[D + 1 + o];
  const int r = nn / a.K;

  float gc[D], gw[OP], gls = 0.0f;
#pragma unroll
  for (int j = 0; j < D; ++j) gc[j] = 0.0f;
#pragma unroll
  for (int o = 0; o < OP; ++o) gw[o] = 0.0f;

  // this wave's query slice: records stream through the scalar cache (wave-uniform addresses)
  constexpr int QS = (D + 1 + OP + 3) & ~3;
  const long slice = (long)blockIdx.y * 4 + wave;
  const long b0 = slice * a.per_wave;
  long b1l = b0 + a.per_wave;
  b1l = b1l < a.B ? b1l : a.B;
  const int nq = b1l > b0 ? (int)(b1l - b0) : 0;
  const float* qp = a.qrec + b0 * QS;
  const float* gmp = a.gamma + b0 * a.R + r;     // GATED only
#pragma unroll 2
  for (int i = 0; i < nq; ++i, qp += QS) {
    float diff[D];
    float r2 = 0.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      diff[j] = qp[j] - c[j];
      r2 = __builtin_fmaf(diff[j], diff[j], r2);
    }
    const float d2 = r2 * s2;
    float phi;
    if constexpr (BC == BC_GAUSS) phi = fast_exp2(r2 * sc);
    else phi = basis_from_r2<BC>(r2, sc, a.basis);
    float gam;
    if constexpr (GATED) gam = gmp[(long)i * a.R];
    else gam = qp[D];
    const float gphi = gam * phi;
    // hbar as two interleaved partial sums: hipcc's SLP turns the pair into v_pk_fma_f32 with an SGPR-pair
    // operand (4.7 cycles per two FMAs; a v_fmac with one SGPR and two VGPR reads costs 4.2 for one).
    // Measured and rejected: 4 queries per step with one block of 4 transcendentals (as in the forward
    // kernel): 90 VGPRs, 411 vs 404 us -- this kernel is bound by its ~38 VALU instructions per pair.
    float hb0 = 0.0f, hb1 = 0.0f;
#pragma unroll
    for (int o = 0; o + 1 < OP; o += 2) {
      const float g0 = qp[D + 1 + o], g1 = qp[D + 2 + o];
      hb0 = __builtin_fmaf(g0, w[o], hb0);
      hb1 = __builtin_fmaf(g1, w[o + 1], hb1);
      gw[o] = __builtin_fmaf(gphi, g0, gw[o]);
      gw[o + 1] = __builtin_fmaf(gphi, g1, gw[o + 1]);
    }
    if constexpr (OP & 1) {
      const float g0 = qp[D + OP];
      hb0 = __builtin_fmaf(g0, w[OP - 1], hb0);
      gw[OP - 1] = __builtin_fmaf(gphi, g0, gw[OP - 1]);
    }
    const float hbar = hb0 + hb1;
    // the centre's own factor -2 / sigma^2 multiplies the finished sums, not every pair:
    // d log_sig += t * (-2 d2) = (-2 s2) * t * r2,   d centre += (-2 t s2) * diff = (-2 s2) * t * diff
    float lsf;
    const float hg = hbar * gam;
    const float t = hg * dphi_dd2<BC>(phi, d2, r2, s2, a.gscale, a.basis, lsf);
    if constexpr (BC == BC_GAUSS || BC == BC_IQ || BC == BC_IMQ) gls = __builtin_fmaf(t, r2, gls);
    else gls = __builtin_fmaf(hg, lsf, gls);
#pragma unroll
    for (int j = 0; j < D; ++j) gc[j] = __builtin_fmaf(t, diff[j], gc[j]);
  }
  {
    const float m2s = -2.0f * s2;
    gls *= m2s;
#pragma unroll
    for (int j = 0; j < D; ++j) gc[j] *= m2s;
  }

  // combine the 4 waves through LDS (fixed order), write the slab row of this block
  float* red = lds;                            // [4][V][LP]
#pragma unroll
  for (int j = 0; j < D; ++j) red[(wave * V + j) * LP + lane] = gc[j];
  red[(wave * V + D) * LP + lane] = gls;
#pragma unroll
  for (int o = 0; o < OP; ++o) red[(wave * V + D + 1 + o) * LP + lane] = gw[o];
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.y * V * a.Npad + (size_t)blockIdx.x * kWave;
  for (int idx = tid; idx < V * kWave; idx += 256) {
    const int v = idx >> 6, l = idx & (kWave - 1);
    const float s = (red[(0 * V + v) * LP + l] + red[(1 * V + v) * LP + l]) +
                    (red[(2 * V + v) * LP + l] + red[(3 * V + v) * LP + l]);
    dst[(size_t)v * a.Npad + l] = s;
  }
}

// Reduce the slabs.  Stage 1: one thread per (value v, centre n) sums the query slices in slice order (consecutive threads
// = consecutive centres: coalesced) -> d centers, d log_sigs, and the per-CENTRE Dense gradient, which for one region IS
// d kernel and for R > 1 goes back into slab 0 (only this thread touches those words).  Stage 2 (R > 1): d kernel[k, o] =
// sum over the regions of the per-centre values, one wave per output, fixed tree.  (Round 1 summed R x QSB terms per
// output value in ONE thread: 100 threads x 128 regions x 24 slices of dependent loads = 2.2 ms of the 2.6 ms training
// step of the reference's 128-region net at its batch size of 80000.)
// Row V of the grid (blockIdx.y == V) is the second stage of the bias gradient: block o sums the per-block column sums of g
// (colsum_partial_kernel) -- a launch of its own until round 3.
__global__ __launch_bounds__(256) void vjp_reduce_kernel(float* __restrict__ part, float* __restrict__ g_centers,
                                                         float* __restrict__ g_log_sigs, float* __restrict__ g_kernel, int QSB,
                                                         int V, int Npad, int N, int K, int R, int D, int DC, int O,
                                                         const float* __restrict__ bpart, float* __restrict__ g_bias, int bias_blocks) {
  // block = (value v, 64 consecutive centres) x 4 slice groups: thread (sg, l) sums slices sg, sg + 4, ... in order, the
  // four partial sums are added in a fixed order (small nets have few centre groups and hundreds of slices)
  __shared__ float sm[4][kWave];
  const int l = threadIdx.x & (kWave - 1), sg = threadIdx.x >> 6;
  if (blockIdx.y == (unsigned)V) {
    const int o = blockIdx.x, t = threadIdx.x;
    if (o >= O || bpart == nullptr) return;
    float* sb = &sm[0][0];
    float sacc = 0.0f;
    for (int b = t; b < bias_blocks; b += 256) sacc += bpart[(size_t)b * O + o];
    sb[t] = sacc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {             // fixed tree -> deterministic
      if (t < w) sb[t] += sb[t + w];
      __syncthreads();
    }
    if (t == 0) g_bias[o] = sb[0];
    return;
  }
  const int v = blockIdx.y, n = blockIdx.x * kWave + l;
  if ((v >= D && v < DC) || v > DC + O) return;            // padded coordinate / output slots (whole block)
  float s = 0.0f;
  if (n < N)
    for (int q = sg; q < QSB; q += 4) s += part[((size_t)q * V + v) * Npad + n];
  sm[sg][l] = s;
  __syncthreads();
  if (sg != 0 || n >= N) return;
  s = (sm[0][l] + sm[1][l]) + (sm[2][l] + sm[3][l]);
  if (v < D) g_centers[(size_t)n * D + v] = s;
  else if (v == DC) g_log_sigs[n] = s;
  else if (R == 1) g_kernel[(size_t)n * O + (v - DC - 1)] = s;
  else part[(size_t)v * Npad + n] = s;                      // slab 0
}

__global__ __launch_bounds__(64) void vjp_reduce_regions_kernel(const float* __restrict__ part, float* __restrict__ g_kernel,
                                                                int Npad, int K, int R, int DC, int O) {
  const int k = blockIdx.x, o = blockIdx.y, lane = threadIdx.x;
  float s = 0.0f;
  for (int r = lane; r < R; r += kWave) s += part[(size_t)(DC + 1 + o) * Npad + (size_t)r * K + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);      // fixed tree: deterministic
  if (lane == 0) g_kernel[(size_t)k * O + o] = s;
}

int launch_vjp_reduce(const irbfn_net* net, float* part, float* g_centers, float* g_log_sigs, float* g_kernel, int QSB, int V,
                      int Npad, hipStream_t s, const float* bpart, float* g_bias, int bias_blocks) {
  unsigned gx = (unsigned)((net->N + kWave - 1) / kWave);
  if (bpart != nullptr && gx < (unsigned)net->O) gx = (unsigned)net->O;
  hipLaunchKernelGGL(vjp_reduce_kernel, dim3(gx, V + (bpart != nullptr ? 1 : 0)), dim3(256), 0, s, part, g_centers,
                     g_log_sigs, g_kernel, QSB, V, Npad, net->N, net->K, net->R, net->D, net->DC, net->O, bpart, g_bias, bias_blocks);
  IRBFN_HIP_CHECK(hipGetLastError());
  if (net->R > 1) {
    hipLaunchKernelGGL(vjp_reduce_regions_kernel, dim3(net->K, net->O), dim3(kWave), 0, s, part, g_kernel, Npad, net->K, net->R,
                       net->DC, net->O);
    IRBFN_HIP_CHECK(hipGetLastError());
  }
  return IRBFN_OK;
}

// d bias[o] = sum_b g[b,o]: per-block column sums, then one block finishes (fixed order)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ g, float* __restrict__ part,
                                                              long B, int O, long rows_per_block,
                                                              float* __restrict__ bmax) {
  extern __shared__ float sm[];                // [256]
  __shared__ float smax[256];
  float mx = 0.0f;                             // max |g| of this block's rows (K2h operand scale), NaN wins
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  r1 = r1 < B ? r1 : B;
  const long total = (r1 - r0) * O;
  const float* base = g + r0 * O;
  // thread t sums flat indices i = t + k*stride with stride a multiple of O -> fixed column per thread
  const int per = 256 / O > 0 ? 256 / O : 1;   // rows covered per sweep when O <= 256
  if (O <= 256) {
    const int stride = per * O;
    const int t = threadIdx.x;
    float s = 0.0f;
    if (t < stride)
      for (long i = t; i < total; i += stride) {
        const float v = base[i];
        s += v;
        const float av = fabsf(v);
        mx = (av > mx || av != av) ? av : mx;
      }
    sm[t] = t < stride ? s : 0.0f;
    __syncthreads();
    if (t < O) {
      float acc = 0.0f;
      for (int p = 0; p < per; ++p) acc += sm[p * O + t];
      part[(size_t)blockIdx.x * O + t] = acc;
    }
  } else {
    for (int o = threadIdx.x; o < O; o += 256) {
      float s = 0.0f;
      for (long rr = 0; rr < r1 - r0; ++rr) {
        const float v = base[rr * O + o];
        s += v;
        const float av = fabsf(v);
        mx = (av > mx || av != av) ? av : mx;
      }
      part[(size_t)blockIdx.x * O + o] = s;
    }
  }
  if (bmax) {                                  // block-uniform
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (threadIdx.x < w) {
        const float o2 = smax[threadIdx.x + w], m2 = smax[threadIdx.x];
        smax[threadIdx.x] = (o2 > m2 || o2 != o2) ? o2 : m2;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) bmax[blockIdx.x] = smax[0];
  }
}

// ------------------------------------------------------------------------------------------------
struct VjpPlan {
  int groups, QSB, per_wave, Npad, V, bias_blocks;
  long rows_per_block;
  size_t off_gamma, off_part, off_bias, off_qrec, off_qblk, off_misc, total;
  int QS;
  bool use_h;      // K2h (matrix-core VJP) instead of K2
  bool use_g;      // K2g (u and the centre gradients on the matrix cores too) in front of K2h
  int CT;
  bool use_sp;     // K2r (region-sparse VJP, rbf_sparse.hip) instead of K2
  int SL;          // K2r: slices per region (<= QSB slabs are allocated)
  size_t off_sp, off_sp_part;   // K2r: pair lists, slabs
};

static VjpPlan make_plan(const irbfn_net* net, int64_t B) {
  VjpPlan p;
  p.groups = (net->N + kWave - 1) / kWave;
  p.Npad = p.groups * kWave;
  p.V = net->DC + 1 + net->OP;
  long qsb = (8192 + (long)p.groups * 4 - 1) / ((long)p.groups * 4);
  long max_qsb = (B + 4 * 64 - 1) / (4 * 64);       // >= 64 queries per wave
  if (max_qsb < 1) max_qsb = 1;
  if (qsb > max_qsb) qsb = max_qsb;
  if (qsb < 1) qsb = 1;
  if (qsb > 4096) qsb = 4096;
  p.QSB = (int)qsb;
  const long slices = qsb * 4;
  p.per_wave = (int)((B + slices - 1) / slices);
  if (p.per_wave < 1) p.per_wave = 1;
  p.bias_blocks = (int)(B < 256 * 64 ? (B + 63) / 64 : 256);
  if (p.bias_blocks < 1) p.bias_blocks = 1;
  p.rows_per_block = (B + p.bias_blocks - 1) / p.bias_blocks;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  p.off_gamma = off; off += al((size_t)(net->R > 1 ? B * net->R : 1) * sizeof(float));
  p.off_part = off;  off += al((size_t)p.QSB * p.V * p.Npad * sizeof(float));
  p.off_bias = off;  off += al((size_t)p.bias_blocks * net->O * sizeof(float));
  p.QS = (net->DC + 1 + net->OP + 3) & ~3;
  p.off_qrec = off;  off += al((size_t)B * p.QS * sizeof(float));
  // K2h: packed 32-query blocks + (absmax, scales); IRBFN_OPT_VJP_KERNEL = IRBFN_VJP_K2 / IRBFN_VJP_K2H forces one
  const int vk = net->opt[IRBFN_OPT_VJP_KERNEL];
  p.use_h = vjph_eligible(net) && vk != IRBFN_VJP_K2 && B >= 2048;
  p.CT = net->opt[IRBFN_OPT_VJP_F16_CT] == 4 ? 4 : 2;
  const size_t blkb = vjph_block_bytes(net) > vjpg_block_bytes() ? vjph_block_bytes(net) : vjpg_block_bytes();
  p.off_qblk = off;  off += al(vjph_eligible(net) ? (size_t)((B + 31) / 32) * blkb : 0);
  // K2g in front of K2h from 8192 queries and 2.5e7 (query, centre) pairs (tools/sweep_vjp_batch.py: 4096 centres: K2g 57.6 vs K2h 64.3 us at
  // B = 12288, a tie at 8192; 1000 centres: 50.1 vs 52.1 at 24576, 46.3 vs 45.0 at 16384)
  p.use_g = p.use_h && vjpg_eligible(net) && (vk == IRBFN_VJP_K2G || (vk == IRBFN_VJP_AUTO && B >= 8192 && (long long)B * net->N >= 25000000LL)) && net->opt[IRBFN_OPT_VJP_F16_CT] == 0;
  p.off_misc = off;  off += al((size_t)(p.bias_blocks + 8) * sizeof(float));
  // K2r: several regions with a sparse gate (automatic where the forward takes K1r; IRBFN_VJP_K2R forces it where eligible)
  p.use_sp = sparse_vjp_eligible(net) && (vk == IRBFN_VJP_K2R || (vk == IRBFN_VJP_AUTO && sparse_preferred(net, B)));
  p.SL = 0;
  p.off_sp = off;
  p.off_sp_part = off;
  if (sparse_vjp_eligible(net)) {
    off += al(sparse_vjp_workspace_bytes(net, B));
    p.SL = sparse_vjp_slices(net, B);
    p.off_sp_part = off;                          // K2r's own slabs [SL][V][Npad]
    off += al((size_t)p.SL * p.V * p.Npad * sizeof(float));
  }
  if (p.use_g) {
    // K2g: a block = 4 waves = 128 centres; slices so that the launch is ONE resident round (3 or 4 waves per SIMD: 768 / 1024 blocks), at
    // least 8 query blocks each (profiles/r03_vjp_qsb_sweep.txt: config 3 198 -> 195 us, the reference's 1000-centre net at
    // B = 80000 107 -> 101 us against 1024 blocks)
    const long nqb = (B + 31) / 32;
    const long gb = ((net->N + 31) / 32 + 3) / 4;
    const long resident = net->O <= 10 ? 1024 : 768;       // rbf_vjp_gram.hip: 4 waves per SIMD where hbar is one MFMA (O <= 10), else 3
    long q2 = (resident + gb - 1) / gb;
    if (q2 > 64) q2 = 64;                                  // small nets: the slab reduce grows with the slices (N = 1000, B = 80000: 64 slices 84 us, 96: 91)
    if (q2 * 8 > nqb) q2 = (nqb + 7) / 8;
    if (net->opt[IRBFN_OPT_VJP_QSB] > 0) q2 = net->opt[IRBFN_OPT_VJP_QSB];
    if (q2 < 1) q2 = 1;
    if (q2 < p.QSB) p.QSB = (int)q2;          // never more slabs than were allocated above
  } else if (p.use_h) {
    // fewer, longer query slices than K2 (3 waves per SIMD resident): halves the slab traffic of the reduce kernel
    const long gh = (net->N + 16 * p.CT - 1) / (16 * p.CT);
    long q2 = (6144 + gh * 4 - 1) / (gh * 4);
    const long nqb = (B + 31) / 32;
    if (q2 * 4 > nqb) q2 = (nqb + 3) / 4;
    if (q2 < 1) q2 = 1;
    if (q2 < p.QSB) p.QSB = (int)q2;          // never more slabs than were allocated above
  }
  p.total = off;
  return p;
}

// ---- ClusterWCRBFNet (src/irbfn_mpc/model.py:341-414): cotangent of the region weights ------------------------------
// d gamma[b,r] = sum_k hbar[b,k] phi[b,r,k],  hbar = g W^T  (the path from `out` back to the softmax gate; the tanh
// gate of WCRBFNet has no parameters, so K2 never needs it).  One lane per query, one wave per (64 queries, region):
// the region's K centre records are wave-uniform scalar streams, g[b,:] sits in registers.
template <int D, int OP>
__global__ __launch_bounds__(64) void dgamma_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                    const float* __restrict__ rec, float* __restrict__ dgamma, long B,
                                                    int Dreal, int O, int K, int R, int S, int bclass, int basis) {
  const int lane = threadIdx.x, r = blockIdx.y;
  const long b = (long)blockIdx.x * kWave + lane;
  const long bb = b < B ? b : B - 1;
  float xq[D], gq[OP];
#pragma unroll
  for (int j = 0; j < D; ++j) xq[j] = j < Dreal ? x[bb * Dreal + j] : 0.0f;
#pragma unroll
  for (int o = 0; o < OP; ++o) gq[o] = o < O ? g[bb * O + o] : 0.0f;
  float acc = 0.0f;
  const float* rp = rec + (size_t)r * K * S;
  for (int k = 0; k < K; ++k, rp += S) {
    float r2 = 0.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float d = xq[j] - rp[j];
      r2 = __builtin_fmaf(d, d, r2);
    }
    float phi;
    switch (bclass) {
      case BC_GAUSS: phi = basis_from_r2<BC_GAUSS>(r2, rp[D], basis); break;
      case BC_IQ: phi = basis_from_r2<BC_IQ>(r2, rp[D], basis); break;
      case BC_IMQ: phi = basis_from_r2<BC_IMQ>(r2, rp[D], basis); break;
      default: phi = basis_from_r2<BC_GENERIC>(r2, rp[D], basis); break;
    }
    float hb = 0.0f;
#pragma unroll
    for (int o = 0; o < OP; ++o) hb = __builtin_fmaf(gq[o], rp[D + 1 + o], hb);
    acc = __builtin_fmaf(hb, phi, acc);
  }
  if (b < B) dgamma[b * R + r] = acc;
}

template <int D>
static int launch_dgamma_d(const irbfn_net* net, const float* x, const float* g, float* dgamma, int64_t B, hipStream_t s) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave), net->R), block(kWave);
#define IRBFN_DG(OPV)                                                                                                \
  case OPV:                                                                                                          \
    hipLaunchKernelGGL((dgamma_kernel<D, OPV>), grid, block, 0, s, x, g, net->rec, dgamma, (long)B, net->D, net->O,  \
                       net->K, net->R, net->S, net->bclass, net->basis);                                             \
    break;
  switch (net->OP) {
    IRBFN_DG(2) IRBFN_DG(4) IRBFN_DG(5) IRBFN_DG(8) IRBFN_DG(10) IRBFN_DG(16)
    default: return IRBFN_ERR_UNSUPPORTED;       // O <= 16
  }
#undef IRBFN_DG
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int launch_dgamma(irbfn_net* net, const float* x, const float* gout, float* dgamma, int64_t B, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  switch (net->DC) {
    case 3: return launch_dgamma_d<3>(net, x, gout, dgamma, B, s);
    case 4: return launch_dgamma_d<4>(net, x, gout, dgamma, B, s);
    case 7: return launch_dgamma_d<7>(net, x, gout, dgamma, B, s);
    case 8: return launch_dgamma_d<8>(net, x, gout, dgamma, B, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

// softmax backward of the cluster gate (model.py:402-404): d logits = gamma * (d gamma - sum_s gamma_s d gamma_s)
// [+ the direct cotangent of the logits, e.g. of the cluster cross-entropy], one thread per query
__global__ __launch_bounds__(256) void cluster_dlogits_kernel(const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                              const float* __restrict__ glogits, float* __restrict__ dlogits,
                                                              long B, int R) {
  const long b = (long)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float dot = 0.0f;
  for (int r = 0; r < R; ++r) dot = __builtin_fmaf(gamma[b * R + r], dgamma[b * R + r], dot);
  for (int r = 0; r < R; ++r) {
    float v = gamma[b * R + r] * (dgamma[b * R + r] - dot);
    if (glogits) v += glogits[b * R + r];
    dlogits[b * R + r] = v;
  }
}

// d Wc[d,r] = sum_b x[b,d] dlogits[b,r] (d < D), d bc[r] = sum_b dlogits[b,r] (d == D).  kGateBwdBlocks blocks walk the
// batch in 64-row tiles (coalesced copies into LDS), thread t owns outputs t, t + 256, ... and adds the tile's rows in
// order; the per-block partial sums are added in block order by cluster_dense_final_kernel: deterministic.  (The first
// version gave one block to an output and let it stride over the whole batch: 99 blocks, uncoalesced -- 102 us at 80000 rows.)
constexpr int kGateBwdBlocks = 256;
__global__ __launch_bounds__(256) void cluster_dense_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dlogits,
                                                                float* __restrict__ part, long B, int D, int R) {
  extern __shared__ float lds[];                 // xs[64][D + 1] (last column = 1), dl[64][R]
  float* xs = lds;
  float* dl = lds + kWave * (D + 1);
  const int tid = threadIdx.x;
  const int nout = (D + 1) * R;
  constexpr int MAXO = 8;                        // outputs per thread: (D + 1) * R <= 2048
  float acc[MAXO];
#pragma unroll
  for (int k = 0; k < MAXO; ++k) acc[k] = 0.0f;
  const long ntiles = (B + kWave - 1) / kWave;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long b0 = tile * kWave;
    const long left = B - b0;
    const int nv = left < kWave ? (int)left : kWave;
    for (int i = tid; i < kWave * (D + 1); i += 256) {
      const int row = i / (D + 1), d = i - row * (D + 1);
      xs[i] = row < nv ? (d < D ? x[(b0 + row) * D + d] : 1.0f) : 0.0f;
    }
    for (int i = tid; i < kWave * R; i += 256) dl[i] = i < nv * R ? dlogits[b0 * R + i] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXO; ++k) {
      const int t = tid + k * 256;
      if (t < nout) {
        const int d = t / R, r = t - d * R;
        float s = acc[k];
        for (int row = 0; row < kWave; ++row) s = __builtin_fmaf(xs[row * (D + 1) + d], dl[row * R + r], s);
        acc[k] = s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < MAXO; ++k) {
    const int t = tid + k * 256;
    if (t < nout) part[(size_t)blockIdx.x * nout + t] = acc[k];
  }
}

__global__ __launch_bounds__(256) void cluster_dense_final_kernel(const float* __restrict__ part, float* __restrict__ g_wc,
                                                                  float* __restrict__ g_bc, int nblk, int D, int R) {
  // 16 outputs per block x 16 block groups: thread (sg, v) adds partials sg, sg + 16, ... in order, then a fixed tree
  __shared__ float sm[16][17];
  const int v = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + v, nout = (D + 1) * R;
  float s = 0.0f;
  if (t < nout)
    for (int k = sg; k < nblk; k += 16) s += part[(size_t)k * nout + t];
  sm[sg][v] = s;
  __syncthreads();
  for (int w = 8; w > 0; w >>= 1) {
    if (sg < w) sm[sg][v] += sm[sg + w][v];
    __syncthreads();
  }
  if (sg != 0 || t >= nout) return;
  if (t < D * R) g_wc[t] = sm[0][v];
  else g_bc[t - D * R] = sm[0][v];
}

int64_t cluster_gate_vjp_workspace_bytes(int D, int R) { return (int64_t)kGateBwdBlocks * (D + 1) * R * (int64_t)sizeof(float); }

int launch_cluster_gate_vjp(const float* x, const float* gamma, const float* dgamma, const float* glogits, float* dlogits,
                            float* g_wc, float* g_bc, int64_t B, int D, int R, float* ws, hipStream_t s) {
  if (B == 0) {
    IRBFN_HIP_CHECK(hipMemsetAsync(g_wc, 0, (size_t)D * R * sizeof(float), s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_bc, 0, (size_t)R * sizeof(float), s));
    return IRBFN_OK;
  }
  if ((D + 1) * R > 2048) return IRBFN_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(cluster_dlogits_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, gamma, dgamma, glogits, dlogits,
                     (long)B, R);
  IRBFN_HIP_CHECK(hipGetLastError());
  const size_t lds = (size_t)kWave * (D + 1 + R) * sizeof(float);
  const long ntiles = (B + kWave - 1) / kWave;
  const int nblk = ntiles < kGateBwdBlocks ? (int)ntiles : kGateBwdBlocks;
  hipLaunchKernelGGL(cluster_dense_bwd_kernel, dim3(nblk), dim3(256), lds, s, x, dlogits, ws, (long)B, D, R);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(cluster_dense_final_kernel, dim3(((D + 1) * R + 15) / 16), dim3(256), 0, s, ws, g_wc, g_bc, nblk, D, R);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// optax.softmax_cross_entropy(logits, labels).mean() (scripts/train_nmpc_frenet.py:431) and its cotangent of the logits:
// loss_b = -sum_r labels[b,r] log_softmax(logits)[b,r];  d loss / d logits = (softmax * sum_r labels - labels) / B.
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                           float* __restrict__ glogits, float* __restrict__ loss_part, long B,
                                                           int R) {
  __shared__ float sm[256];
  float lsum = 0.0f;
  const float invB = 1.0f / (float)B;
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
    float mx = -INFINITY;
    for (int r = 0; r < R; ++r) mx = fmaxf(mx, logits[b * R + r]);
    float se = 0.0f, sl = 0.0f;
    for (int r = 0; r < R; ++r) { se += expf(logits[b * R + r] - mx); sl += labels[b * R + r]; }
    const float lse = mx + logf(se);
    for (int r = 0; r < R; ++r) {
      const float lp = logits[b * R + r] - lse;
      lsum += -labels[b * R + r] * lp * invB;
      glogits[b * R + r] = (expf(lp) * sl - labels[b * R + r]) * invB;
    }
  }
  sm[threadIdx.x] = lsum;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_part[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void xent_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out, int accumulate) {
  __shared__ float sm[256];
  float v = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) v += part[i];
  sm[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + sm[0] : sm[0];
}

int launch_softmax_xent(const float* logits, const float* labels, float* glogits, float* loss, float* partials, int accumulate,
                        int64_t B, int R, hipStream_t s) {
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(kRedBlocks), dim3(256), 0, s, logits, labels, glogits, partials, (long)B, R);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(256), 0, s, partials, kRedBlocks, loss, accumulate);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int64_t vjp_workspace_bytes(const irbfn_net* net, int64_t B) {
  if (B <= 0) return 0;
  return (int64_t)make_plan(net, B).total;
}

template <int D, int OP>
static int launch_vjp_bc(const VjpArgs& a, int bc, bool gated, dim3 grid, hipStream_t s) {
  constexpr int V = D + 1 + OP;
  const size_t lds = (size_t)4 * V * (kWave + 1) * sizeof(float);
#define IRBFN_VCASE(BCV)                                                                                  \
  case BCV: {                                                                                             \
    if (gated) {                                                                                          \
      auto k = rbf_vjp_kernel<D, OP, BCV, true>;                                                          \
      if (lds > 48 * 1024)                                                                                \
        IRBFN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                                  \
    } else {                                                                                              \
      auto k = rbf_vjp_kernel<D, OP, BCV, false>;                                                         \
      if (lds > 48 * 1024)                                                                                \
        IRBFN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
      hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                                  \
    }                                                                                                     \
    break;                                                                                                \
  }
  switch (bc) {
    IRBFN_VCASE(BC_GAUSS)
    IRBFN_VCASE(BC_IQ)
    IRBFN_VCASE(BC_IMQ)
    IRBFN_VCASE(BC_GENERIC)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_VCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int D>
static int launch_vjp_d(const VjpArgs& a, int OP, int bc, bool gated, dim3 grid, hipStream_t s) {
  switch (OP) {
    case 2: return launch_vjp_bc<D, 2>(a, bc, gated, grid, s);
    case 4: return launch_vjp_bc<D, 4>(a, bc, gated, grid, s);
    case 5: return launch_vjp_bc<D, 5>(a, bc, gated, grid, s);
    case 8: return launch_vjp_bc<D, 8>(a, bc, gated, grid, s);
    case 10: return launch_vjp_bc<D, 10>(a, bc, gated, grid, s);
    case 16: return launch_vjp_bc<D, 16>(a, bc, gated, grid, s);
    case 32: return launch_vjp_bc<D, 32>(a, bc, gated, grid, s);
    case 64: return launch_vjp_bc<D, 64>(a, bc, gated, grid, s);
    case 100: return launch_vjp_bc<D, 100>(a, bc, gated, grid, s);
    case 128: return launch_vjp_bc<D, 128>(a, bc, gated, grid, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

int launch_vjp(irbfn_net* net, const float* x, const float* gout, float* g_centers, float* g_log_sigs,
               float* g_kernel, float* g_bias, int64_t B, void* ws, int64_t ws_bytes, hipStream_t s,
               const float* gamma_ext) {
  (void)ws_bytes;
  const long n_c = (long)net->N * net->D, n_l = net->N, n_k = (long)net->K * net->O;
  if (B == 0) {
    IRBFN_HIP_CHECK(hipMemsetAsync(g_centers, 0, n_c * sizeof(float), s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_log_sigs, 0, n_l * sizeof(float), s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_kernel, 0, n_k * sizeof(float), s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_bias, 0, (size_t)net->O * sizeof(float), s));
    return IRBFN_OK;
  }
  VjpPlan p = make_plan(net, B);
  if (gamma_ext) p.use_h = false;                // caller-provided region weights (ClusterWCRBFNet): the gated K2
  if (net->opt[IRBFN_OPT_VJP_KERNEL] == IRBFN_VJP_K2G && !(p.use_h && p.use_g)) return IRBFN_ERR_UNSUPPORTED;   // a forced kernel that cannot take the net
  char* base = static_cast<char*>(ws);
  float* gamma = reinterpret_cast<float*>(base + p.off_gamma);
  float* part = reinterpret_cast<float*>(base + p.off_part);
  float* bpart = reinterpret_cast<float*>(base + p.off_bias);
  const long total_h = n_c + n_l + n_k;
  if (p.use_h) {
    float* bmax = reinterpret_cast<float*>(base + p.off_misc);           // [bias_blocks] max |g| per block
    float* scales = bmax + p.bias_blocks;                                // [2]
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(p.bias_blocks), dim3(256), 256 * sizeof(float), s, gout, bpart,
                       (long)B, net->O, p.rows_per_block, bmax);
    IRBFN_HIP_CHECK(hipGetLastError());
    int rch;
    const int* run_if = nullptr;
    int run_gen = 0;
    if (p.use_g) {
      // K2g first; a query outside its representable box raises the flag, K2g returns at once and K2h -- launched behind it
      // with the complementary test -- does the work
      // The flag is a generation number in a small ring of net-owned words (zero at creation): call `gen` stores gen into word
      // gen % 64 when it has such a query, nobody resets anything -- a memset per call was a launch of its own.
      if (++net->vjp_gen <= 0) net->vjp_gen = 1;
      run_gen = net->vjp_gen;
      int* flag = net->vjp_flags + (run_gen & 63);
      rch = launch_vjp_gram(net, x, gout, B, reinterpret_cast<unsigned char*>(base + p.off_qblk), bmax, p.bias_blocks, scales, flag,
                            run_gen, part, p.QSB, p.Npad, s);
      if (rch != IRBFN_OK) return rch;
      run_if = flag;
    } else if (net->opt[IRBFN_OPT_VJP_KERNEL] == IRBFN_VJP_K2G) {
      return IRBFN_ERR_UNSUPPORTED;
    }
    rch = launch_vjp_f16(net, x, gout, B, reinterpret_cast<unsigned char*>(base + p.off_qblk), bmax, p.bias_blocks,
                         scales, part, p.QSB, p.Npad, p.CT, s, run_if, run_gen);
    if (rch != IRBFN_OK) return rch;
    return launch_vjp_reduce(net, part, g_centers, g_log_sigs, g_kernel, p.QSB, p.V, p.Npad, s, bpart, g_bias, p.bias_blocks);
  }

  if (p.use_sp && !gamma_ext) {
    // K2r: pair lists per region -> one wave per (region, slice) -> the same fixed-order slab reduce; no query packing
    float* sp_part = reinterpret_cast<float*>(base + p.off_sp_part);
    int rcs = launch_vjp_sparse(net, x, gout, B, base + p.off_sp, sp_part, p.SL, p.Npad, s);
    if (rcs != IRBFN_ERR_UNSUPPORTED) {
      if (rcs != IRBFN_OK) return rcs;
      hipLaunchKernelGGL(colsum_partial_kernel, dim3(p.bias_blocks), dim3(256), 256 * sizeof(float), s, gout, bpart,
                         (long)B, net->O, p.rows_per_block, (float*)nullptr);
      IRBFN_HIP_CHECK(hipGetLastError());
      return launch_vjp_reduce(net, sp_part, g_centers, g_log_sigs, g_kernel, p.SL, p.V, p.Npad, s, bpart, g_bias, p.bias_blocks);
    }
    if (net->opt[IRBFN_OPT_VJP_KERNEL] == IRBFN_VJP_K2R) return IRBFN_ERR_UNSUPPORTED;
  } else if (net->opt[IRBFN_OPT_VJP_KERNEL] == IRBFN_VJP_K2R && !gamma_ext) {
    return IRBFN_ERR_UNSUPPORTED;                  // a forced kernel that cannot take the net
  }
  float* qrec = reinterpret_cast<float*>(base + p.off_qrec);
  {
    const size_t glds = ((size_t)net->nsplit * net->max_ranges * kWave + (size_t)net->gate().n_ranges * net->nsplit) * sizeof(float);
    if (glds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vjp_pack_queries_kernel, dim3((unsigned)((B + kWave - 1) / kWave)), dim3(256), glds, s, x, gout,
                       qrec, gamma, net->gate(), (long)B, net->D, net->DC, net->O, net->OP, p.QS, net->R);
    IRBFN_HIP_CHECK(hipGetLastError());
  }
  int rc = IRBFN_OK;

  VjpArgs a;
  a.qrec = qrec; a.gamma = gamma_ext ? gamma_ext : gamma; a.rec = net->rec; a.sig2 = net->sig2; a.part = part;
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.N = net->N; a.K = net->K; a.R = net->R; a.S = net->S;
  a.basis = net->basis; a.Npad = p.Npad; a.per_wave = p.per_wave; a.gscale = gauss_scale(net->basis);
  const dim3 grid(p.groups, p.QSB);
  const bool gated = net->R > 1 || gamma_ext != nullptr;
  switch (net->DC) {
    case 3: rc = launch_vjp_d<3>(a, net->OP, net->bclass, gated, grid, s); break;
    case 4: rc = launch_vjp_d<4>(a, net->OP, net->bclass, gated, grid, s); break;
    case 7: rc = launch_vjp_d<7>(a, net->OP, net->bclass, gated, grid, s); break;
    case 8: rc = launch_vjp_d<8>(a, net->OP, net->bclass, gated, grid, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc != IRBFN_OK) return rc;

  hipLaunchKernelGGL(colsum_partial_kernel, dim3(p.bias_blocks), dim3(256), 256 * sizeof(float), s, gout, bpart,
                     (long)B, net->O, p.rows_per_block, (float*)nullptr);
  IRBFN_HIP_CHECK(hipGetLastError());
  return launch_vjp_reduce(net, part, g_centers, g_log_sigs, g_kernel, p.QSB, p.V, p.Npad, s, bpart, g_bias, p.bias_blocks);
}

}  // namespace irbfn
