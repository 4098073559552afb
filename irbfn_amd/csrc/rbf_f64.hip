// float64 mode: WCRBFNet forward and parameter VJP evaluated in float64 on the f64 vector pipe.
//
// The reference trains and evaluates in float64 under --use_float64 (scripts/train_nmpc.py:41-42: jax_enable_x64; several of
// its checkpoints hold float64 centers / log_sigs).  The float32 kernels stay within 1e-5 of such a run;
// this file is the mode itself: the same expressions (flax_rbf.py:34-111, :275-283; model.py:42-95, :187-196; App. A.2 for
// the VJP) in double, straight from the checkpoint-layout arrays, tanh / exp from ocml's double versions (the gate's zero is
// float64's: tanh saturates at |z| ~ 19, not at 9.01 as in float32).  Not a throughput path (fp64 vector peak 78 TFLOP/s, the
// transcendental functions are software): plain lane-per-query / lane-per-centre kernels, deterministic (no atomics).
#include <math.h>
#include <string.h>

#include "common.h"

namespace irbfn {

struct F64Card {                                 // device copy of irbfn_f64_card
  int D, R, K, O, basis, nsplit, max_ranges, n_ranges;
  const double* lo;
  const double* hi;
  const double* delta;
  const int* dim_ranges;
};

__device__ __forceinline__ double basis_f64(double d2, int basis) {      // flax_rbf.py:34-111 as functions of d^2 = (d)^2
  const double d = sqrt(d2);
  switch (basis) {
    case IRBFN_GAUSSIAN: return exp(-1.0 * d2);
    case IRBFN_GAUSSIAN_WIDE: return exp(-0.1 * d2);
    case IRBFN_GAUSSIAN_WIDER: return exp(-0.01 * d2);
    case IRBFN_INVERSE_QUADRATIC: return 1.0 / (1.0 + d2);
    case IRBFN_LINEAR: return d;
    case IRBFN_QUADRATIC: return d2;
    case IRBFN_MULTIQUADRIC: return sqrt(1.0 + d2);
    case IRBFN_INVERSE_MULTIQUADRIC: return 1.0 / sqrt(1.0 + d2);
    case IRBFN_SPLINE: return d2 * log(d + 1.0);
    case IRBFN_POISSON_ONE: return (d - 1.0) * exp(-d);
    case IRBFN_POISSON_TWO: return ((d - 2.0) / 2.0) * d * exp(-d);
    case IRBFN_MATERN32: return (1.0 + 1.7320508075688772 * d) * exp(-1.7320508075688772 * d);
    case IRBFN_MATERN52: return (1.0 + 2.23606797749979 * d + (5.0 / 3.0) * d2) * exp(-2.23606797749979 * d);
    default: return 0.0;
  }
}

// d phi / d(d^2) and the log_sig factor (units of dphi_dd2 * r2; the caller multiplies by -2 s2): as rbf_vjp.hip
__device__ __forceinline__ double dphi_dd2_f64(double phi, double d2, double r2, double s2, int basis, double& ls_term) {
  double t;
  switch (basis) {
    case IRBFN_GAUSSIAN: t = -phi; break;
    case IRBFN_GAUSSIAN_WIDE: t = -0.1 * phi; break;
    case IRBFN_GAUSSIAN_WIDER: t = -0.01 * phi; break;
    case IRBFN_INVERSE_QUADRATIC: t = -(phi * phi); break;
    case IRBFN_INVERSE_MULTIQUADRIC: t = -0.5 * phi * phi * phi; break;
    case IRBFN_MULTIQUADRIC: t = 0.5 / phi; break;
    case IRBFN_QUADRATIC: t = 1.0; break;
    default: {
      const double d = sqrt(d2);
      double dp;
      switch (basis) {
        case IRBFN_LINEAR: dp = 1.0; break;
        case IRBFN_SPLINE: dp = 2.0 * d * log(d + 1.0) + d2 / (d + 1.0); break;
        case IRBFN_POISSON_ONE: dp = (2.0 - d) * exp(-d); break;
        case IRBFN_POISSON_TWO: dp = (2.0 * d - 1.0 - 0.5 * d2) * exp(-d); break;
        case IRBFN_MATERN32: dp = -3.0 * d * exp(-1.7320508075688772 * d); break;
        case IRBFN_MATERN52: dp = -(5.0 / 3.0) * d * (1.0 + 2.23606797749979 * d) * exp(-2.23606797749979 * d); break;
        default: dp = 0.0; break;
      }
      ls_term = 0.5 * dp * d / s2;               // width path does not go through the sqrt: finite at d = 0
      return dp * (0.5 / d);                     // centre path: NaN on a centre, like jax.grad of (sum sq) ** 0.5
    }
  }
  ls_term = t * r2;
  return t;
}

__device__ __forceinline__ double gate_factor_f64(double xv, double lo, double hi, double delta) {   // model.py:83-85
  return ((tanh(delta * (xv - lo)) + 1.0) / 2.0) * ((tanh(delta * (hi - xv)) + 1.0) / 2.0);
}

// gamma[B][R] (model.py:42-95): one lane per query, the per-(dimension, range) factors in an LDS column
__global__ __launch_bounds__(64) void f64_gate_kernel(const F64Card c, const double* __restrict__ x, double* __restrict__ gamma, long B) {
  extern __shared__ double ft[];                 // [nsplit * max_ranges][64]
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  const long bb = b < B ? b : B - 1;
  const int E = c.nsplit * c.max_ranges;
  for (int e = 0; e < E; ++e) {
    const int d = e / c.max_ranges;
    ft[e * kWave + lane] = gate_factor_f64(x[bb * c.D + d], c.lo[e], c.hi[e], c.delta[d]);
  }
  if (b >= B) return;
  for (int r = 0; r < c.R; ++r) {
    double g = 0.0;                              // regions without a range stay 0 (model.py:70)
    if (r < c.n_ranges) {
      g = 1.0;
      for (int d = 0; d < c.nsplit; ++d) g *= ft[(d * c.max_ranges + c.dim_ranges[r * c.nsplit + d]) * kWave + lane];
    }
    gamma[b * c.R + r] = g;
  }
}

// forward: one lane per query, centres / widths / weights wave-uniform (scalar loads); outputs in tiles of 16
constexpr int kF64OT = 16;
__global__ __launch_bounds__(64) void f64_forward_kernel(const F64Card c, const double* __restrict__ centers,
                                                         const double* __restrict__ log_sigs, const double* __restrict__ kernel,
                                                         const double* __restrict__ bias, const double* __restrict__ x,
                                                         const double* __restrict__ gamma, double* __restrict__ out, long B) {
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  const long bb = b < B ? b : B - 1;
  double xq[kMaxD];
#pragma unroll
  for (int j = 0; j < kMaxD; ++j) xq[j] = j < c.D ? x[bb * c.D + j] : 0.0;
  for (int o0 = 0; o0 < c.O; o0 += kF64OT) {
    double acc[kF64OT];
#pragma unroll
    for (int o = 0; o < kF64OT; ++o) acc[o] = 0.0;
    for (int r = 0; r < c.R; ++r) {
      const double g = gamma[bb * c.R + r];
      for (int k = 0; k < c.K; ++k) {
        const int n = r * c.K + k;
        const double* cp = centers + (size_t)n * c.D;
        double r2 = 0.0;
#pragma unroll
        for (int j = 0; j < kMaxD; ++j) {
          if (j < c.D) {
            const double df = xq[j] - cp[j];     // flax_rbf.py:280
            r2 = fma(df, df, r2);
          }
        }
        const double sg = exp(log_sigs[n]);
        const double dd = sqrt(r2) / sg;         // d = sqrt(sum sq) / exp(log_sig)
        const double pg = basis_f64(dd * dd, c.basis) * g;              // model.py:193
        const double* wr = kernel + (size_t)k * c.O + o0;
#pragma unroll
        for (int o = 0; o < kF64OT; ++o)
          if (o0 + o < c.O) acc[o] = fma(pg, wr[o], acc[o]);           // model.py:196
      }
    }
    if (b < B) {
#pragma unroll
      for (int o = 0; o < kF64OT; ++o)
        if (o0 + o < c.O) out[b * c.O + o0 + o] = acc[o] + bias[o0 + o];
    }
  }
}

// parameter VJP: one lane per centre, the block's query slice streams past it (wave-uniform rows); O <= 16
struct F64VjpArgs {
  const double* centers; const double* log_sigs; const double* kernel;
  const double* x; const double* g; const double* gamma;
  double* part;                                  // [slices][V][Npad]
  long B;
  int per_slice, Npad;
};
__global__ __launch_bounds__(64) void f64_vjp_kernel(const F64Card c, const F64VjpArgs a) {
  const int lane = threadIdx.x;
  const int N = c.R * c.K;
  const int n = blockIdx.x * kWave + lane;
  const int nn = n < N ? n : N - 1;
  const int r = nn / c.K, k = nn - r * c.K;
  double cc[kMaxD], w[kF64OT];
#pragma unroll
  for (int j = 0; j < kMaxD; ++j) cc[j] = j < c.D ? a.centers[(size_t)nn * c.D + j] : 0.0;
#pragma unroll
  for (int o = 0; o < kF64OT; ++o) w[o] = o < c.O ? a.kernel[(size_t)k * c.O + o] : 0.0;
  const double sg = exp(a.log_sigs[nn]);
  const double s2 = 1.0 / (sg * sg);
  double gc[kMaxD], gw[kF64OT], gls = 0.0;
#pragma unroll
  for (int j = 0; j < kMaxD; ++j) gc[j] = 0.0;
#pragma unroll
  for (int o = 0; o < kF64OT; ++o) gw[o] = 0.0;
  const long b0 = (long)blockIdx.y * a.per_slice;
  long b1 = b0 + a.per_slice;
  b1 = b1 < a.B ? b1 : a.B;
  for (long b = b0; b < b1; ++b) {
    const double* xr = a.x + b * c.D;
    const double* gr = a.g + b * c.O;
    double diff[kMaxD], r2 = 0.0;
#pragma unroll
    for (int j = 0; j < kMaxD; ++j) {
      diff[j] = j < c.D ? xr[j] - cc[j] : 0.0;
      r2 = fma(diff[j], diff[j], r2);
    }
    const double dd = sqrt(r2) / sg;
    const double d2 = dd * dd;
    const double phi = basis_f64(d2, c.basis);
    const double gam = a.gamma[b * c.R + r];
    double hb = 0.0;
    const double gphi = gam * phi;
#pragma unroll
    for (int o = 0; o < kF64OT; ++o) {
      if (o < c.O) {
        hb = fma(gr[o], w[o], hb);
        gw[o] = fma(gphi, gr[o], gw[o]);
      }
    }
    double lsf;
    const double hg = hb * gam;
    const double t = hg * dphi_dd2_f64(phi, d2, r2, s2, c.basis, lsf);
    gls = fma(hg, lsf, gls);
#pragma unroll
    for (int j = 0; j < kMaxD; ++j) gc[j] = fma(t, diff[j], gc[j]);
  }
  if (n < N) {
    const int V = c.D + 1 + c.O;
    double* dst = a.part + (size_t)blockIdx.y * V * a.Npad + n;
    const double m2s = -2.0 * s2;
#pragma unroll
    for (int j = 0; j < kMaxD; ++j)
      if (j < c.D) dst[(size_t)j * a.Npad] = gc[j] * m2s;
    dst[(size_t)c.D * a.Npad] = gls * m2s;
#pragma unroll
    for (int o = 0; o < kF64OT; ++o)
      if (o < c.O) dst[(size_t)(c.D + 1 + o) * a.Npad] = gw[o];
  }
}

// slices (and, for the Dense kernel, regions) summed in a fixed order
__global__ __launch_bounds__(256) void f64_vjp_reduce_kernel(const F64Card c, const double* __restrict__ part, int slices, int Npad,
                                                             double* __restrict__ g_centers, double* __restrict__ g_log_sigs,
                                                             double* __restrict__ g_kernel) {
  const int N = c.R * c.K, V = c.D + 1 + c.O;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n_c = (long)N * c.D, n_k = (long)c.K * c.O;
  if (i < n_c) {
    const int n = (int)(i / c.D), j = (int)(i - (long)n * c.D);
    double s = 0.0;
    for (int q = 0; q < slices; ++q) s += part[((size_t)q * V + j) * Npad + n];
    g_centers[i] = s;
  } else if (i < n_c + N) {
    const int n = (int)(i - n_c);
    double s = 0.0;
    for (int q = 0; q < slices; ++q) s += part[((size_t)q * V + c.D) * Npad + n];
    g_log_sigs[n] = s;
  } else if (i < n_c + N + n_k) {
    const long e = i - n_c - N;
    const int k = (int)(e / c.O), o = (int)(e - (long)k * c.O);
    double s = 0.0;
    for (int r = 0; r < c.R; ++r)
      for (int q = 0; q < slices; ++q) s += part[((size_t)q * V + c.D + 1 + o) * Npad + (size_t)r * c.K + k];
    g_kernel[e] = s;
  }
}

__global__ __launch_bounds__(256) void f64_colsum_kernel(const double* __restrict__ g, double* __restrict__ g_bias, long B, int O) {
  __shared__ double sm[256];
  const int o = blockIdx.x, t = threadIdx.x;
  double s = 0.0;
  for (long b = t; b < B; b += 256) s += g[b * O + o];
  sm[t] = s;
  __syncthreads();
  for (int w2 = 128; w2 > 0; w2 >>= 1) {
    if (t < w2) sm[t] += sm[t + w2];
    __syncthreads();
  }
  if (t == 0) g_bias[o] = sm[0];
}

static int f64_slices(const F64Card& c, int64_t B) {
  const long groups = ((long)c.R * c.K + kWave - 1) / kWave;
  long sl = (4096 + groups - 1) / groups;         // enough waves for the chip
  const long maxs = (B + 63) / 64;
  sl = sl > maxs ? maxs : sl;
  return (int)(sl < 1 ? 1 : (sl > 1024 ? 1024 : sl));
}

static bool f64_card_ok(const irbfn_f64_card* card) {
  return card && card->D >= 1 && card->D <= kMaxD && card->R >= 1 && card->K >= 1 && card->O >= 1 && card->nsplit >= 0 &&
         card->nsplit <= card->D && card->nsplit <= kMaxSplit && card->max_ranges >= (card->nsplit > 0 ? 1 : 0) && card->n_ranges >= 0 &&
         card->basis >= IRBFN_GAUSSIAN && card->basis <= IRBFN_MATERN52 &&
         (card->nsplit == 0 || (card->lo_dev && card->hi_dev && card->delta_dev)) &&
         (card->nsplit == 0 || card->n_ranges == 0 || card->dim_ranges_dev);
}

static F64Card f64_card(const irbfn_f64_card* card) {
  F64Card c;
  c.D = card->D; c.R = card->R; c.K = card->K; c.O = card->O; c.basis = card->basis; c.nsplit = card->nsplit;
  c.max_ranges = card->max_ranges > 0 ? card->max_ranges : 1;
  c.n_ranges = card->n_ranges < card->R ? card->n_ranges : card->R;        // .at[:, i].set beyond num_regions is dropped
  c.lo = card->lo_dev; c.hi = card->hi_dev; c.delta = card->delta_dev; c.dim_ranges = card->dim_ranges_dev;
  return c;
}

}  // namespace irbfn

using namespace irbfn;

extern "C" {

int64_t irbfn_f64_workspace_bytes(const irbfn_f64_card* card, int64_t B, int with_vjp) {
  if (!f64_card_ok(card) || B < 0) return IRBFN_ERR_BAD_ARG;
  const F64Card c = f64_card(card);
  int64_t bytes = ((int64_t)B * c.R * 8 + 255) & ~(int64_t)255;           // gamma[B][R]
  if (with_vjp) {
    const int64_t Npad = (((int64_t)c.R * c.K + kWave - 1) / kWave) * kWave;
    bytes += (int64_t)f64_slices(c, B) * (c.D + 1 + c.O) * Npad * 8;
  }
  return bytes;
}

int irbfn_f64_forward(const irbfn_f64_card* card, const double* centers_dev, const double* log_sigs_dev, const double* kernel_dev,
                      const double* bias_dev, const double* x_dev, double* out_dev, int64_t B, void* workspace_dev,
                      int64_t workspace_bytes, void* stream) {
  if (!f64_card_ok(card) || B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!centers_dev || !log_sigs_dev || !kernel_dev || !bias_dev || !x_dev || !out_dev || !workspace_dev) return IRBFN_ERR_BAD_ARG;
  if (workspace_bytes < irbfn_f64_workspace_bytes(card, B, 0)) return IRBFN_ERR_BAD_ARG;
  const F64Card c = f64_card(card);
  const size_t glds = (size_t)(c.nsplit > 0 ? c.nsplit : 1) * c.max_ranges * kWave * sizeof(double);
  if (glds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  double* gamma = static_cast<double*>(workspace_dev);
  const unsigned grid = (unsigned)((B + kWave - 1) / kWave);
  hipLaunchKernelGGL(f64_gate_kernel, dim3(grid), dim3(kWave), glds, s, c, x_dev, gamma, (long)B);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(f64_forward_kernel, dim3(grid), dim3(kWave), 0, s, c, centers_dev, log_sigs_dev, kernel_dev, bias_dev, x_dev,
                     gamma, out_dev, (long)B);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_f64_vjp(const irbfn_f64_card* card, const double* centers_dev, const double* log_sigs_dev, const double* kernel_dev,
                  const double* x_dev, const double* gout_dev, double* g_centers_dev, double* g_log_sigs_dev, double* g_kernel_dev,
                  double* g_bias_dev, int64_t B, void* workspace_dev, int64_t workspace_bytes, void* stream) {
  if (!f64_card_ok(card) || B < 0 || !g_centers_dev || !g_log_sigs_dev || !g_kernel_dev || !g_bias_dev) return IRBFN_ERR_BAD_ARG;
  const F64Card c = f64_card(card);
  if (c.O > kF64OT) return IRBFN_ERR_UNSUPPORTED;                         // the reference's nets: O = 2, 5, 10
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long N = (long)c.R * c.K;
  if (B == 0) {
    IRBFN_HIP_CHECK(hipMemsetAsync(g_centers_dev, 0, (size_t)N * c.D * 8, s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_log_sigs_dev, 0, (size_t)N * 8, s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_kernel_dev, 0, (size_t)c.K * c.O * 8, s));
    IRBFN_HIP_CHECK(hipMemsetAsync(g_bias_dev, 0, (size_t)c.O * 8, s));
    return IRBFN_OK;
  }
  if (!centers_dev || !log_sigs_dev || !kernel_dev || !x_dev || !gout_dev || !workspace_dev) return IRBFN_ERR_BAD_ARG;
  if (workspace_bytes < irbfn_f64_workspace_bytes(card, B, 1)) return IRBFN_ERR_BAD_ARG;
  const size_t glds = (size_t)(c.nsplit > 0 ? c.nsplit : 1) * c.max_ranges * kWave * sizeof(double);
  if (glds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  char* base = static_cast<char*>(workspace_dev);
  double* gamma = reinterpret_cast<double*>(base);
  double* part = reinterpret_cast<double*>(base + (((size_t)B * c.R * 8 + 255) & ~(size_t)255));
  hipLaunchKernelGGL(f64_gate_kernel, dim3((unsigned)((B + kWave - 1) / kWave)), dim3(kWave), glds, s, c, x_dev, gamma, (long)B);
  IRBFN_HIP_CHECK(hipGetLastError());
  const int slices = f64_slices(c, B);
  F64VjpArgs a;
  a.centers = centers_dev; a.log_sigs = log_sigs_dev; a.kernel = kernel_dev; a.x = x_dev; a.g = gout_dev; a.gamma = gamma;
  a.part = part; a.B = (long)B; a.per_slice = (int)((B + slices - 1) / slices);
  a.Npad = (int)(((N + kWave - 1) / kWave) * kWave);
  hipLaunchKernelGGL(f64_vjp_kernel, dim3((unsigned)(a.Npad / kWave), slices), dim3(kWave), 0, s, c, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  const long total = N * c.D + N + (long)c.K * c.O;
  hipLaunchKernelGGL(f64_vjp_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, c, part, slices, a.Npad,
                     g_centers_dev, g_log_sigs_dev, g_kernel_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(f64_colsum_kernel, dim3(c.O), dim3(256), 0, s, gout_dev, g_bias_dev, (long)B, c.O);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

}  // extern "C"
