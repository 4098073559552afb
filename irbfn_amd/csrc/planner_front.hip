// Batched planner front / back end: what surrounds pred_step in the reference planners.
//   * query construction + mirror trick of IRBFNPlanner.plan (src/irbfn_mpc/irbfn_planner.py:181-208) and
//     IRBFNFrenetPlanner.plan (:456-492), one lane per (pose, goal) pair, float64 like the NumPy host
//     code it replaces, float32 out (jnp.array(...) under the default x64-off config);
//   * the sign flip of the steer-velocity controls of mirrored rows (:203-204, :487-488);
//   * the explicit-MPC table look-ups the learned network is benchmarked against:
//     per-axis searchsorted grid lookup (src/irbfn_mpc/explicit_planner.py:165-175) and the exact nearest
//     neighbour (scipy KDTree.query at :383; brute force here -- HBM-bound streaming of the table).
#include <math.h>
#include <string.h>

#include "common.h"

namespace irbfn {

// Python's float modulo: result has the sign of the divisor (b > 0 here)
__device__ __forceinline__ double py_mod(double a, double b) {
  double r = fmod(a, b);
  if (r != 0.0 && (r < 0.0) != (b < 0.0)) r += b;
  return r;
}

__global__ __launch_bounds__(256) void plan_queries_cartesian_kernel(const double* __restrict__ pose,
                                                                     const double* __restrict__ goal,
                                                                     float* __restrict__ x_out,
                                                                     float* __restrict__ state0_out,
                                                                     int* __restrict__ mirror_out, long B) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* p = pose + b * 7;                // [x, y, delta, v, theta, angv, beta]  (:240)
  const double* g = goal + b * 4;                // ref_point [x, y, theta, v]            (:170-171)
  const double x = p[0], y = p[1], v = p[3], theta = p[4], angv = p[5], beta = p[6];
  const double c = cos(-theta), s = sin(-theta); // rot = [[c, -s], [s, c]]               (:181-183)
  const double dx = g[0] - x, dy = g[1] - y;
  const double gl0 = c * dx + (-s) * dy;         // np.dot(rot, d): products then one add, no FMA
  const double gl1 = s * dx + c * dy;
  const double gt = g[2] - theta;
  const bool m = gl1 < 0.0;                      // goal_needs_mirror                     (:188)
  const double pi = 3.141592653589793;
  float* xo = x_out + b * 7;                     // [v, x_g, y_g, t_g, v_g, beta, angv]   (:189-201)
  xo[0] = (float)v;
  xo[1] = (float)gl0;
  xo[2] = (float)(m ? -gl1 : gl1);
  xo[3] = (float)(m ? py_mod(-gt, pi) : py_mod(gt, pi));
  xo[4] = (float)g[3];
  xo[5] = (float)beta;
  xo[6] = (float)angv;
  if (state0_out) {
#pragma unroll
    for (int i = 0; i < 7; ++i) state0_out[b * 7 + i] = (float)p[i];
  }
  mirror_out[b] = m ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_queries_frenet_kernel(const double* __restrict__ fr,
                                                                  const double* __restrict__ vx_goal,
                                                                  float* __restrict__ x_out,
                                                                  float* __restrict__ state0_out,
                                                                  int* __restrict__ mirror_out, long B) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* p = fr + b * 8;                  // [s, ey, delta, vx, vy, wz, epsi, curv]  (:491-502)
  const bool m = p[1] < -0.05;                   // goal_needs_mirror = ey < -0.05           (:457)
  float* xo = x_out + b * 8;                     // [ey, delta, vx, vy, vx_goal, wz, epsi, curv]  (:459-478)
  xo[0] = (float)(m ? -p[1] : p[1]);
  xo[1] = (float)p[2];
  xo[2] = (float)p[3];
  xo[3] = (float)(m ? -p[4] : p[4]);
  xo[4] = (float)vx_goal[b];
  xo[5] = (float)(m ? -p[5] : p[5]);
  xo[6] = (float)(m ? -p[6] : p[6]);
  xo[7] = (float)p[7];
  if (state0_out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) state0_out[b * 8 + i] = (float)p[i];
  }
  mirror_out[b] = m ? 1 : 0;
}

__global__ __launch_bounds__(256) void unmirror_kernel(float* __restrict__ controls, const int* __restrict__ mirror,
                                                       long B, int O, int sv0) {
  const int w = O - sv0;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * w) return;
  const long b = i / w;
  const int o = sv0 + (int)(i - b * w);
  if (mirror[b] != 0) controls[b * O + o] = -controls[b * O + o];
}

int launch_unmirror(float* controls, const int* mirror, int64_t B, int O, int sv0, hipStream_t s) {
  const long n = (long)B * (O - sv0);
  if (n <= 0) return IRBFN_OK;
  hipLaunchKernelGGL(unmirror_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, controls, mirror, (long)B, O,
                     sv0);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// ---- grid look-up: idx_d = min(shape_d - 1, searchsorted(keys_d, x_d, side="right"))   explicit_planner.py:165-172
constexpr int kLutMaxD = 16;
struct LutGrid {
  int off[kLutMaxD + 1];     // keys of axis d are keys[off[d] .. off[d+1])
  int shape[kLutMaxD];       // table shape per axis (row-major, last axis fastest)
};

__global__ __launch_bounds__(256) void lut_grid_lookup_kernel(const double* __restrict__ keys, const LutGrid g,
                                                              const float* __restrict__ table,
                                                              const double* __restrict__ x, long* __restrict__ idx_out,
                                                              float* __restrict__ out, long B, int D, int OW) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long flat = 0;
  for (int d = 0; d < D; ++d) {
    const double v = x[b * D + d];
    int lo = g.off[d], hi = g.off[d + 1];         // first index with keys[i] > v  (side="right")
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] <= v) lo = mid + 1;           // NaN compares false -> lo stays: searchsorted sorts NaN last
      else hi = mid;
    }
    int i = lo - g.off[d];
    if (v != v) i = g.off[d + 1] - g.off[d];      // NaN -> len(keys), as numpy
    if (i > g.shape[d] - 1) i = g.shape[d] - 1;
    flat = flat * g.shape[d] + i;
  }
  idx_out[b] = flat;
  if (out) {
    const float* row = table + flat * OW;
    for (int o = 0; o < OW; ++o) out[b * OW + o] = row[o];
  }
}

// ---- exact nearest neighbour: argmin_n ||inputs[n] - x_b||  (ties -> lowest n)
// One lane per table row, grid-stride; the QT queries of a tile are wave-uniform (scalar loads); per-block
// (d2, n) minima go to a [B][nblocks] slab, a second kernel takes the fixed-order minimum: deterministic.
template <int D, int QT>
__global__ __launch_bounds__(256) void lut_nearest_kernel(const float* __restrict__ inputs, const float* __restrict__ x,
                                                          float* __restrict__ part_d, long* __restrict__ part_i, long N,
                                                          long B, int nblocks) {
  __shared__ float sd[4][QT];
  __shared__ long si[4][QT];
  const long q0 = (long)blockIdx.y * QT;
  float best[QT];
  long bidx[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) { best[q] = INFINITY; bidx[q] = 0x7fffffffffffffffL; }
  for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long)gridDim.x * blockDim.x) {
    float r[D];
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = inputs[n * D + d];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const long qb = q0 + q < B ? q0 + q : B - 1;
      float d2 = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) { const float t = r[d] - x[qb * D + d]; d2 = __builtin_fmaf(t, t, d2); }
      if (d2 < best[q]) { best[q] = d2; bidx[q] = n; }     // rows ascend per lane: strict < keeps the lowest n
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    float bd = best[q];
    long bi = bidx[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float od = __shfl_xor(bd, off);
      const long oi = __shfl_xor(bi, off);
      if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) { sd[wave][q] = bd; si[wave][q] = bi; }
  }
  __syncthreads();
  if (threadIdx.x < QT && q0 + threadIdx.x < B) {
    const int q = threadIdx.x;
    float bd = sd[0][q];
    long bi = si[0][q];
    for (int w = 1; w < 4; ++w)
      if (sd[w][q] < bd || (sd[w][q] == bd && si[w][q] < bi)) { bd = sd[w][q]; bi = si[w][q]; }
    part_d[(q0 + q) * nblocks + blockIdx.x] = bd;
    part_i[(q0 + q) * nblocks + blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(64) void lut_nearest_final_kernel(const float* __restrict__ part_d,
                                                               const long* __restrict__ part_i, int nblocks,
                                                               const float* __restrict__ table, long* __restrict__ idx_out,
                                                               float* __restrict__ dist_out, float* __restrict__ out,
                                                               int OW) {
  const long b = blockIdx.x;
  const int lane = threadIdx.x;
  float bd = INFINITY;
  long bi = 0x7fffffffffffffffL;
  for (int k = lane; k < nblocks; k += 64) {
    const float d = part_d[b * nblocks + k];
    const long i = part_i[b * nblocks + k];
    if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float od = __shfl_xor(bd, off);
    const long oi = __shfl_xor(bi, off);
    if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
  }
  if (lane == 0) {
    idx_out[b] = bi;
    if (dist_out) dist_out[b] = sqrtf(bd);
  }
  if (out && table && bi != 0x7fffffffffffffffL)
    for (int o = lane; o < OW; o += 64) out[b * OW + o] = table[bi * OW + o];
}

template <int D>
static int nearest_launch(const float* inputs, const float* x, float* pd, long* pi, long N, long B, int nblocks,
                          hipStream_t s) {
  constexpr int QT = 8;
  hipLaunchKernelGGL((lut_nearest_kernel<D, QT>), dim3((unsigned)nblocks, (unsigned)((B + QT - 1) / QT)), dim3(256), 0, s,
                     inputs, x, pd, pi, N, B, nblocks);
  return IRBFN_OK;
}

}  // namespace irbfn

using namespace irbfn;

// ---- way-point geometry of the pure-pursuit front end (src/irbfn_mpc/planner_utils.py:109-240) -----------------
// The reference runs nearest_point / intersect_point once per planner tick under numba; here they are batched over B
// query points against ONE piecewise-linear trajectory (staged in LDS), one lane per point.  The arithmetic types
// follow the reference line by line: float32 segment vectors / squared lengths / dot products, float64 everything
// else in nearest_point (:125-141); float32 trajectory, +1e-6f end points, float32 t in intersect_point (:160-170),
// with the point kept in float64 as the callers pass it.  numba's typing of the mixed expressions cannot be
// checked here (numba is not importable): parity unpinned, tested against the NumPy statement of the same lines.
struct WayArgs {
  const double* __restrict__ pts;    // [B][2]
  const double* __restrict__ traj;   // [N][2]
  long B;
  int N;
};

// One WAVE per point: the lanes take the segments lane, lane + 64, ... and the wave picks the smallest distance (the
// smallest segment index among equal distances: np.argmin's first minimum).  (One thread per point walking all N
// segments with a float64 divide and square root each took 200 us at ANY batch size -- the planner calls this with one
// point per tick.)
__global__ __launch_bounds__(256) void nearest_point_kernel(const WayArgs a, double* __restrict__ proj,
                                                            double* __restrict__ dist, double* __restrict__ tt,
                                                            int* __restrict__ seg) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;                          // the whole wave
  const double* __restrict__ wp = a.traj;
  const double px = a.pts[2 * b], py = a.pts[2 * b + 1];
  double best = INFINITY, bt = 0.0, bx = 0.0, by = 0.0;
  int bi = 0x7fffffff;
  for (int i = lane; i + 1 < a.N; i += 64) {
    const double x0 = wp[2 * i], y0 = wp[2 * i + 1];
    const float dx = (float)(wp[2 * i + 2] - x0), dy = (float)(wp[2 * i + 3] - y0);       // diffs.astype(float32) :125
    const float l2 = dx * dx + dy * dy;                                                  // :126
    const float lx = (float)(px - x0), ly = (float)(py - y0);                            // :129
    const float dot = lx * dx + ly * dy;                                                 // np.dot of two float32 pairs :130
    double t = (double)dot / (double)l2;                                                 // :131
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);                                             // :132-133 (nan stays)
    const double qx = x0 + t * (double)dx, qy = y0 + t * (double)dy;                     // :134
    const double ex = px - qx, ey = py - qy;
    const double d = sqrt(ex * ex + ey * ey);                                            // :137-138
    if (d < best) { best = d; bt = t; bx = qx; by = qy; bi = i; }                        // first minimum of this lane's segments
  }
  double gbest = best;
  int gbi = bi;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double od = __shfl_xor(gbest, off);
    const int oi = __shfl_xor(gbi, off);
    if (od < gbest || (od == gbest && oi < gbi)) { gbest = od; gbi = oi; }
  }
  if (gbi == 0x7fffffff) {                       // no segment compared below infinity (all NaN): the scalar loop's defaults
    if (lane == 0) { proj[2 * b] = 0.0; proj[2 * b + 1] = 0.0; dist[b] = INFINITY; tt[b] = 0.0; seg[b] = 0; }
  } else if (bi == gbi) {
    proj[2 * b] = bx; proj[2 * b + 1] = by;
    dist[b] = best; tt[b] = bt; seg[b] = bi;
  }
}

__device__ __forceinline__ bool circle_hit(const double* wp, int N, int i, double px, double py, float radius,
                                           bool first, float start_t, float& t_out, float& qx, float& qy) {
  const int i0 = ((i % N) + N) % N, i1 = (((i + 1) % N) + N) % N;
  const float sx = (float)wp[2 * i0], sy = (float)wp[2 * i0 + 1];                         // trajectory.astype(float32) :160
  const float ex = (float)wp[2 * i1] + 1e-6f, ey = (float)wp[2 * i1 + 1] + 1e-6f;         // :163
  const float vx = ex - sx, vy = ey - sy;
  const float av = vx * vx + vy * vy;                                                    // :166
  const double bq = 2.0 * ((double)vx * ((double)sx - px) + (double)vy * ((double)sy - py));   // :167
  const double cq = (double)(sx * sx + sy * sy) + (px * px + py * py) - 2.0 * ((double)sx * px + (double)sy * py) -
                    (double)radius * (double)radius;                                     // :168-173
  double disc = bq * bq - 4.0 * (double)av * cq;                                         // :174
  if (disc < 0.0 || disc != disc) return false;
  disc = sqrt(disc);
  const double t1 = (-bq - disc) / (2.0 * (double)av), t2 = (-bq + disc) / (2.0 * (double)av);   // :182-183
  double t;
  if (first) {                                                                           // :184-193
    if (t1 >= 0.0 && t1 <= 1.0 && t1 >= (double)start_t) t = t1;
    else if (t2 >= 0.0 && t2 <= 1.0 && t2 >= (double)start_t) t = t2;
    else return false;
  } else {                                                                               // :194-203
    if (t1 >= 0.0 && t1 <= 1.0) t = t1;
    else if (t2 >= 0.0 && t2 <= 1.0) t = t2;
    else return false;
  }
  t_out = (float)t;
  qx = (float)((double)sx + t * (double)vx);
  qy = (float)((double)sy + t * (double)vy);
  return true;
}

// One wave per point: the search order of the reference (segments start_i .. N - 2, then, with wrap, -1 .. start_i - 1) is
// walked 64 segments at a time; the first round with a hit ends the search and its lowest lane is the first intersection.
__global__ __launch_bounds__(256) void intersect_point_kernel(const WayArgs a, const double* __restrict__ t0, float radius,
                                                              int wrap, float* __restrict__ first_p, int* __restrict__ first_i,
                                                              float* __restrict__ first_t, int* __restrict__ found) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;                          // the whole wave
  const double* __restrict__ wp = a.traj;
  const double px = a.pts[2 * b], py = a.pts[2 * b + 1];
  const double ts = t0 ? t0[b] : 0.0;
  const int start_i = (int)ts;                                                           // :155
  const float start_t = (float)fmod(ts, 1.0);                                            // :156
  const int nfwd = a.N - 1 - start_i > 0 ? a.N - 1 - start_i : 0;                        // i = start_i .. N - 2
  const int nwrap = wrap ? (start_i + 1 > 0 ? start_i + 1 : 0) : 0;                      // i = -1 .. start_i - 1   :205-231
  float t = NAN, qx = NAN, qy = NAN;                                                     // (None, None, None)
  int fi = 0, hit = 0;
  for (int base = 0; base < nfwd + nwrap; base += 64) {
    const int ord = base + lane;
    bool h = false;
    int i = 0;
    if (ord < nfwd) {
      i = start_i + ord;
      h = circle_hit(wp, a.N, i, px, py, radius, ord == 0, start_t, t, qx, qy);
    } else if (ord < nfwd + nwrap) {
      i = ord - nfwd - 1;
      h = circle_hit(wp, a.N, i, px, py, radius, false, 0.0f, t, qx, qy);
    }
    const unsigned long long m = __ballot(h);
    if (m != 0ull) {
      const int src = __ffsll((long long)m) - 1;                                         // the first hit in search order
      t = __shfl(t, src); qx = __shfl(qx, src); qy = __shfl(qy, src); fi = __shfl(i, src);
      hit = 1;
      break;
    }
    t = NAN; qx = NAN; qy = NAN;
  }
  if (lane == 0) {
    found[b] = hit;
    first_i[b] = fi;
    first_t[b] = t;
    first_p[2 * b] = qx;
    first_p[2 * b + 1] = qy;
  }
}

extern "C" {

int irbfn_plan_queries_cartesian(const double* pose_dev, const double* goal_dev, float* x_dev, float* state0_dev,
                                 int32_t* mirror_dev, int64_t B, void* stream) {
  if (B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!pose_dev || !goal_dev || !x_dev || !mirror_dev) return IRBFN_ERR_BAD_ARG;
  hipLaunchKernelGGL(plan_queries_cartesian_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), pose_dev, goal_dev, x_dev, state0_dev, mirror_dev, (long)B);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_plan_queries_frenet(const double* frenet_dev, const double* vx_goal_dev, float* x_dev, float* state0_dev,
                              int32_t* mirror_dev, int64_t B, void* stream) {
  if (B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!frenet_dev || !vx_goal_dev || !x_dev || !mirror_dev) return IRBFN_ERR_BAD_ARG;
  hipLaunchKernelGGL(plan_queries_frenet_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), frenet_dev, vx_goal_dev, x_dev, state0_dev, mirror_dev,
                     (long)B);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_lut_grid_lookup(const double* keys_dev, const int32_t* key_offsets_host, const int32_t* shape_host,
                          const float* table_dev, const double* x_dev, int64_t* idx_dev, float* out_dev, int64_t B,
                          int D, int OW, void* stream) {
  if (B < 0 || D < 1 || D > kLutMaxD || OW < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!keys_dev || !key_offsets_host || !shape_host || !x_dev || !idx_dev || (out_dev && !table_dev))
    return IRBFN_ERR_BAD_ARG;
  LutGrid g;
  memset(&g, 0, sizeof(g));
  for (int d = 0; d <= D; ++d) g.off[d] = key_offsets_host[d];
  for (int d = 0; d < D; ++d) {
    g.shape[d] = shape_host[d];
    if (g.shape[d] < 1 || g.off[d + 1] < g.off[d]) return IRBFN_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(lut_grid_lookup_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), keys_dev, g, table_dev, x_dev,
                     reinterpret_cast<long*>(idx_dev), out_dev, (long)B, D, OW);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int64_t irbfn_lut_nearest_workspace_bytes(int64_t N, int64_t B) {
  if (N < 0 || B < 0) return -1;
  const int nblocks = 2048;
  return (int64_t)B * nblocks * (int64_t)(sizeof(float) + sizeof(long));
}

int irbfn_lut_nearest(const float* inputs_dev, const float* table_dev, const float* x_dev, int64_t* idx_dev,
                      float* dist_dev, float* out_dev, int64_t N, int64_t B, int D, int OW, void* ws_dev,
                      int64_t ws_bytes, void* stream) {
  if (N < 1 || B < 0 || OW < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!inputs_dev || !x_dev || !idx_dev || !ws_dev || (out_dev && !table_dev)) return IRBFN_ERR_BAD_ARG;
  if (ws_bytes < irbfn_lut_nearest_workspace_bytes(N, B)) return IRBFN_ERR_BAD_ARG;
  const int cap = 2048;
  long want = (N + 255) / 256;
  const int nblocks = (int)(want < cap ? want : cap);
  float* pd = reinterpret_cast<float*>(ws_dev);
  long* pi = reinterpret_cast<long*>(reinterpret_cast<char*>(ws_dev) + (size_t)B * cap * sizeof(float));
  // 8-byte alignment of the index slab: B * cap * 4 is a multiple of 8 because cap is even
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (D) {
    case 3: nearest_launch<3>(inputs_dev, x_dev, pd, pi, N, B, nblocks, s); break;
    case 4: nearest_launch<4>(inputs_dev, x_dev, pd, pi, N, B, nblocks, s); break;
    case 7: nearest_launch<7>(inputs_dev, x_dev, pd, pi, N, B, nblocks, s); break;
    case 8: nearest_launch<8>(inputs_dev, x_dev, pd, pi, N, B, nblocks, s); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(lut_nearest_final_kernel, dim3((unsigned)B), dim3(64), 0, s, pd, pi, nblocks, table_dev,
                     reinterpret_cast<long*>(idx_dev), dist_dev, out_dev, OW);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}


/* planner_utils.py:109-146 and :149-233, batched over B query points against one trajectory (see the kernels). */
int irbfn_nearest_point(const double* points_dev, const double* trajectory_dev, double* proj_dev, double* dist_dev,
                        double* t_dev, int32_t* seg_dev, int64_t B, int N, void* stream) {
  if (B < 0 || N < 2) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!points_dev || !trajectory_dev || !proj_dev || !dist_dev || !t_dev || !seg_dev) return IRBFN_ERR_BAD_ARG;
  WayArgs a{points_dev, trajectory_dev, (long)B, N};
  hipLaunchKernelGGL(nearest_point_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a, proj_dev, dist_dev, t_dev, seg_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_intersect_point(const double* points_dev, const double* trajectory_dev, const double* t_start_dev, float radius,
                          int wrap, float* first_p_dev, int32_t* first_i_dev, float* first_t_dev, int32_t* found_dev,
                          int64_t B, int N, void* stream) {
  if (B < 0 || N < 2) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!points_dev || !trajectory_dev || !first_p_dev || !first_i_dev || !first_t_dev || !found_dev) return IRBFN_ERR_BAD_ARG;
  WayArgs a{points_dev, trajectory_dev, (long)B, N};
  hipLaunchKernelGGL(intersect_point_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a, t_start_dev, radius, wrap, first_p_dev, first_i_dev, first_t_dev, found_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

}  // extern "C"
