// K1 instantiations for ONE compiled input width D = IRBFN_INST_D (compiled once per D so that the
// translation units build in parallel).  Design notes: rbf_forward.h.
#include "rbf_forward.h"
#include "rollout_step.h"

#ifndef IRBFN_INST_D
#error "compile with -DIRBFN_INST_D=<3|4|7|8>"
#endif

namespace irbfn {

template <int D, int OP>
struct RecLayout {
  static constexpr int S = (D + 1 + OP + 3) & ~3;   // floats per record (16-byte multiple)
};

// (query, centre) pair: distance, basis, weight-row FMA.  `rp` is wave-uniform -> SGPR operands.
template <int D, int OP, int Q, int BC, bool GATED>
__device__ __forceinline__ void pair_body(const float* __restrict__ rp_, const float (&xq)[Q][D],
                                          float (&acc)[Q][OP], const float (&g)[Q], int basis) {
  typedef const float __attribute__((address_space(4)))* crec_t;
  const crec_t rp = (crec_t)(uintptr_t)rp_;    // scalar (s_load) path, see group_body
  float r2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) r2[q] = 0.0f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float c = rp[j];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const float d = xq[q][j] - c;            // flax_rbf.py:280  (x_e - centers)
      r2[q] = __builtin_fmaf(d, d, r2[q]);     //                  ** 2 .sum(-1)
    }
  }
  const float sc = rp[D];
  float phi[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    phi[q] = basis_from_r2<BC>(r2[q], sc, basis);
    if constexpr (GATED) phi[q] *= g[q];       // model.py:193  gamma_rep * all_x
  }
#pragma unroll
  for (int o = 0; o < OP; ++o) {
    const float w = rp[D + 1 + o];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q][o] = __builtin_fmaf(phi[q], w, acc[q][o]);   // model.py:196
  }
}

// G centres at once: G distances, then the G*Q transcendentals as ONE block (rbf_forward.h: an isolated
// transcendental costs ~4x a batched one), then the G weight rows.  Fast bases only.
constexpr int kRollStagePitch = 65;   // ROLL: floats per row of the states staging tile (T * S <= 64), odd
#ifndef IRBFN_FWD_G
#define IRBFN_FWD_G 4
#endif
template <int D, int OP, int Q, int BC, bool GATED, int G>
__device__ __forceinline__ void group_body(const float* __restrict__ rp, int S, const float (&xq)[Q][D],
                                           float (&acc)[Q][OP], const float (&g)[Q]) {
  // constant address space: keeps the record reads on the scalar path (s_load); with a volatile asm in the
  // loop hipcc can no longer prove the buffer unclobbered and would fall back to per-lane global_load
  typedef const float __attribute__((address_space(4)))* crec_t;
  float t[G * Q];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const crec_t r = (crec_t)(uintptr_t)(rp + k * S);
    float r2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r2[q] = 0.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float c = r[j];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float d = xq[q][j] - c;
        r2[q] = __builtin_fmaf(d, d, r2[q]);
      }
    }
    const float sc = r[D];
#pragma unroll
    for (int q = 0; q < Q; ++q) t[k * Q + q] = basis_arg<BC>(r2[q], sc);
  }
  trans_block<BC, G * Q>(t);
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const crec_t r = (crec_t)(uintptr_t)(rp + k * S);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      float phi = t[k * Q + q];
      if constexpr (GATED) phi *= g[q];
#pragma unroll
      for (int o = 0; o < OP; ++o) acc[q][o] = __builtin_fmaf(phi, r[D + 1 + o], acc[q][o]);
    }
  }
}

template <int D, int OP, int Q, int BC, bool GATED, bool ROLL>
__global__ __launch_bounds__((OP * Q > 48) ? 512 : 1024) void rbf_fwd_qlane(const FwdArgs a) {
  extern __shared__ float lds[];
  constexpr int ROWS = kWave * Q;
  constexpr int S = RecLayout<D, OP>::S;
  constexpr int OC = OP < 16 ? OP : 16;          // outputs reduced per LDS pass
  constexpr int LP = kWave + 1;                  // padded lane pitch of the reduction buffer

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int nw = nthreads >> 6;
  const long row0 = (long)blockIdx.x * ROWS;
  const long left = a.B - row0;
  const int nvalid = left < ROWS ? (int)left : ROWS;
  const int Dr = a.Dreal;
  const GateTables gt = a.gate;

  // ---- stage the query tile (contiguous in HBM -> coalesced) and pull it into registers
  float* xs = lds;                               // [ROWS][Dr]
  {
    const float* src = a.x + row0 * Dr;
    for (int i = tid; i < nvalid * Dr; i += nthreads) xs[i] = src[i];
  }
  __syncthreads();
  float xq[Q][D];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    int rr = lane + kWave * q;
    rr = rr < nvalid ? rr : nvalid - 1;
#pragma unroll
    for (int j = 0; j < D; ++j) xq[q][j] = j < Dr ? xs[rr * Dr + j] : 0.0f;
  }

  // ---- smooth region gate (model.py:42-95)
  float gam[Q];
  float* gtab = lds + ROWS * Dr;                 // GATED: [nsplit*max_ranges][ROWS]
  if constexpr (!GATED) {
#pragma unroll
    for (int q = 0; q < Q; ++q) gam[q] = 0.0f;
    if (wave == 0) {                             // R == 1: gamma factors out of the k-sum, only wave 0 applies it
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        float g = gt.n_ranges > 0 ? 1.0f : 0.0f; // model.py:70: regions without a range stay 0
#pragma unroll
        for (int d = 0; d < D; ++d) {
          if (d < gt.nsplit && gt.n_ranges > 0) {
            const int e = d * gt.max_ranges + gt.dim_ranges[d];
            g *= gate_factor(xq[q][d], gt.lo[e], gt.hi[e], gt.delta[d]);
          }
        }
        gam[q] = g;
      }
    }
  } else {
    const int E = gt.nsplit * gt.max_ranges;
    for (int idx = tid; idx < E * ROWS; idx += nthreads) {
      const int e = idx / ROWS, row = idx - e * ROWS;
      const int d = e / gt.max_ranges;
      const int rr = row < nvalid ? row : nvalid - 1;
      gtab[idx] = gate_factor(xs[rr * Dr + d], gt.lo[e], gt.hi[e], gt.delta[d]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) gam[q] = 0.0f;
  }

  // ---- hot loop: this wave's slice of the centre records
  float acc[Q][OP];
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int o = 0; o < OP; ++o) acc[q][o] = 0.0f;

  const int per = (a.N + nw - 1) / nw;
  const int n0 = wave * per;
  const int n1 = (n0 + per) < a.N ? (n0 + per) : a.N;
  if constexpr (!GATED) {
    int n = n0;
    constexpr int G = IRBFN_FWD_G;
    if constexpr (BC != BC_GENERIC && G > 1 && (G * Q <= 16)) {
#ifdef IRBFN_DBG_RECMASK      // diagnosis only: every wave re-reads the same few records (scalar-cache hits)
      for (; n + G <= n1; n += G) group_body<D, OP, Q, BC, false, G>(a.rec + (size_t)(n & IRBFN_DBG_RECMASK) * S, S, xq, acc, gam);
#else
      for (; n + G <= n1; n += G) group_body<D, OP, Q, BC, false, G>(a.rec + (size_t)n * S, S, xq, acc, gam);
#endif
    }
#pragma unroll 2
    for (; n < n1; ++n)
      pair_body<D, OP, Q, BC, false>(a.rec + (size_t)n * S, xq, acc, gam, a.basis);
  } else {
    int n = n0;
    while (n < n1) {
      const int r = n / a.K;
      const int nend = ((r + 1) * a.K) < n1 ? ((r + 1) * a.K) : n1;
      bool any = false;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        float g = 0.0f;
        if (a.gamma_ext != nullptr) {            // external gate (ClusterWCRBFNet softmax, model.py:402-408)
          int rr = lane + kWave * q;
          rr = rr < nvalid ? rr : nvalid - 1;
          g = a.gamma_ext[(row0 + rr) * a.R + r];
        } else if (r < gt.n_ranges) {            // model.py:88-93
          g = 1.0f;
          for (int d = 0; d < gt.nsplit; ++d) {
            const int e = d * gt.max_ranges + gt.dim_ranges[r * gt.nsplit + d];
            g *= gtab[e * ROWS + q * kWave + lane];
          }
        }
        gam[q] = g;
        any |= (g != 0.0f);
      }
      if (__ballot(any) == 0ull) {               // no query of this wave is inside region r
        n = nend;
        continue;
      }
      constexpr int G = IRBFN_FWD_G;
      if constexpr (BC != BC_GENERIC && G > 1 && (G * Q <= 16)) {
        for (; n + G <= nend; n += G) group_body<D, OP, Q, BC, true, G>(a.rec + (size_t)n * S, S, xq, acc, gam);
      }
      for (; n < nend; ++n)
        pair_body<D, OP, Q, BC, true>(a.rec + (size_t)n * S, xq, acc, gam, a.basis);
    }
  }

  // ---- combine the NW partial sums (fixed order -> deterministic), apply gamma / bias, store
  __syncthreads();                               // xs / gtab are dead from here on
  float* red = lds;                              // [nw][Q][OC][LP]
  float* grow = red + nw * Q * OC * LP;          // [ROWS] gamma per row (R == 1)
  float* ctrl = grow + ROWS;                     // ROLL: [ROWS][O] controls
  if constexpr (!GATED) {
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < Q; ++q) grow[q * kWave + lane] = gam[q];
    }
  }
#pragma unroll
  for (int o0 = 0; o0 < OP; o0 += OC) {
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int oc = 0; oc < OC; ++oc)
        if (o0 + oc < OP) red[((wave * Q + q) * OC + oc) * LP + lane] = acc[q][o0 + oc];
    __syncthreads();
    int ocn = a.O - o0;
    ocn = ocn < OC ? ocn : OC;
    if (ocn > 0) {
      for (int idx = tid; idx < nvalid * ocn; idx += nthreads) {
        const int row = idx / ocn, oc = idx - row * ocn;
        const int q = row >> 6, l = row & (kWave - 1);
        float s = 0.0f;
        for (int w = 0; w < nw; ++w) s += red[((w * Q + q) * OC + oc) * LP + l];
        float v = GATED ? s : grow[row] * s;     // R == 1: gamma factors out of the k-sum
        v += a.bias[o0 + oc];                    // Dense bias, model.py:196
        if (a.mirror != nullptr && o0 + oc >= a.sv0 && a.mirror[row0 + row] != 0) v = -v;
        if (a.out) a.out[(row0 + row) * a.O + o0 + oc] = v;
        if constexpr (ROLL) ctrl[row * a.O + o0 + oc] = v;
      }
    }
    __syncthreads();
  }

  // ---- fused roll-out: the controls stay in LDS (batched IRBFNPlanner.plan, irbfn_planner.py:205-212); the T x S states
  // of the block's rows -- one contiguous piece of HBM -- are staged row by row in LDS and leave as coalesced dwords
  // (round 1 stored them per lane, 4 bytes at a stride of T x S x 4)
  if constexpr (ROLL) {
    const int T = a.T;
    const int Sdim = (a.mode == IRBFN_ROLLOUT_FULLINT) ? 5 : (a.mode == IRBFN_ROLLOUT_FRENET_LS ? 8 : 7);
    float* stage = ctrl + ROWS * a.O;            // [ROWS][kRollStagePitch]
    for (int row = tid; row < nvalid; row += nthreads) {
      const float* u = ctrl + row * a.O;
      const long b = row0 + row;
      float* o = stage + row * kRollStagePitch;
      if (a.mode == IRBFN_ROLLOUT_ST_SELECT || a.mode == IRBFN_ROLLOUT_ST_KS) {
        float s[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) s[i] = a.state0[b * 7 + i];
        for (int t = 0; t < T; ++t) {
          if (a.mode == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, u[t], u[T + t], a.dp);
          else st_step<false>(s, u[t], u[T + t], a.dp);
#pragma unroll
          for (int i = 0; i < 7; ++i) o[t * 7 + i] = s[i];
        }
      } else if (a.mode == IRBFN_ROLLOUT_FRENET_LS) {
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = a.state0[b * 8 + i];
        for (int t = 0; t < T; ++t) {
          frenet_step(s, u[t], u[T + t], a.dp);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[t * 8 + i] = s[i];
        }
      } else if (a.mode == IRBFN_ROLLOUT_FULLINT) {
        float s[5] = {0.0f, 0.0f, 0.0f, clipf(a.state0[b], 0.0f, 7.0f), 0.0f};   // train_nmpc.py:319
        for (int t = 0; t < T; ++t) {
          fullint_step(s, u[t], u[T + t]);
#pragma unroll
          for (int i = 0; i < 5; ++i) o[t * 5 + i] = s[i];
        }
      }
    }
    __syncthreads();
    const int rowf = T * Sdim;                   // floats per trajectory (<= 64)
    float* gout = a.states + row0 * (long)rowf;
    for (int idx = tid; idx < nvalid * rowf; idx += nthreads) {
      const int r = idx / rowf, c = idx - r * rowf;
      gout[idx] = stage[r * kRollStagePitch + c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host-side dispatch for this D
// ------------------------------------------------------------------------------------------------
template <int D, int OP, int Q, int BC, bool GATED, bool ROLL>
static int launch_one(const FwdArgs& a, int nw, size_t lds_bytes, hipStream_t s, int* grid_out) {
  constexpr int ROWS = kWave * Q;
  const long tiles = (a.B + ROWS - 1) / ROWS;
  auto kern = rbf_fwd_qlane<D, OP, Q, BC, GATED, ROLL>;
  if (lds_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(nw * kWave), lds_bytes, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  *grid_out = (int)tiles;
  return IRBFN_OK;
}

template <int D, int OP, int Q, bool ROLL>
static int launch_bc(const FwdArgs& a, int bc, bool gated, int nw, size_t lds, hipStream_t s, int* g) {
#define IRBFN_CASE(BCV)                                                                   \
  case BCV:                                                                               \
    return gated ? launch_one<D, OP, Q, BCV, true, ROLL>(a, nw, lds, s, g)                \
                 : launch_one<D, OP, Q, BCV, false, ROLL>(a, nw, lds, s, g);
  switch (bc) {
    IRBFN_CASE(BC_GAUSS)
    IRBFN_CASE(BC_IQ)
    IRBFN_CASE(BC_IMQ)
    case BC_GENERIC:
      if constexpr (ROLL) return IRBFN_ERR_UNSUPPORTED;   // fused roll-out: fast bases only
      else if (Q != 1) return IRBFN_ERR_UNSUPPORTED;
      else
        return gated ? launch_one<D, OP, 1, BC_GENERIC, true, false>(a, nw, lds, s, g)
                     : launch_one<D, OP, 1, BC_GENERIC, false, false>(a, nw, lds, s, g);
  }
#undef IRBFN_CASE
  return IRBFN_ERR_UNSUPPORTED;
}

#define IRBFN_CAT2(a, b) a##b
#define IRBFN_CAT(a, b) IRBFN_CAT2(a, b)

// int launch_forward_d<D>(args, OP, Q, bclass, gated, roll, nw, lds_bytes, stream, &grid)
int IRBFN_CAT(launch_forward_d, IRBFN_INST_D)(const FwdArgs& a, int OP, int Q, int bc, bool gated, bool roll,
                                             int nw, size_t lds, hipStream_t s, int* grid_out) {
  constexpr int D = IRBFN_INST_D;
#define IRBFN_OPQ(OPV, QV)                                                      \
  if (OP == OPV && Q == QV) {                                                   \
    return roll ? launch_bc<D, OPV, QV, true>(a, bc, gated, nw, lds, s, grid_out) \
                : launch_bc<D, OPV, QV, false>(a, bc, gated, nw, lds, s, grid_out); \
  }
  IRBFN_OPQ(2, 1)
  IRBFN_OPQ(2, 2)
  IRBFN_OPQ(4, 1)
  IRBFN_OPQ(5, 1)
  IRBFN_OPQ(5, 2)
  IRBFN_OPQ(8, 1)
  IRBFN_OPQ(10, 1)
  IRBFN_OPQ(10, 2)
  IRBFN_OPQ(16, 1)
  IRBFN_OPQ(32, 1)
  IRBFN_OPQ(64, 1)
  IRBFN_OPQ(100, 1)
  IRBFN_OPQ(128, 1)
#undef IRBFN_OPQ
  return IRBFN_ERR_UNSUPPORTED;
}

}  // namespace irbfn
