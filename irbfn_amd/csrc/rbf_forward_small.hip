// K1s: latency kernel for small batches (the closed-loop planner calls the net with B = 1,
// src/irbfn_mpc/irbfn_planner.py:205).  With one query there is nothing to put on 64 query lanes, so
// the mapping flips ("centre-lane"): consecutive lanes own consecutive centres (coalesced reads of the
// packed centre records, 80 B per lane), the query is wave-uniform (SGPRs), every lane evaluates phi
// for its centres and multiplies by its own weight row, and the O partial sums are combined by
// wavefront shuffles, LDS across the 4 waves and -- across workgroups -- a fixed-order sum performed
// by the last workgroup to arrive (agent-scope release/acquire around a ticket counter; no float
// atomics, so the result does not depend on arrival order).
//
//   grid = (centre blocks NB, queries B), block = 256; workspace part[NB][B][OP] + ticket[B].
#include "rbf_forward.h"

namespace irbfn {

struct SmallArgs {
  const float* __restrict__ x;      // [B][D]
  const float* __restrict__ rec;    // [N][S]
  const float* __restrict__ bias;   // [OP]
  float* __restrict__ out;          // [B][O]
  float* part;                      // [NB][B][OP]
  unsigned int* ticket;             // [B], zero between launches
  GateTables gate;
  int B, Dreal, O, N, K, S, basis, cpl;   // cpl = centres per lane
};

template <int D, int OP, int BC, int QT>
__global__ __launch_bounds__(256) void rbf_fwd_clane(const SmallArgs a) {
  __shared__ float gtab[QT][kMaxSplit * 32];      // gate factors of this block's queries (nsplit*max_ranges <= 256)
  __shared__ float wsum[4][QT][OP];
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.y * QT;
  const GateTables gt = a.gate;

  // gate factors (model.py:74-86): one thread per (dim, range), for each of the QT queries
  const int E = gt.nsplit * gt.max_ranges;
  if (tid < E) {
    const int d = tid / gt.max_ranges;
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int bq = (q0 + q) < a.B ? (q0 + q) : a.B - 1;
      gtab[q][tid] = gate_factor(a.x[(long)bq * a.Dreal + d], gt.lo[tid], gt.hi[tid], gt.delta[d]);
    }
  }
  __syncthreads();

  float xq[QT][D];                                // wave-uniform -> scalar loads / SGPR operands
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    const int bq = (q0 + q) < a.B ? (q0 + q) : a.B - 1;
    const float* xb = a.x + (long)bq * a.Dreal;
#pragma unroll
    for (int j = 0; j < D; ++j) xq[q][j] = j < a.Dreal ? xb[j < a.Dreal ? j : 0] : 0.0f;
  }

  // one region (all BASELINE nets): gamma depends on the query only -> computed once, outside the loop
  const bool single_region = a.N == a.K;
  float g1[QT];
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    float g = gt.n_ranges > 0 ? 1.0f : 0.0f;
    if (single_region && gt.n_ranges > 0)
      for (int d = 0; d < gt.nsplit; ++d) g *= gtab[q][d * gt.max_ranges + gt.dim_ranges[d]];
    g1[q] = g;
  }

  float acc[QT][OP];
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int o = 0; o < OP; ++o) acc[q][o] = 0.0f;
  const int n0 = blockIdx.x * (256 * a.cpl);
  for (int i = 0; i < a.cpl; ++i) {
    const int n = n0 + i * 256 + tid;
    if (n < a.N) {
      constexpr int S = (D + 1 + OP + 3) & ~3;    // record floats (16-byte multiple, 16-byte aligned base)
      float rp[S];
      const float4* rp4 = reinterpret_cast<const float4*>(a.rec + (size_t)n * S);
#pragma unroll
      for (int k = 0; k < S / 4; ++k) {           // coalesced: consecutive lanes read consecutive records
        const float4 v = rp4[k];
        rp[4 * k] = v.x; rp[4 * k + 1] = v.y; rp[4 * k + 2] = v.z; rp[4 * k + 3] = v.w;
      }
      const int r = n / a.K;
      float gq[QT];                               // gamma[q][r]: model.py:88-93 (regions without a range stay 0)
      if (single_region) {
#pragma unroll
        for (int q = 0; q < QT; ++q) gq[q] = g1[q];
      } else {
#pragma unroll
        for (int q = 0; q < QT; ++q) gq[q] = r < gt.n_ranges ? 1.0f : 0.0f;
        if (r < gt.n_ranges) {
          for (int d = 0; d < gt.nsplit; ++d) {
            const int e = d * gt.max_ranges + gt.dim_ranges[r * gt.nsplit + d];
#pragma unroll
            for (int q = 0; q < QT; ++q) gq[q] *= gtab[q][e];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        float r2 = 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float df = xq[q][j] - rp[j];      // flax_rbf.py:280
          r2 = __builtin_fmaf(df, df, r2);
        }
        float phi = basis_from_r2<BC>(r2, rp[D], a.basis);
        phi *= gq[q];                             // model.py:193
#pragma unroll
        for (int o = 0; o < OP; ++o) acc[q][o] = __builtin_fmaf(phi, rp[D + 1 + o], acc[q][o]);   // model.py:196
      }
    }
  }
  // wavefront shuffle partial sums, then LDS across the 4 waves
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int o = 0; o < OP; ++o) {
      float v = acc[q][o];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) wsum[wave][q][o] = v;
    }
  __syncthreads();
  const int NB = gridDim.x;
  for (int idx = tid; idx < QT * OP; idx += 256) {
    const int q = idx / OP, o = idx - q * OP;
    if (q0 + q < a.B)
      a.part[((size_t)blockIdx.x * a.B + q0 + q) * OP + o] =
          (wsum[0][q][o] + wsum[1][q][o]) + (wsum[2][q][o] + wsum[3][q][o]);
  }
  // publish: every storing wave drains, workgroup barrier, agent-scope release, ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add(a.ticket + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != (unsigned)(NB - 1)) return;     // not the last workgroup of this query tile
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.ticket[blockIdx.y] = 0u;                    // ready for the next launch
  }
  __syncthreads();
  for (int idx = tid; idx < QT * a.O; idx += 256) {
    const int q = idx / a.O, o = idx - q * a.O;
    if (q0 + q < a.B) {
      float s = 0.0f;
      for (int k = 0; k < NB; ++k)                // fixed order -> deterministic
        s += __builtin_nontemporal_load(a.part + ((size_t)k * a.B + q0 + q) * OP + o);
      a.out[(long)(q0 + q) * a.O + o] = s + a.bias[o];
    }
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int kSmallMaxB = 64;            // QT = 1 (latency) up to here
// A QT = 8 query-tiled variant for 64 < B <= 8192 was measured and rejected: the 80 wavefront-shuffle
// reductions per workgroup dominate its 8 x 256 pairs (B = 1024: 132 us vs 49 us for K1), so mid-size
// batches stay on K1.
constexpr int kTiledMaxB = kSmallMaxB;
constexpr int kSmallMaxNB = 256;
constexpr size_t kSmallWsFloats = (size_t)4 << 20;   // 16 MB of partial sums per descriptor
constexpr int kSmallTickets = 8192;

size_t small_workspace_floats(int) { return kSmallWsFloats; }
int small_ticket_count() { return kSmallTickets; }

bool small_eligible(const irbfn_net* net, int64_t B) {
  if (net->small_part == nullptr || net->nsplit * net->max_ranges > kMaxSplit * 32 || net->OP > 128) return false;
  if (B <= kSmallMaxB) return true;
  return B <= kTiledMaxB && net->OP <= 16;
}

template <int D, int OP, int QT>
static int launch_small_bc(const SmallArgs& a, int bc, dim3 grid, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_GAUSS, QT>), grid, dim3(256), 0, s, a); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_IQ, QT>), grid, dim3(256), 0, s, a); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_IMQ, QT>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_GENERIC, QT>), grid, dim3(256), 0, s, a); break;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int D>
static int launch_small_d(const SmallArgs& a, int OP, int QT, int bc, dim3 grid, hipStream_t s) {
#define IRBFN_SCASE(OPV)                                                     \
  case OPV:                                                                  \
    if (QT == 1) return launch_small_bc<D, OPV, 1>(a, bc, grid, s);          \
    return IRBFN_ERR_UNSUPPORTED;
  switch (OP) {
    IRBFN_SCASE(2)
    IRBFN_SCASE(4)
    IRBFN_SCASE(5)
    IRBFN_SCASE(8)
    IRBFN_SCASE(10)
    IRBFN_SCASE(16)
    IRBFN_SCASE(32)
    IRBFN_SCASE(64)
    IRBFN_SCASE(100)
    IRBFN_SCASE(128)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_SCASE
}

int launch_forward_small(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s) {
  SmallArgs a;
  a.x = x; a.rec = net->rec; a.bias = net->bias; a.out = out; a.part = net->small_part;
  a.ticket = net->small_ticket; a.gate = net->gate();
  a.B = (int)B; a.Dreal = net->D; a.O = net->O; a.N = net->N; a.K = net->K; a.S = net->S; a.basis = net->basis;
  const int QT = B <= kSmallMaxB ? 1 : 8;
  const long qtiles = (B + QT - 1) / QT;
  // enough workgroups to spread over the chip, bounded by the ticket / partial-sum workspace
  int cpl = 1;
  long nb = ((long)net->N + 255) / 256;
  while (nb > 1 && (nb > kSmallMaxNB || nb * qtiles > 4096 || (size_t)nb * B * net->OP > kSmallWsFloats)) {
    cpl *= 2;
    nb = ((long)net->N + 256L * cpl - 1) / (256L * cpl);
  }
  if ((size_t)nb * B * net->OP > kSmallWsFloats || qtiles > kSmallTickets) return IRBFN_ERR_UNSUPPORTED;
  a.cpl = cpl;
  const dim3 grid((unsigned)nb, (unsigned)qtiles);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_small_d<3>(a, net->OP, QT, net->bclass, grid, s); break;
    case 4: rc = launch_small_d<4>(a, net->OP, QT, net->bclass, grid, s); break;
    case 7: rc = launch_small_d<7>(a, net->OP, QT, net->bclass, grid, s); break;
    case 8: rc = launch_small_d<8>(a, net->OP, QT, net->bclass, grid, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_clane<D=%d,OP=%d,BC=%d,QT=%d>", net->DC, net->OP,
             net->bclass, QT);
    net->last_grid = (int)(nb * qtiles);
    net->last_block = 256;
  }
  return rc;
}

}  // namespace irbfn
