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

template <int D, int OP, int BC>
__global__ __launch_bounds__(256) void rbf_fwd_clane(const SmallArgs a) {
  __shared__ float gtab[kMaxSplit * 32];          // gate factors of this query (nsplit * max_ranges <= 256)
  __shared__ float wsum[4][OP];
  __shared__ unsigned int s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const GateTables gt = a.gate;
  const float* xb = a.x + (long)b * a.Dreal;      // wave-uniform -> scalar loads

  // gate factors (model.py:74-86): one thread per (dim, range)
  const int E = gt.nsplit * gt.max_ranges;
  if (tid < E) {
    const int d = tid / gt.max_ranges;
    gtab[tid] = gate_factor(xb[d], gt.lo[tid], gt.hi[tid], gt.delta[d]);
  }
  __syncthreads();

  float xq[D];
#pragma unroll
  for (int j = 0; j < D; ++j) xq[j] = j < a.Dreal ? xb[j < a.Dreal ? j : 0] : 0.0f;

  float acc[OP];
#pragma unroll
  for (int o = 0; o < OP; ++o) acc[o] = 0.0f;
  const int n0 = blockIdx.x * (256 * a.cpl);
  for (int i = 0; i < a.cpl; ++i) {
    const int n = n0 + i * 256 + tid;
    if (n < a.N) {
      constexpr int S = (D + 1 + OP + 3) & ~3;    // record floats (16-byte multiple, 16-byte aligned base)
      float rp[S];
      const float4* rp4 = reinterpret_cast<const float4*>(a.rec + (size_t)n * S);
#pragma unroll
      for (int q = 0; q < S / 4; ++q) {           // coalesced: consecutive lanes read consecutive records
        const float4 v = rp4[q];
        rp[4 * q] = v.x; rp[4 * q + 1] = v.y; rp[4 * q + 2] = v.z; rp[4 * q + 3] = v.w;
      }
      float r2 = 0.0f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float df = xq[j] - rp[j];           // flax_rbf.py:280
        r2 = __builtin_fmaf(df, df, r2);
      }
      float phi = basis_from_r2<BC>(r2, rp[D], a.basis);
      const int r = n / a.K;
      float g = 0.0f;                             // model.py:88-93 (regions without a range stay 0)
      if (r < gt.n_ranges) {
        g = 1.0f;
        for (int d = 0; d < gt.nsplit; ++d) g *= gtab[d * gt.max_ranges + gt.dim_ranges[r * gt.nsplit + d]];
      }
      phi *= g;                                   // model.py:193
#pragma unroll
      for (int o = 0; o < OP; ++o) acc[o] = __builtin_fmaf(phi, rp[D + 1 + o], acc[o]);   // model.py:196
    }
  }
  // wavefront shuffle partial sums, then LDS across the 4 waves
#pragma unroll
  for (int o = 0; o < OP; ++o) {
    float v = acc[o];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) wsum[wave][o] = v;
  }
  __syncthreads();
  const int NB = gridDim.x;
  float* mypart = a.part + ((size_t)blockIdx.x * a.B + b) * OP;
  if (tid < OP) mypart[tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
  // publish: every storing wave drains, workgroup barrier, agent-scope release, ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket != (unsigned)(NB - 1)) return;     // not the last workgroup of this query
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.ticket[b] = 0u;                             // ready for the next launch
  }
  __syncthreads();
  if (tid < a.O) {
    float s = 0.0f;
    for (int k = 0; k < NB; ++k)                  // fixed order -> deterministic
      s += __builtin_nontemporal_load(a.part + ((size_t)k * a.B + b) * OP + tid);
    a.out[(long)b * a.O + tid] = s + a.bias[tid];
  }
}

// ------------------------------------------------------------------------------------------------
constexpr int kSmallMaxB = 64;
constexpr int kSmallMaxNB = 256;

size_t small_workspace_floats(int OP) { return (size_t)kSmallMaxNB * kSmallMaxB * OP; }

bool small_eligible(const irbfn_net* net, int64_t B) {
  return B <= kSmallMaxB && net->small_part != nullptr && net->nsplit * net->max_ranges <= kMaxSplit * 32 &&
         net->OP <= 128;
}

template <int D, int OP>
static int launch_small_bc(const SmallArgs& a, int bc, dim3 grid, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_GAUSS>), grid, dim3(256), 0, s, a); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_IQ>), grid, dim3(256), 0, s, a); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_IMQ>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((rbf_fwd_clane<D, OP, BC_GENERIC>), grid, dim3(256), 0, s, a); break;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int D>
static int launch_small_d(const SmallArgs& a, int OP, int bc, dim3 grid, hipStream_t s) {
  switch (OP) {
    case 2: return launch_small_bc<D, 2>(a, bc, grid, s);
    case 4: return launch_small_bc<D, 4>(a, bc, grid, s);
    case 5: return launch_small_bc<D, 5>(a, bc, grid, s);
    case 8: return launch_small_bc<D, 8>(a, bc, grid, s);
    case 10: return launch_small_bc<D, 10>(a, bc, grid, s);
    case 16: return launch_small_bc<D, 16>(a, bc, grid, s);
    case 32: return launch_small_bc<D, 32>(a, bc, grid, s);
    case 64: return launch_small_bc<D, 64>(a, bc, grid, s);
    case 100: return launch_small_bc<D, 100>(a, bc, grid, s);
    case 128: return launch_small_bc<D, 128>(a, bc, grid, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

int launch_forward_small(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s) {
  SmallArgs a;
  a.x = x; a.rec = net->rec; a.bias = net->bias; a.out = out; a.part = net->small_part;
  a.ticket = net->small_ticket; a.gate = net->gate();
  a.B = (int)B; a.Dreal = net->D; a.O = net->O; a.N = net->N; a.K = net->K; a.S = net->S; a.basis = net->basis;
  // enough workgroups to spread over the chip, at most kSmallMaxNB per query
  int cpl = 1;
  long nb = ((long)net->N + 255) / 256;
  while (nb > kSmallMaxNB || nb * B > 2048) {
    cpl *= 2;
    nb = ((long)net->N + 256L * cpl - 1) / (256L * cpl);
    if (nb <= 1) break;
  }
  a.cpl = cpl;
  const dim3 grid((unsigned)nb, (unsigned)B);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_small_d<3>(a, net->OP, net->bclass, grid, s); break;
    case 4: rc = launch_small_d<4>(a, net->OP, net->bclass, grid, s); break;
    case 7: rc = launch_small_d<7>(a, net->OP, net->bclass, grid, s); break;
    case 8: rc = launch_small_d<8>(a, net->OP, net->bclass, grid, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_clane<D=%d,OP=%d,BC=%d>", net->DC, net->OP, net->bclass);
    net->last_grid = (int)(nb * B);
    net->last_block = 256;
  }
  return rc;
}

}  // namespace irbfn
