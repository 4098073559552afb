// K1m: fused WCRBFNet forward with the Phi x W reduction on the f32 matrix cores (gfx950).
//
// Profiling K1 (rbf_fwd_qlane) on MI355X showed it VALU-issue bound: ~99 % VALU-active, and the ten
// weight-row FMAs per (query, centre) pair are ~40 % of the issue time.  gfx950 has an exact f32-input
// MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, bit-for-bit f32) that runs on its own pipe
// beside the VALU.  K1m keeps the distance + basis evaluation on the VALU in the direct, well
// conditioned form sum_j (x_j - c_j)^2 and hands phi to the matrix core:
//
//     D[16 queries x 16 outputs] += A[16 queries x 4 centres] * B[4 centres x 16 outputs]
//
// Lane l = (g = l >> 4, qs = l & 15) evaluates phi for centre slot g and queries {qs + 16 j}, which is
// exactly the A-operand layout (A[row = l & 15][k = l >> 4]); B[k = g][col = qs] = W[centre g][qs].
// The centre records are no longer wave-uniform (4 centres per step), so each wave streams its slice
// of the records HBM/L2 -> registers -> a wave-private LDS ring and reads them back as 16-lane
// broadcasts (2 x ds_read_b128 + NT x ds_read_b32 per step, amortised over QJ queries per lane).
// The NW waves of a workgroup split the centres; partial tiles are combined through LDS in a fixed
// order (deterministic), then gamma / bias / coalesced store as in K1.
//
// Used for R == 1 nets with the d^2-only fast bases; everything else stays on K1.
#include "rbf_forward.h"

namespace irbfn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kChunk = 16;     // centres per LDS chunk (4 MFMA k-steps)

struct MfmaArgs {
  const float* __restrict__ x;      // [B][D]
  const float* __restrict__ recm;   // [Npad][RS]
  const float* __restrict__ bias;   // [>= O]
  float* __restrict__ out;          // [B][O]
  GateTables gate;
  long B;
  int O, Npad, basis;
};

template <int D, int NT, int QJ, int BC>
__global__ __launch_bounds__(1024) void rbf_fwd_mfma(const MfmaArgs a) {
  extern __shared__ float lds[];
  constexpr int CW = mfma_cw(D);
  constexpr int OW = 16 * NT;
  constexpr int RS = CW + OW;                    // floats per record
  constexpr int ROWS = 16 * QJ;                  // queries per workgroup tile
  constexpr int CHF = kChunk * RS;               // floats per chunk
  constexpr int CH4 = CHF / 4;                   // float4 per chunk
  constexpr int NLD = (CH4 + kWave - 1) / kWave; // float4 loads per lane per chunk

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int nw = nthreads >> 6;
  const int qs = lane & 15, g = lane >> 4;
  const long row0 = (long)blockIdx.x * ROWS;
  const long left = a.B - row0;
  const int nvalid = left < ROWS ? (int)left : ROWS;
  const GateTables gt = a.gate;

  // ---- stage the query tile, pull this lane's QJ queries into registers
  float* xs = lds;                               // [ROWS][D]
  {
    const float* src = a.x + row0 * D;
    for (int i = tid; i < nvalid * D; i += nthreads) xs[i] = src[i];
  }
  __syncthreads();
  float xq[QJ][D];
  float gam[QJ];
#pragma unroll
  for (int j = 0; j < QJ; ++j) {
    int rr = qs + 16 * j;
    rr = rr < nvalid ? rr : nvalid - 1;
#pragma unroll
    for (int d = 0; d < D; ++d) xq[j][d] = xs[rr * D + d];
    float gm = gt.n_ranges > 0 ? 1.0f : 0.0f;    // model.py:70
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < gt.nsplit && gt.n_ranges > 0) {
        const int e = d * gt.max_ranges + gt.dim_ranges[d];
        gm *= gate_factor(xq[j][d], gt.lo[e], gt.hi[e], gt.delta[d]);   // model.py:83-85
      }
    }
    gam[j] = gm;
  }
  __syncthreads();                               // xs dead; the ring may overlap it

  // ---- wave-private LDS ring for the centre records: [nw][CHF]
  float* ring = lds + wave * CHF;
  const int chunks = a.Npad / kChunk;
  const int cpw = (chunks + nw - 1) / nw;
  const int c0 = wave * cpw;
  const int c1 = (c0 + cpw) < chunks ? (c0 + cpw) : chunks;

  f32x4 acc[QJ][NT];
#pragma unroll
  for (int j = 0; j < QJ; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[j][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // register prefetch of the first chunk
  float4 pf[NLD];
  auto fetch = [&](int c) {
    const float4* src = reinterpret_cast<const float4*>(a.recm + (size_t)c * CHF);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = lane + i * kWave;
      pf[i] = idx < CH4 ? src[idx] : float4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  if (c0 < c1) fetch(c0);
  for (int c = c0; c < c1; ++c) {
    // publish the prefetched chunk to the wave's ring (in-order LDS queue: no barrier inside a wave)
    {
      float4* dst = reinterpret_cast<float4*>(ring);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = lane + i * kWave;
        if (idx < CH4) dst[idx] = pf[i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c + 1 < c1) fetch(c + 1);                // next chunk in flight during the 4 MFMA steps
#pragma unroll
    for (int st = 0; st < kChunk / 4; ++st) {
      const float* rp = ring + (st * 4 + g) * RS;  // this lane group's centre
      float cv[CW];
#pragma unroll
      for (int i = 0; i < CW / 4; ++i) {
        const float4 v = reinterpret_cast<const float4*>(rp)[i];
        cv[4 * i + 0] = v.x; cv[4 * i + 1] = v.y; cv[4 * i + 2] = v.z; cv[4 * i + 3] = v.w;
      }
      float wv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wv[t] = rp[CW + 16 * t + qs];
      const float sc = cv[D];
#pragma unroll
      for (int j = 0; j < QJ; ++j) {
        float r2 = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const float df = xq[j][d] - cv[d];       // flax_rbf.py:280
          r2 = __builtin_fmaf(df, df, r2);
        }
        const float phi = basis_from_r2<BC>(r2, sc, a.basis);
#pragma unroll
        for (int t = 0; t < NT; ++t)               // model.py:196 on the matrix core (exact f32)
          acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(phi, wv[t], acc[j][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();             // all lanes done reading before the ring is rewritten
  }

  // ---- combine the NW partial tiles (fixed order), gamma, bias, coalesced store
  __syncthreads();
  constexpr int OPITCH = OW + 1;
  float* red = lds;                              // [nw][ROWS][OPITCH]
  float* grow = red + nw * ROWS * OPITCH;        // [ROWS]
  if (wave == 0 && g == 0) {
#pragma unroll
    for (int j = 0; j < QJ; ++j) grow[qs + 16 * j] = gam[j];
  }
#pragma unroll
  for (int j = 0; j < QJ; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)                // D[row = 4 g + r][col = qs]
        red[(wave * ROWS + 16 * j + 4 * g + r) * OPITCH + 16 * t + qs] = acc[j][t][r];
  __syncthreads();
  const int O = a.O;
  for (int idx = tid; idx < nvalid * O; idx += nthreads) {
    const int row = idx / O, o = idx - row * O;
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += red[(w * ROWS + row) * OPITCH + o];
    a.out[(row0 + row) * O + o] = grow[row] * s + a.bias[o];   // R == 1: gamma factors out (model.py:193)
  }
}

// ------------------------------------------------------------------------------------------------
size_t mfma_record_floats(int D, int O) { return (size_t)mfma_cw(D) + 16 * ((O + 15) / 16); }

bool mfma_eligible(const irbfn_net* net) {
  return net->R == 1 && net->bclass != BC_GENERIC && net->D >= 2 && net->D <= 8 && net->O <= 128;
}


template <int D, int NT, int QJ>
static int launch_mfma_bc(const MfmaArgs& a, int bc, int nw, size_t lds, long tiles, hipStream_t s) {
#define IRBFN_MCASE(BCV)                                                                             \
  case BCV: {                                                                                        \
    auto k = rbf_fwd_mfma<D, NT, QJ, BCV>;                                                           \
    if (lds > 48 * 1024) {                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                      \
    }                                                                                                \
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(nw * kWave), lds, s, a);                       \
    break;                                                                                           \
  }
  switch (bc) {
    IRBFN_MCASE(BC_GAUSS)
    IRBFN_MCASE(BC_IQ)
    IRBFN_MCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_MCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int D>
static int launch_mfma_d(const MfmaArgs& a, int NT, int QJ, int bc, int nw, size_t lds, long tiles, hipStream_t s) {
  if (QJ == 4) {
    switch (NT) {
      case 1: return launch_mfma_bc<D, 1, 4>(a, bc, nw, lds, tiles, s);
      case 2: return launch_mfma_bc<D, 2, 4>(a, bc, nw, lds, tiles, s);
      case 3: return launch_mfma_bc<D, 3, 4>(a, bc, nw, lds, tiles, s);
      case 4: return launch_mfma_bc<D, 4, 4>(a, bc, nw, lds, tiles, s);
    }
  } else if (QJ == 2) {
    switch (NT) {
      case 1: return launch_mfma_bc<D, 1, 2>(a, bc, nw, lds, tiles, s);
      case 2: return launch_mfma_bc<D, 2, 2>(a, bc, nw, lds, tiles, s);
      case 3: return launch_mfma_bc<D, 3, 2>(a, bc, nw, lds, tiles, s);
      case 4: return launch_mfma_bc<D, 4, 2>(a, bc, nw, lds, tiles, s);
      case 7: return launch_mfma_bc<D, 7, 2>(a, bc, nw, lds, tiles, s);
      case 8: return launch_mfma_bc<D, 8, 2>(a, bc, nw, lds, tiles, s);
    }
  } else if (QJ == 1) {
    switch (NT) {
      case 1: return launch_mfma_bc<D, 1, 1>(a, bc, nw, lds, tiles, s);
      case 2: return launch_mfma_bc<D, 2, 1>(a, bc, nw, lds, tiles, s);
      case 3: return launch_mfma_bc<D, 3, 1>(a, bc, nw, lds, tiles, s);
      case 4: return launch_mfma_bc<D, 4, 1>(a, bc, nw, lds, tiles, s);
      case 7: return launch_mfma_bc<D, 7, 1>(a, bc, nw, lds, tiles, s);
      case 8: return launch_mfma_bc<D, 8, 1>(a, bc, nw, lds, tiles, s);
    }
  }
  return IRBFN_ERR_UNSUPPORTED;
}

int launch_forward_mfma(irbfn_net* net, const float* x, float* out, int64_t B, int QJ, int nw, hipStream_t s) {
  const int D = net->D, NT = (net->O + 15) / 16, OW = 16 * NT;
  const int RS = mfma_cw(D) + OW;
  const int ROWS = 16 * QJ;
  const long tiles = (B + ROWS - 1) / ROWS;
  size_t ring = (size_t)nw * kChunk * RS;
  size_t stage = (size_t)ROWS * D;
  size_t red = (size_t)nw * ROWS * (OW + 1) + ROWS;
  size_t fl = ring > stage ? ring : stage;
  fl = fl > red ? fl : red;
  const size_t lds = fl * sizeof(float);
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  MfmaArgs a;
  a.x = x; a.recm = net->recm; a.bias = net->bias; a.out = out; a.gate = net->gate(); a.B = (long)B;
  a.O = net->O; a.Npad = net->Npad; a.basis = net->basis;
  int rc;
  switch (D) {
    case 2: rc = launch_mfma_d<2>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 3: rc = launch_mfma_d<3>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 4: rc = launch_mfma_d<4>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 5: rc = launch_mfma_d<5>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 6: rc = launch_mfma_d<6>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 7: rc = launch_mfma_d<7>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    case 8: rc = launch_mfma_d<8>(a, NT, QJ, net->bclass, nw, lds, tiles, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_mfma<D=%d,NT=%d,QJ=%d,BC=%d>", D, NT, QJ, net->bclass);
    net->last_grid = (int)tiles;
    net->last_block = nw * kWave;
  }
  return rc;
}

}  // namespace irbfn
