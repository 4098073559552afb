// Dense head of DeeperWCRBFNet (src/irbfn_mpc/model.py:254-256, 283-287):
//     out = linear(relu(linear_pre2(relu(out_pre1))))
// where out_pre1 = linear_pre1(rbf_out) [B, H1] is produced by the fused RBF forward (K1 / K1m with
// the H1-wide `linear_pre1` as its Dense layer).  One lane owns one row: relu(out_pre1) (H1 values) and
// the H2 hidden accumulators live in VGPRs; the weights are wave-uniform and stream through the scalar
// cache.  H1 = H2 = 64 in the reference; the work is ~5 kFLOP per row, negligible next to the RBF stage.
#include "common.h"

namespace irbfn {

template <int H1, int H2>
__global__ __launch_bounds__(64) void mlp_head_kernel(const float* __restrict__ h1, const float* __restrict__ W2,
                                                      const float* __restrict__ b2, const float* __restrict__ W3,
                                                      const float* __restrict__ b3, float* __restrict__ out, long B,
                                                      int O) {
  extern __shared__ float tile[];                // [64][H1 + 1]
  const int lane = threadIdx.x;
  const long b0 = (long)blockIdx.x * kWave;
  const long left = B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const float* src = h1 + b0 * H1;               // the wave's rows are contiguous: coalesced copy
  for (int i = lane; i < nvalid * H1; i += kWave) tile[(i / H1) * (H1 + 1) + (i % H1)] = src[i];
  __syncthreads();
  const int rr = lane < nvalid ? lane : nvalid - 1;
  float z[H1];
#pragma unroll
  for (int i = 0; i < H1; ++i) z[i] = fmaxf(tile[rr * (H1 + 1) + i], 0.0f);      // nn.relu(out_pre1)
  float a2[H2];
#pragma unroll
  for (int j = 0; j < H2; ++j) a2[j] = b2[j];
#pragma unroll
  for (int i = 0; i < H1; ++i) {
    const float* wrow = W2 + i * H2;             // linear_pre2.kernel[i, :]  (uniform -> SGPRs)
#pragma unroll
    for (int j = 0; j < H2; ++j) a2[j] = __builtin_fmaf(z[i], wrow[j], a2[j]);
  }
#pragma unroll
  for (int j = 0; j < H2; ++j) a2[j] = fmaxf(a2[j], 0.0f);                         // nn.relu(out_pre2)
  __syncthreads();
  // final Dense (H2 -> O): stage through the tile for a coalesced store
  float* orow = tile + lane * (H1 + 1);
  for (int o = 0; o < O; ++o) {
    float acc = b3[o];
#pragma unroll
    for (int j = 0; j < H2; ++j) acc = __builtin_fmaf(a2[j], W3[j * O + o], acc);
    if (o < H1) orow[o] = acc;
    else if (lane < nvalid) out[(b0 + lane) * O + o] = acc;                        // O > H1: direct
  }
  __syncthreads();
  const int oc = O < H1 ? O : H1;
  for (int i = lane; i < nvalid * oc; i += kWave) {
    const int r = i / oc, o = i - r * oc;
    out[(b0 + r) * O + o] = tile[r * (H1 + 1) + o];
  }
}

// ---- backward of the head (the reference trains DeeperWCRBFNet: scripts/train_nmpc_frenet.py:339-421 with
// model.py:254-287) ---------------------------------------------------------------------------------------------
// gout[B,O] -> gh1[B,H1] (the cotangent of linear_pre1's output: the seed of the RBF-stage VJP) and the gradients of
// linear_pre2 / linear.  One lane per row recomputes the forward (relu masks), back-propagates through the two
// Dense layers with the weights as wave-uniform scalars, and parks relu(h1), d z2, relu(z2), gout in LDS; the weight
// gradients are then outer-product sums over the block's rows with lane l owning column l.  Blocks walk the batch
// with a grid stride and keep their sums in registers; per-block partial slabs are added up in block order by
// mlp_head_bwd_reduce_kernel: bitwise reproducible, no float atomics.
constexpr int kHeadBwdBlocks = 256;

template <int H1, int H2>
__global__ __launch_bounds__(64) void mlp_head_bwd_kernel(const float* __restrict__ h1, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, const float* __restrict__ W3,
                                                          const float* __restrict__ gout, float* __restrict__ gh1,
                                                          float* __restrict__ part, long B, int O) {
  extern __shared__ float lds[];                 // z1[64][H1+1], dz2[64][H2+1], a2[64][H2+1], g[64][O+1]
  constexpr int P1 = H1 + 1, P2 = H2 + 1;
  float* z1t = lds;
  float* dzt = z1t + kWave * P1;
  float* a2t = dzt + kWave * P2;
  float* gt = a2t + kWave * P2;
  const int PO = O + 1;
  const int lane = threadIdx.x;
  float aW2[H1], aW3[16];                        // lane l: d linear_pre2.kernel[:, l], d linear.kernel[l, :O] (O <= 16)
  float ab2 = 0.0f, ab3 = 0.0f;                  // d linear_pre2.bias[l], d linear.bias[l] (l < O)
#pragma unroll
  for (int i = 0; i < H1; ++i) aW2[i] = 0.0f;
#pragma unroll
  for (int o = 0; o < 16; ++o) aW3[o] = 0.0f;
  const long ntiles = (B + kWave - 1) / kWave;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long b0 = tile * kWave;
    const long left = B - b0;
    const int nvalid = left < kWave ? (int)left : kWave;
    // coalesced copies of the tile's h1 rows (into z1t, raw) and gout rows
    for (int i = lane; i < nvalid * H1; i += kWave) z1t[(i / H1) * P1 + (i % H1)] = h1[b0 * H1 + i];
    for (int i = lane; i < nvalid * O; i += kWave) gt[(i / O) * PO + (i % O)] = gout[b0 * O + i];
    __syncthreads();
    const bool live = lane < nvalid;
    float z[H1], a2[H2];
#pragma unroll
    for (int i = 0; i < H1; ++i) z[i] = live ? z1t[lane * P1 + i] : 0.0f;        // raw h1 (sign = relu mask)
#pragma unroll
    for (int j = 0; j < H2; ++j) a2[j] = b2[j];
#pragma unroll
    for (int i = 0; i < H1; ++i) {
      const float zi = fmaxf(z[i], 0.0f);
      const float* wrow = W2 + i * H2;
#pragma unroll
      for (int j = 0; j < H2; ++j) a2[j] = __builtin_fmaf(zi, wrow[j], a2[j]);   // z2 = relu(h1) W2 + b2
    }
    float dz[H2];
#pragma unroll
    for (int j = 0; j < H2; ++j) {
      float acc = 0.0f;
      for (int o = 0; o < O; ++o) acc = __builtin_fmaf(live ? gt[lane * PO + o] : 0.0f, W3[j * O + o], acc);   // d a2 = g W3^T
      dz[j] = a2[j] > 0.0f ? acc : 0.0f;                                                                     // relu'
    }
    __syncthreads();                               // every lane has read its raw h1 row
#pragma unroll
    for (int i = 0; i < H1; ++i) {
      const float* wrow = W2 + i * H2;
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < H2; ++j) acc = __builtin_fmaf(dz[j], wrow[j], acc);       // d a1 = d z2 W2^T
      const float gi = z[i] > 0.0f ? acc : 0.0f;
      z1t[lane * P1 + i] = live ? fmaxf(z[i], 0.0f) : 0.0f;                        // park relu(h1)
      z[i] = gi;
    }
#pragma unroll
    for (int j = 0; j < H2; ++j) {
      dzt[lane * P2 + j] = live ? dz[j] : 0.0f;
      a2t[lane * P2 + j] = live ? fmaxf(a2[j], 0.0f) : 0.0f;
    }
    if (!live)
      for (int o = 0; o < O; ++o) gt[lane * PO + o] = 0.0f;
    __syncthreads();
    // d h1 rows out (through a2's slot? no: z holds them) -- per-lane rows, H1 contiguous floats each
    if (live) {
      float* dst = gh1 + (b0 + lane) * H1;
#pragma unroll
      for (int i = 0; i < H1; i += 4) *reinterpret_cast<float4*>(dst + i) = float4{z[i], z[i + 1], z[i + 2], z[i + 3]};
    }
    // weight gradients of this tile: lane l owns column l
    for (int b = 0; b < kWave; ++b) {
      const float dzl = dzt[b * P2 + lane], a2l = a2t[b * P2 + lane];
      ab2 += dzl;
#pragma unroll
      for (int i = 0; i < H1; ++i) aW2[i] = __builtin_fmaf(z1t[b * P1 + i], dzl, aW2[i]);
#pragma unroll
      for (int o = 0; o < 16; ++o)
        if (o < O) aW3[o] = __builtin_fmaf(a2l, gt[b * PO + o], aW3[o]);
      if (lane < O) ab3 += gt[b * PO + lane];
    }
    __syncthreads();
  }
  // partial slab of this block: [H1*H2 | H2 | H2*O | O]
  float* pp = part + (size_t)blockIdx.x * (H1 * H2 + H2 + H2 * O + O);
#pragma unroll
  for (int i = 0; i < H1; ++i) pp[i * H2 + lane] = aW2[i];
  pp[H1 * H2 + lane] = ab2;
#pragma unroll
  for (int o = 0; o < 16; ++o)
    if (o < O) pp[H1 * H2 + H2 + lane * O + o] = aW3[o];
  if (lane < O) pp[H1 * H2 + H2 + H2 * O + lane] = ab3;
}

__global__ __launch_bounds__(256) void mlp_head_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int n,
                                                                  float* __restrict__ gw2, float* __restrict__ gb2,
                                                                  float* __restrict__ gw3, float* __restrict__ gb3,
                                                                  int n_w2, int n_b2, int n_w3) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float acc = 0.0f;
  for (int k = 0; k < nblk; ++k) acc += part[(size_t)k * n + e];                   // fixed order
  if (e < n_w2) gw2[e] = acc;
  else if (e < n_w2 + n_b2) gb2[e - n_w2] = acc;
  else if (e < n_w2 + n_b2 + n_w3) gw3[e - n_w2 - n_b2] = acc;
  else gb3[e - n_w2 - n_b2 - n_w3] = acc;
}

}  // namespace irbfn

using namespace irbfn;

extern "C" int irbfn_mlp_head_forward(const float* h1_dev, const float* w2_dev, const float* b2_dev,
                                      const float* w3_dev, const float* b3_dev, float* out_dev, int64_t B, int H1,
                                      int H2, int O, void* stream) {
  if (B < 0 || O < 1) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!h1_dev || !w2_dev || !b2_dev || !w3_dev || !b3_dev || !out_dev) return IRBFN_ERR_BAD_ARG;
  if (H1 != 64 || H2 != 64) return IRBFN_ERR_UNSUPPORTED;      // the reference hard-codes Dense(64), Dense(64)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = (size_t)kWave * (64 + 1) * sizeof(float);
  hipLaunchKernelGGL((mlp_head_kernel<64, 64>), dim3((unsigned)((B + kWave - 1) / kWave)), dim3(kWave), lds, s, h1_dev,
                     w2_dev, b2_dev, w3_dev, b3_dev, out_dev, (long)B, O);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

extern "C" int64_t irbfn_mlp_head_vjp_workspace_bytes(int H1, int H2, int O) {
  if (H1 != 64 || H2 != 64 || O < 1 || O > 16) return IRBFN_ERR_UNSUPPORTED;
  return (int64_t)kHeadBwdBlocks * (H1 * H2 + H2 + H2 * O + O) * (int64_t)sizeof(float);
}

extern "C" int irbfn_mlp_head_vjp(const float* h1_dev, const float* w2_dev, const float* b2_dev, const float* w3_dev,
                                  const float* gout_dev, float* gh1_dev, float* gw2_dev, float* gb2_dev, float* gw3_dev,
                                  float* gb3_dev, int64_t B, int H1, int H2, int O, void* ws_dev, int64_t ws_bytes,
                                  void* stream) {
  if (B < 0 || O < 1) return IRBFN_ERR_BAD_ARG;
  if (H1 != 64 || H2 != 64 || O > 16) return IRBFN_ERR_UNSUPPORTED;
  if (!w2_dev || !b2_dev || !w3_dev || !gw2_dev || !gb2_dev || !gw3_dev || !gb3_dev || !ws_dev) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!h1_dev || !gout_dev || !gh1_dev)) return IRBFN_ERR_BAD_ARG;
  const int n = H1 * H2 + H2 + H2 * O + O;
  if (ws_bytes < (int64_t)kHeadBwdBlocks * n * (int64_t)sizeof(float)) return IRBFN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = ((size_t)kWave * (65 + 65 + 65) + (size_t)kWave * (O + 1)) * sizeof(float);
  auto k = mlp_head_bwd_kernel<64, 64>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  }
  hipLaunchKernelGGL(k, dim3(kHeadBwdBlocks), dim3(kWave), lds, s, h1_dev, w2_dev, b2_dev, w3_dev, gout_dev, gh1_dev,
                     static_cast<float*>(ws_dev), (long)B, O);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(mlp_head_bwd_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<const float*>(ws_dev),
                     kHeadBwdBlocks, n, gw2_dev, gb2_dev, gw3_dev, gb3_dev, H1 * H2, H2, H2 * O);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}
