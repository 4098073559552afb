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
