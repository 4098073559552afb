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

// ---- the same backward on the f32 matrix cores (the default) -----------------------------------------------------------
// The kernel above gives a lane one row and does the weight gradients as per-row outer products fed from LDS, one wave per
// block with 54 KB of LDS: two waves per CU, 789 us at the reference's batch size (80000 rows) -- 40x its arithmetic.
// The five products of the head's backward are dense GEMMs with M or K = the batch:
//     z2 = relu(h1) W2 + b2            d z2 = (g W3^T) * [z2 > 0]            d h1 = (d z2 W2^T) * [h1 > 0]
//     d W2 = relu(h1)^T d z2           d W3 = relu(z2)^T g                   d b2 = sum_b d z2,  d b3 = sum_b g
// and run here on v_mfma_f32_16x16x4_f32 (exact f32 fma chains).  A wave owns 32-row tiles: h1 and g rows staged in LDS
// (A operands), W2 / W3 in a block-shared LDS copy (B operands), z2 / d z2 as D tiles in registers.  The K index of the
// two weight-gradient products is the batch, and ANY order of its terms will do: step k of those products is defined as
// "row 4g + r of row block mb" -- then the B operand (d z2, resp. the A operand relu(z2)) of lane (n, g) is the lane's own
// D-tile register r: no transpose, no LDS round trip.  Only d h1 needs d z2 with the batch on the A rows: one pass
// through an LDS tile.  Weight gradients accumulate in registers over the block's tiles; the four waves are summed in wave
// order through LDS, block slabs in block order by mlp_head_bwd_reduce_kernel: bitwise reproducible, no float atomics.
typedef float hf4 __attribute__((ext_vector_type(4)));
constexpr int kHeadTR = 32;                      // rows per wave tile
constexpr int kHeadP = 65;                       // LDS pitch of the 64-wide tiles
constexpr int kHeadGP = 17;                      // LDS pitch of the 16-wide tiles

template <int H1, int H2>
__global__ __launch_bounds__(256) void mlp_head_bwd_mfma_kernel(const float* __restrict__ h1, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, const float* __restrict__ W3,
                                                                const float* __restrict__ gout, float* __restrict__ gh1,
                                                                float* __restrict__ part, long B, int O) {
  static_assert(H1 == 64 && H2 == 64, "the reference hard-codes Dense(64), Dense(64)");
  extern __shared__ float lds[];
  constexpr int P = kHeadP, GP = kHeadGP, TR = kHeadTR, MB = TR / 16;
  float* W2s = lds;                              // [64][P]   W2[i][j]
  float* W3s = W2s + 64 * P;                     // [64][GP]  W3[j][o], o padded to 16 with zeros
  constexpr int WS = 2 * TR * P + TR * GP;       // floats per wave: Hs, Dz, Gs
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  float* Hs = W3s + 64 * GP + wave * WS;         // [TR][P]  raw h1 rows of the tile
  float* Dz = Hs + TR * P;                       // [TR][P]  d z2 with the batch on the rows
  float* Gs = Dz + TR * P;                       // [TR][GP] g rows, o padded to 16 with zeros
  {                                              // block-shared copies of the weights (loads first, then the LDS writes)
    float wv[16], w3v[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = W2[tid + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, j = i >> 4, o = i & 15;
      w3v[k] = o < O ? W3[j * O + o] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + 256 * k;
      W2s[(i >> 6) * P + (i & 63)] = wv[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k;
      W3s[(i >> 4) * GP + (i & 15)] = w3v[k];
    }
  }
  __syncthreads();
  auto wave_sync = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  hf4 dW2[4][4], dW3[4];
  float db2p[4] = {0.0f, 0.0f, 0.0f, 0.0f}, db3p = 0.0f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    dW3[mi] = hf4{0, 0, 0, 0};
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) dW2[mi][nj] = hf4{0, 0, 0, 0};
  }
  float b2n[4];
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) b2n[nj] = b2[nj * 16 + n];
  const long ntiles = (B + TR - 1) / TR;
  // a tile's h1 rows (8 x 16-byte loads per lane: row 4k + lane / 16, columns 4 (lane % 16) ..) and g rows, requested one
  // tile ahead
  hf4 hv[TR / 4];
  float gv[TR / 4];
  auto fetch = [&](long tile) {
    const long b0 = tile * TR;
#pragma unroll
    for (int k = 0; k < TR / 4; ++k) {
      const long row = b0 + 4 * k + (lane >> 4);
      const bool ok = tile < ntiles && row < B;
      hv[k] = ok ? *reinterpret_cast<const hf4*>(h1 + row * 64 + 4 * (lane & 15)) : hf4{0, 0, 0, 0};
      gv[k] = (ok && (lane & 15) < O) ? gout[row * O + (lane & 15)] : 0.0f;
    }
  };
  fetch((long)blockIdx.x * 4 + wave);
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long b0 = tile * TR;
    const long left = B - b0;
    const int nvalid = left < TR ? (int)left : TR;
#pragma unroll
    for (int k = 0; k < TR / 4; ++k) {
      float* d = Hs + (4 * k + (lane >> 4)) * P + 4 * (lane & 15);
      d[0] = hv[k].x; d[1] = hv[k].y; d[2] = hv[k].z; d[3] = hv[k].w;
      Gs[(4 * k + (lane >> 4)) * GP + (lane & 15)] = gv[k];
    }
    fetch(tile + (long)gridDim.x * 4);           // the next tile's rows travel while this one is computed
    wave_sync();
    // (1) z2 = relu(h1) W2 + b2: D tile (mb, nj) = rows b = 16 mb + 4 g + r, column j = 16 nj + n
    hf4 z2[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) z2[mb][nj] = hf4{b2n[nj], b2n[nj], b2n[nj], b2n[nj]};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      float av[MB], bv[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = fmaxf(Hs[(mb * 16 + n) * P + ks * 4 + g], 0.0f);
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) bv[nj] = W2s[(ks * 4 + g) * P + nj * 16 + n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) z2[mb][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nj], z2[mb][nj], 0, 0, 0);
    }
    // (2) d z2 = (g W3^T) * [z2 > 0]
    hf4 dz[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) dz[mb][nj] = hf4{0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float av[MB], bv[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = Gs[(mb * 16 + n) * GP + ks * 4 + g];
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) bv[nj] = W3s[(nj * 16 + n) * GP + ks * 4 + g];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) dz[mb][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nj], dz[mb][nj], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dz[mb][nj][r] = z2[mb][nj][r] > 0.0f ? dz[mb][nj][r] : 0.0f;          // relu'
          z2[mb][nj][r] = fmaxf(z2[mb][nj][r], 0.0f);                            // a2 = relu(z2)
          db2p[nj] += dz[mb][nj][r];
          Dz[(mb * 16 + 4 * g + r) * P + nj * 16 + n] = dz[mb][nj][r];
        }
    // (5) d W3[j][o] += sum_b a2[b][j] g[b][o], (4) d W2[i][j] += sum_b relu(h1[b][i]) d z2[b][j]: k-step = (mb, r), b = 16 mb + 4 g + r
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mb * 16 + 4 * g + r;
        const float gq = Gs[row * GP + n];
        db3p += gq;
#pragma unroll
        for (int mj = 0; mj < 4; ++mj) dW3[mj] = __builtin_amdgcn_mfma_f32_16x16x4f32(z2[mb][mj][r], gq, dW3[mj], 0, 0, 0);
        float av[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) av[mi] = fmaxf(Hs[row * P + mi * 16 + n], 0.0f);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj) dW2[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], dz[mb][nj][r], dW2[mi][nj], 0, 0, 0);
      }
    wave_sync();                                 // Dz is complete
    // (3) d h1 = (d z2 W2^T) * [h1 > 0]: D tile (mb, ni) = rows b, column i = 16 ni + n
    hf4 da[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) da[mb][ni] = hf4{0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      float av[MB], bv[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = Dz[(mb * 16 + n) * P + ks * 4 + g];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bv[ni] = W2s[(ni * 16 + n) * P + ks * 4 + g];        // B[k = j][col i] = W2[i][j]
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) da[mb][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[ni], da[mb][ni], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mb * 16 + 4 * g + r;
        if (row < nvalid) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const int i = ni * 16 + n;
            gh1[(b0 + row) * 64 + i] = Hs[row * P + i] > 0.0f ? da[mb][ni][r] : 0.0f;
          }
        }
      }
    wave_sync();                                 // Hs / Gs / Dz are free for the next tile
  }
  // ---- the block's slab [H1*H2 | H2 | H2*O | O]: the four waves summed in wave order through LDS
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    db2p[nj] += __shfl_xor(db2p[nj], 16);
    db2p[nj] += __shfl_xor(db2p[nj], 32);
  }
  db3p += __shfl_xor(db3p, 16);
  db3p += __shfl_xor(db3p, 32);
  const int nslab = H1 * H2 + H2 + H2 * O + O;
  __syncthreads();                               // W2s / W3s / the wave tiles are dead
  float* comb = lds + (size_t)wave * (64 * 64 + 64 + 64 * 16 + 16);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int r = 0; r < 4; ++r) comb[(mi * 16 + 4 * g + r) * 64 + nj * 16 + n] = dW2[mi][nj][r];
  if (g == 0) {
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) comb[64 * 64 + nj * 16 + n] = db2p[nj];
    comb[64 * 64 + 64 + 64 * 16 + n] = db3p;
  }
#pragma unroll
  for (int mj = 0; mj < 4; ++mj)
#pragma unroll
    for (int r = 0; r < 4; ++r) comb[64 * 64 + 64 + (mj * 16 + 4 * g + r) * 16 + n] = dW3[mj][r];
  __syncthreads();
  float* pp = part + (size_t)blockIdx.x * nslab;
  constexpr int CS = 64 * 64 + 64 + 64 * 16 + 16;
  for (int e = tid; e < nslab; e += 256) {
    int src;                                     // slab index -> index in the waves' (O padded to 16) layout
    if (e < 64 * 64 + 64) src = e;
    else if (e < 64 * 64 + 64 + 64 * O) { const int t = e - 64 * 64 - 64; src = 64 * 64 + 64 + (t / O) * 16 + (t % O); }
    else src = 64 * 64 + 64 + 64 * 16 + (e - 64 * 64 - 64 - 64 * O);
    pp[e] = (lds[src] + lds[CS + src]) + (lds[2 * CS + src] + lds[3 * CS + src]);
  }
}

// ---- forward head on the f32 matrix cores (the default): out = relu(relu(h1) W2 + b2) W3 + b3, 32-row tiles per wave,
// relu(z2) turned from D layout into A layout through the wave's LDS tile (mlp_head_kernel above: one lane per row, 52 us
// at 80000 rows; this one 12 us)
template <int H1, int H2>
__global__ __launch_bounds__(256) void mlp_head_fwd_mfma_kernel(const float* __restrict__ h1, const float* __restrict__ W2,
                                                                const float* __restrict__ b2, const float* __restrict__ W3,
                                                                const float* __restrict__ b3, float* __restrict__ out, long B, int O) {
  static_assert(H1 == 64 && H2 == 64, "the reference hard-codes Dense(64), Dense(64)");
  extern __shared__ float lds[];
  constexpr int P = kHeadP, GP = kHeadGP, TR = kHeadTR, MB = TR / 16;
  float* W2s = lds;                              // [64][P]
  float* W3s = W2s + 64 * P;                     // [64][GP]  W3[j][o], o padded to 16 with zeros
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  float* Hs = W3s + 64 * GP + wave * (TR * P);   // [TR][P]: relu(h1) rows, then relu(z2) rows
  {                                              // block-shared copies of the weights (loads first, then the LDS writes)
    float wv[16], w3v[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) wv[k] = W2[tid + 256 * k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, j = i >> 4, o = i & 15;
      w3v[k] = o < O ? W3[j * O + o] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + 256 * k;
      W2s[(i >> 6) * P + (i & 63)] = wv[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k;
      W3s[(i >> 4) * GP + (i & 15)] = w3v[k];
    }
  }
  __syncthreads();
  auto wave_sync = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  float b2n[4];
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) b2n[nj] = b2[nj * 16 + n];
  const float b3n = n < O ? b3[n] : 0.0f;
  const long ntiles = (B + TR - 1) / TR;
  // a tile's h1 rows: 8 x 16-byte loads per lane (row 4k + lane / 16, columns 4 (lane % 16) ..), requested one tile ahead
  hf4 hv[TR / 4];
  auto fetch = [&](long tile) {
    const long b0 = tile * TR;
#pragma unroll
    for (int k = 0; k < TR / 4; ++k) {
      const long row = b0 + 4 * k + (lane >> 4);
      hv[k] = (tile < ntiles && row < B) ? *reinterpret_cast<const hf4*>(h1 + row * 64 + 4 * (lane & 15)) : hf4{0, 0, 0, 0};
    }
  };
  fetch((long)blockIdx.x * 4 + wave);
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long b0 = tile * TR;
    const long left = B - b0;
    const int nvalid = left < TR ? (int)left : TR;
#pragma unroll
    for (int k = 0; k < TR / 4; ++k) {
      float* d = Hs + (4 * k + (lane >> 4)) * P + 4 * (lane & 15);
      d[0] = fmaxf(hv[k].x, 0.0f); d[1] = fmaxf(hv[k].y, 0.0f); d[2] = fmaxf(hv[k].z, 0.0f); d[3] = fmaxf(hv[k].w, 0.0f);   // nn.relu(out_pre1)
    }
    fetch(tile + (long)gridDim.x * 4);           // the next tile's rows travel while this one is computed
    wave_sync();
    hf4 z2[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) z2[mb][nj] = hf4{b2n[nj], b2n[nj], b2n[nj], b2n[nj]};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      float av[MB], bv[4];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) av[mb] = Hs[(mb * 16 + n) * P + ks * 4 + g];
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) bv[nj] = W2s[(ks * 4 + g) * P + nj * 16 + n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) z2[mb][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nj], z2[mb][nj], 0, 0, 0);
    }
    wave_sync();                                 // every read of relu(h1) is done: the tile takes relu(z2)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) Hs[(mb * 16 + 4 * g + r) * P + nj * 16 + n] = fmaxf(z2[mb][nj][r], 0.0f);   // nn.relu(out_pre2)
    wave_sync();
    hf4 oacc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) oacc[mb] = hf4{b3n, b3n, b3n, b3n};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const float bv = W3s[(ks * 4 + g) * GP + n];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        oacc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Hs[(mb * 16 + n) * P + ks * 4 + g], bv, oacc[mb], 0, 0, 0);
    }
    if (n < O) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mb * 16 + 4 * g + r;
          if (row < nvalid) out[(b0 + row) * O + n] = oacc[mb][r];
        }
    }
    wave_sync();
  }
}

// 64 consecutive values per block; thread (sg, v) sums slabs sg, sg + 4, ... in order, the four partial sums are added in
// a fixed order: deterministic, coalesced, no single thread walks all the slabs
__global__ __launch_bounds__(256) void mlp_head_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int n,
                                                                  float* __restrict__ gw2, float* __restrict__ gb2,
                                                                  float* __restrict__ gw3, float* __restrict__ gb3,
                                                                  int n_w2, int n_b2, int n_w3) {
  __shared__ float sm[4][64];
  const int v = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + v;
  float acc = 0.0f;
  if (e < n)
    for (int k = sg; k < nblk; k += 4) acc += part[(size_t)k * n + e];
  sm[sg][v] = acc;
  __syncthreads();
  if (sg != 0 || e >= n) return;
  acc = (sm[0][v] + sm[1][v]) + (sm[2][v] + sm[3][v]);
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
  if (O <= 16) {                                               // matrix-core head (every model card of the reference: O = 2 or 10)
    const size_t lds = ((size_t)64 * kHeadP + 64 * kHeadGP + 4 * (size_t)kHeadTR * kHeadP) * sizeof(float);
    auto k = mlp_head_fwd_mfma_kernel<64, 64>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
    const long tiles = (B + kHeadTR - 1) / kHeadTR;
    const long blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL(k, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), lds, s, h1_dev, w2_dev, b2_dev, w3_dev,
                       b3_dev, out_dev, (long)B, O);
    IRBFN_HIP_CHECK(hipGetLastError());
    return IRBFN_OK;
  }
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
  {
    // working set: W2 / W3 copies + four wave tile sets; the slab combine (4 x 5200 floats) reuses it
    size_t lf = 64 * kHeadP + 64 * kHeadGP + 4 * (size_t)(2 * kHeadTR * kHeadP + kHeadTR * kHeadGP);
    const size_t cf = 4 * (size_t)(64 * 64 + 64 + 64 * 16 + 16);
    lf = lf > cf ? lf : cf;
    const size_t lds = lf * sizeof(float);
    auto k = mlp_head_bwd_mfma_kernel<64, 64>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
    hipLaunchKernelGGL(k, dim3(kHeadBwdBlocks), dim3(256), lds, s, h1_dev, w2_dev, b2_dev, w3_dev, gout_dev, gh1_dev,
                       static_cast<float*>(ws_dev), (long)B, O);
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(mlp_head_bwd_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, s, static_cast<const float*>(ws_dev),
                     kHeadBwdBlocks, n, gw2_dev, gb2_dev, gw3_dev, gb3_dev, H1 * H2, H2, H2 * O);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}
