// Prototype of a matrix-core forward for NARROW outputs (O <= 16): Phi x W on the f16 MFMA with Phi and W
// split into f16 (hi, lo) pairs (~22 significant bits each; products exact in f32), so the 2*O weight
// FMAs per (query, centre) pair leave the VALU.  Gaussian basis, R = 1, D = 7.  Stand-alone: builds its
// own synthetic cfg-2 problem, checks against a float64 CPU evaluation, times the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/proto_f16split tools/proto_f16split.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int D = 7;
constexpr int CH = 32;                 // centres per chunk (= MFMA K)
constexpr int RECF = 8;                // floats per centre record: c[7], scale
constexpr int CHUNK_BYTES = CH * RECF * 4 + 2 * (4 * 16 * 16);   // 1024 B records + Whi + Wlo (1 KiB each)
constexpr float PHI_SCALE_LOG2 = 14.0f;

// chunk image in global memory == LDS image:
//   [0, 1024)    rec[32][8] floats
//   [1024, 2048) Whi[g][n][8] halfs: W[8g + j][n] / s_n   (hi part)
//   [2048, 3072) Wlo
template <int NW, int NMF>
__global__ __launch_bounds__(64 * NW) void fwd_f16split(const float* __restrict__ x, const unsigned char* __restrict__ chunks,
                                                       const float* __restrict__ oscale, const float* __restrict__ bias,
                                                       float* __restrict__ out, long B, int nchunks, int O) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 2 x CHUNK_BYTES
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * NW + wave) * 32;                  // this wave's 32 queries
  float xq[2][D];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    if (q >= B) q = B - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) xq[t][i] = x[q * D + i];
  }
  f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  // prologue: chunk 0 -> buffer 0
  constexpr int NV = CHUNK_BYTES / 16;                                  // 192 x 16 B
  for (int i = tid; i < NV; i += 64 * NW) ((u4*)lds)[i] = ((const u4*)chunks)[i];
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char* cur = lds + (c & 1) * CHUNK_BYTES;
    unsigned char* nxt = lds + ((c + 1) & 1) * CHUNK_BYTES;
    u4 pre;
    const bool has_next = c + 1 < nchunks;
    if (has_next && tid < NV) pre = ((const u4*)(chunks + (size_t)(c + 1) * CHUNK_BYTES))[tid];
    const h8 bh = *(const h8*)(cur + 1024 + (g * 16 + n) * 16);
    const h8 bl = *(const h8*)(cur + 2048 + (g * 16 + n) * 16);
    float phi[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f4 r0 = *(const f4*)(cur + ((8 * g + j) * RECF) * 4);
      const f4 r1 = *(const f4*)(cur + ((8 * g + j) * RECF + 4) * 4);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float d = xq[t][0] - r0.x; float r2 = d * d;
        d = xq[t][1] - r0.y; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][2] - r0.z; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][3] - r0.w; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][4] - r1.x; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][5] - r1.y; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][6] - r1.z; r2 = __builtin_fmaf(d, d, r2);
        phi[t][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(r2, r1.w, PHI_SCALE_LOG2));
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      h8 ah, al;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float p0 = phi[t][2 * jj], p1 = phi[t][2 * jj + 1];
        const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
        const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
        const h2 hh = __builtin_amdgcn_cvt_pkrtz(h0, h1);
        const h2 ll = __builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1);
        ah[2 * jj] = (_Float16)hh[0]; ah[2 * jj + 1] = (_Float16)hh[1];
        al[2 * jj] = (_Float16)ll[0]; al[2 * jj + 1] = (_Float16)ll[1];
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
      if (NMF >= 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
      if (NMF >= 3) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
      if (NMF >= 4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, acc[t], 0, 0, 0);
    }
    if (has_next && tid < NV) ((u4*)nxt)[tid] = pre;
    __syncthreads();
  }
  // D layout: col = lane & 15 (output), row = 4 * (lane >> 4) + reg (query within the tile)
  if (n < O) {
    const float sc = oscale[n] * 6.103515625e-05f;                      // s_n * 2^-14
    const float bi = bias[n];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long q = q0 + t * 16 + 4 * g + r;
        if (q < B) out[q * O + n] = __builtin_fmaf(acc[t][r], sc, bi);
      }
  }
}


// v2: wave-private chunk stream (no block barriers in the loop), W operands straight from global into
// VGPRs, centre slices across waves (S slices x QG query groups per block), MFMAs of step c-1 interleaved
// with the distance work of step c.
template <int S, int QG, int NMF, bool BATCH, int ABL = 0>
__global__ __launch_bounds__(64 * S * QG) void fwd_f16split_v2(const float* __restrict__ x, const unsigned char* __restrict__ chunks,
                                                              const float* __restrict__ oscale, const float* __restrict__ bias,
                                                              float* __restrict__ out, long B, int nchunks, int O) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // per wave 2 x 1 KiB; reused for the final reduce
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slice = wave % S, qg = wave / S;
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * 32;
  float xq[2][D];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    if (q >= B) q = B - 1;
    if (q < 0) q = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) xq[t][i] = x[q * D + i];
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < D; ++i) asm volatile("" : "+v"(xq[t][i]));      // loads complete before the loop
  unsigned char* mylds = lds + wave * 2048;
  const int c0 = (int)((long)nchunks * slice / S), c1 = (int)((long)nchunks * (slice + 1) / S);
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  f4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  h8 ah[2], al[2], bh, bl;                       // operands of the PREVIOUS step (MFMAs deferred by one step)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ah[t][j] = 0; al[t][j] = 0; }
#pragma unroll
  for (int j = 0; j < 8; ++j) { bh[j] = 0; bl[j] = 0; }
  if (c0 < c1) {
    const unsigned char* p = chunks + (size_t)c0 * CHUNK_BYTES;
    ((u4*)mylds)[lane] = ((const u4*)p)[lane];
  }
  wave_sync();
  for (int c = c0; c < c1; ++c) {
    const unsigned char* cur = mylds + ((c - c0) & 1) * 1024;
    unsigned char* nxt = mylds + ((c - c0 + 1) & 1) * 1024;
    const unsigned char* gp = chunks + (size_t)c * CHUNK_BYTES;
    const bool has_next = c + 1 < c1;
    u4 pre = {0, 0, 0, 0};
    if (has_next) pre = ((const u4*)(gp + CHUNK_BYTES))[lane];
    const h8 nbh = *(const h8*)(gp + 1024 + lane * 16);
    const h8 nbl = *(const h8*)(gp + 2048 + lane * 16);
    float phi[2][8];
    const f4 hoist0 = *(const f4*)(mylds + g * 32), hoist1 = *(const f4*)(mylds + g * 32 + 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f4 r0, r1;
      if (ABL == 4) { r0 = hoist0; r1 = hoist1; asm volatile("" : "+v"(r0), "+v"(r1)); }
      else {
        r0 = *(const f4*)(cur + ((8 * g + j) * RECF) * 4);
        r1 = *(const f4*)(cur + ((8 * g + j) * RECF + 4) * 4);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float d = xq[t][0] - r0.x; float r2 = d * d;
        d = xq[t][1] - r0.y; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][2] - r0.z; r2 = __builtin_fmaf(d, d, r2);
        if (ABL != 5) {
        d = xq[t][3] - r0.w; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][4] - r1.x; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][5] - r1.y; r2 = __builtin_fmaf(d, d, r2);
        d = xq[t][6] - r1.z; r2 = __builtin_fmaf(d, d, r2);
        }
        if (BATCH) phi[t][j] = __builtin_fmaf(r2, r1.w, PHI_SCALE_LOG2);
        else phi[t][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(r2, r1.w, PHI_SCALE_LOG2));
      }
      // one deferred MFMA per centre: (tile, term) = (j & 1, j >> 1)
      const int t = j & 1, m = j >> 1;
      if (m < NMF) {
        const h8 a = (m == 0 || m == 2) ? ah[t] : al[t];
        const h8 b = (m == 0 || m == 1) ? bh : bl;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
      }
    }
    if (BATCH && ABL != 1) {
      // the transcendental unit costs ~33 cycles for an isolated v_exp_f32 but 8 per instruction back to back
      // (tools/ubench_trans.hip): issue the 16 of this step as one block
      asm volatile(
          "v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n"
          "v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n"
          "v_exp_f32_e32 %8, %8\n v_exp_f32_e32 %9, %9\n v_exp_f32_e32 %10, %10\n v_exp_f32_e32 %11, %11\n"
          "v_exp_f32_e32 %12, %12\n v_exp_f32_e32 %13, %13\n v_exp_f32_e32 %14, %14\n v_exp_f32_e32 %15, %15\n"
          "s_nop 1\n"
          : "+v"(phi[0][0]), "+v"(phi[0][1]), "+v"(phi[0][2]), "+v"(phi[0][3]), "+v"(phi[0][4]), "+v"(phi[0][5]),
            "+v"(phi[0][6]), "+v"(phi[0][7]), "+v"(phi[1][0]), "+v"(phi[1][1]), "+v"(phi[1][2]), "+v"(phi[1][3]),
            "+v"(phi[1][4]), "+v"(phi[1][5]), "+v"(phi[1][6]), "+v"(phi[1][7]));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float p0 = phi[t][2 * jj], p1 = phi[t][2 * jj + 1];
        const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
        const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
        const h2 hh = ABL == 2 ? __builtin_amdgcn_cvt_pkrtz(p0, p1) : __builtin_amdgcn_cvt_pkrtz(h0, h1);
        const h2 ll = ABL == 2 ? hh : __builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1);
        ah[t][2 * jj] = (_Float16)hh[0]; ah[t][2 * jj + 1] = (_Float16)hh[1];
        al[t][2 * jj] = (_Float16)ll[0]; al[t][2 * jj + 1] = (_Float16)ll[1];
      }
    }
    bh = nbh; bl = nbl;
    if (has_next) ((u4*)nxt)[lane] = pre;
    wave_sync();
  }
  // drain the deferred MFMAs of the last step
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t], 0, 0, 0);
    if (NMF >= 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t], 0, 0, 0);
    if (NMF >= 3) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[t], 0, 0, 0);
    if (NMF >= 4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bl, acc[t], 0, 0, 0);
  }
  // reduce the S slices in fixed order through LDS: red[qg][slice][t][r][lane]
  __syncthreads();
  float* red = (float*)lds;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(((qg * S + slice) * 2 + t) * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  if (slice == 0 && n < O) {
    const float sc = oscale[n] * 6.103515625e-05f;
    const float bi = bias[n];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
        for (int s2 = 0; s2 < S; ++s2) v += red[(((qg * S + s2) * 2 + t) * 4 + r) * 64 + lane];
        const long q = q0 + t * 16 + 4 * g + r;
        if (q < B) out[q * O + n] = __builtin_fmaf(v, sc, bi);
      }
  }
}

// v3: TL tiles of 16 queries per wave (TL = 2 or 4), NMF MFMA terms, exps of a step as one block.
template <int NT> __device__ __forceinline__ void exp_block(float (&v)[NT]);
template <> __device__ __forceinline__ void exp_block<16>(float (&v)[16]) {
  asm volatile(
      "v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n"
      "v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n"
      "v_exp_f32_e32 %8, %8\n v_exp_f32_e32 %9, %9\n v_exp_f32_e32 %10, %10\n v_exp_f32_e32 %11, %11\n"
      "v_exp_f32_e32 %12, %12\n v_exp_f32_e32 %13, %13\n v_exp_f32_e32 %14, %14\n v_exp_f32_e32 %15, %15\n s_nop 1\n"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
        "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

template <int S, int QG, int NMF, int TL>
__global__ __launch_bounds__(64 * S * QG) void fwd_f16split_v3(const float* __restrict__ x, const unsigned char* __restrict__ chunks,
                                                              const float* __restrict__ oscale, const float* __restrict__ bias,
                                                              float* __restrict__ out, long B, int nchunks, int O) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slice = wave % S, qg = wave / S;
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * (16 * TL);
  float xq[TL][D];
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    long q = q0 + t * 16 + n;
    if (q >= B) q = B - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) xq[t][i] = x[q * D + i];
  }
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int i = 0; i < D; ++i) asm volatile("" : "+v"(xq[t][i]));
  unsigned char* mylds = lds + wave * 2048;
  const int c0 = (int)((long)nchunks * slice / S), c1 = (int)((long)nchunks * (slice + 1) / S);
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  f4 acc[TL];
  h8 ah[TL], al[TL], bh, bl;
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    acc[t] = f4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) { ah[t][j] = 0; al[t][j] = 0; }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { bh[j] = 0; bl[j] = 0; }
  if (c0 < c1) ((u4*)mylds)[lane] = ((const u4*)(chunks + (size_t)c0 * CHUNK_BYTES))[lane];
  wave_sync();
  for (int c = c0; c < c1; ++c) {
    const unsigned char* cur = mylds + ((c - c0) & 1) * 1024;
    unsigned char* nxt = mylds + ((c - c0 + 1) & 1) * 1024;
    const unsigned char* gp = chunks + (size_t)c * CHUNK_BYTES;
    const bool has_next = c + 1 < c1;
    u4 pre = {0, 0, 0, 0};
    if (has_next) pre = ((const u4*)(gp + CHUNK_BYTES))[lane];
    const h8 nbh = *(const h8*)(gp + 1024 + lane * 16);
    const h8 nbl = *(const h8*)(gp + 2048 + lane * 16);
#pragma unroll
    for (int tp = 0; tp < TL; tp += 2) {          // two tiles (16 exps) at a time
      float phi[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f4 r0 = *(const f4*)(cur + ((8 * g + j) * RECF) * 4);
        const f4 r1 = *(const f4*)(cur + ((8 * g + j) * RECF + 4) * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float d = xq[tp + t][0] - r0.x; float r2 = d * d;
          d = xq[tp + t][1] - r0.y; r2 = __builtin_fmaf(d, d, r2);
          d = xq[tp + t][2] - r0.z; r2 = __builtin_fmaf(d, d, r2);
          d = xq[tp + t][3] - r0.w; r2 = __builtin_fmaf(d, d, r2);
          d = xq[tp + t][4] - r1.x; r2 = __builtin_fmaf(d, d, r2);
          d = xq[tp + t][5] - r1.y; r2 = __builtin_fmaf(d, d, r2);
          d = xq[tp + t][6] - r1.z; r2 = __builtin_fmaf(d, d, r2);
          phi[t * 8 + j] = __builtin_fmaf(r2, r1.w, PHI_SCALE_LOG2);
        }
        // deferred MFMAs of the previous step for this tile pair: term m = j >> 1 on tile tp + (j & 1)
        const int t = tp + (j & 1), m = j >> 1;
        if (m < NMF) {
          const h8 a = (m == 0 || m == 2) ? ah[t] : al[t];
          const h8 b = (m == 0 || m == 1) ? bh : bl;
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
        }
      }
      exp_block<16>(phi);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float p0 = phi[t * 8 + 2 * jj], p1 = phi[t * 8 + 2 * jj + 1];
          const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p0) & 0xFFFFE000u);
          const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p1) & 0xFFFFE000u);
          const h2 hh = __builtin_amdgcn_cvt_pkrtz(h0, h1);
          const h2 ll = __builtin_amdgcn_cvt_pkrtz(p0 - h0, p1 - h1);
          ah[tp + t][2 * jj] = (_Float16)hh[0]; ah[tp + t][2 * jj + 1] = (_Float16)hh[1];
          al[tp + t][2 * jj] = (_Float16)ll[0]; al[tp + t][2 * jj + 1] = (_Float16)ll[1];
        }
    }
    // NOTE: ah/al of tile pair tp are overwritten before the NEXT step's MFMAs read them -> the deferred MFMAs
    // of a tile pair use this step's B operands only after bh/bl are updated below; keep previous B for them.
    bh = nbh; bl = nbl;
    if (has_next) ((u4*)nxt)[lane] = pre;
    wave_sync();
  }
#pragma unroll
  for (int t = 0; t < TL; ++t) {
    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t], 0, 0, 0);
    if (NMF >= 2) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acc[t], 0, 0, 0);
    if (NMF >= 3) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acc[t], 0, 0, 0);
    if (NMF >= 4) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bl, acc[t], 0, 0, 0);
  }
  __syncthreads();
  float* red = (float*)lds;
#pragma unroll
  for (int t = 0; t < TL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(((qg * S + slice) * TL + t) * 4 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  if (slice == 0 && n < O) {
    const float sc = oscale[n] * 6.103515625e-05f;
    const float bi = bias[n];
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.0f;
        for (int s2 = 0; s2 < S; ++s2) v += red[(((qg * S + s2) * TL + t) * 4 + r) * 64 + lane];
        const long q = q0 + t * 16 + 4 * g + r;
        if (q < B) out[q * O + n] = __builtin_fmaf(v, sc, bi);
      }
  }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main(int argc, char** argv) {
  const long B = argc > 1 ? atol(argv[1]) : 65536;
  const int N = argc > 2 ? atoi(argv[2]) : 4096;
  const int O = 10;
  srand(123);
  auto uni = [](double lo, double hi) { return lo + (hi - lo) * (rand() / (double)RAND_MAX); };
  auto nrm = [&]() { double u = uni(1e-12, 1), v = uni(0, 1); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
  const double lo[7] = {0, 0, 0, -3.1, 0, -0.6, -3.0}, hi[7] = {7, 3.6, 3.6, 3.2, 7, 0.4, 2.5};
  std::vector<float> x(B * D), cen(N * D), ls(N), W((size_t)N * O), bias(O);
  for (long b = 0; b < B; ++b) for (int i = 0; i < D; ++i) x[b * D + i] = (float)uni(lo[i], hi[i]);
  for (int k = 0; k < N; ++k) { for (int i = 0; i < D; ++i) cen[k * D + i] = (float)uni(lo[i] - 1, hi[i] + 1); ls[k] = (float)uni(0, 2); }
  for (size_t i = 0; i < W.size(); ++i) W[i] = (float)nrm();
  for (int o = 0; o < O; ++o) bias[o] = (float)(0.1 * nrm());
  // pack
  const int nchunks = N / CH;
  std::vector<unsigned char> img((size_t)nchunks * CHUNK_BYTES, 0);
  std::vector<float> oscale(16, 1.0f);
  for (int o = 0; o < O; ++o) { float m = 0; for (int k = 0; k < N; ++k) m = fmaxf(m, fabsf(W[(size_t)k * O + o])); int e; frexpf(m, &e); oscale[o] = ldexpf(1.0f, e); }
  for (int c = 0; c < nchunks; ++c) {
    unsigned char* p = img.data() + (size_t)c * CHUNK_BYTES;
    float* rec = (float*)p;
    unsigned short* wh = (unsigned short*)(p + 1024);
    unsigned short* wl = (unsigned short*)(p + 2048);
    for (int k = 0; k < CH; ++k) {
      const int kk = c * CH + k;
      for (int i = 0; i < D; ++i) rec[k * RECF + i] = cen[kk * D + i];
      rec[k * RECF + 7] = (float)(-1.4426950408889634 * exp(-2.0 * (double)ls[kk]));
    }
    for (int g = 0; g < 4; ++g) for (int n = 0; n < 16; ++n) for (int j = 0; j < 8; ++j) {
      const int kk = c * CH + 8 * g + j;
      const float w = n < O ? W[(size_t)kk * O + n] / oscale[n] : 0.0f;
      const unsigned short h = f2h(w);
      wh[(g * 16 + n) * 8 + j] = h;
      wl[(g * 16 + n) * 8 + j] = f2h(w - h2f(h));
    }
  }
  float *dx, *dos, *dbias, *dout; unsigned char* dimg;
  CHECK(hipMalloc(&dx, x.size() * 4)); CHECK(hipMalloc(&dimg, img.size())); CHECK(hipMalloc(&dos, 64)); CHECK(hipMalloc(&dbias, 64));
  CHECK(hipMalloc(&dout, (size_t)B * O * 4));
  CHECK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dos, oscale.data(), 64, hipMemcpyHostToDevice)); std::vector<float> b16(16, 0); for (int o = 0; o < O; ++o) b16[o] = bias[o];
  CHECK(hipMemcpy(dbias, b16.data(), 64, hipMemcpyHostToDevice));
  // reference on a subset (double)
  const int NS = 512;
  std::vector<double> ref((size_t)NS * O), mag((size_t)NS * O);
  for (int s = 0; s < NS; ++s) {
    const long b = (long)s * (B / NS);
    for (int o = 0; o < O; ++o) { ref[s * O + o] = bias[o]; mag[s * O + o] = fabs(bias[o]); }
    for (int k = 0; k < N; ++k) {
      double r2 = 0; for (int i = 0; i < D; ++i) { double d = (double)x[b * D + i] - cen[k * D + i]; r2 += d * d; }
      const double ph = exp(-r2 * exp(-2.0 * (double)ls[k]));
      for (int o = 0; o < O; ++o) { ref[s * O + o] += ph * W[(size_t)k * O + o]; mag[s * O + o] += fabs(ph * W[(size_t)k * O + o]); }
    }
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](auto kern, int nw, const char* name) {
    const int grid = (int)((B + 32 * nw - 1) / (32 * nw));
    CHECK(hipMemset(dout, 0, (size_t)B * O * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), 2 * CHUNK_BYTES, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), 2 * CHUNK_BYTES, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> o((size_t)B * O);
    CHECK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, erel = 0, omax = 0;
    for (int s = 0; s < NS; ++s) { const long b = (long)s * (B / NS); for (int oo = 0; oo < O; ++oo) {
      const double e = fabs(o[b * O + oo] - ref[s * O + oo]); emax = fmax(emax, e); erel = fmax(erel, e / mag[s * O + oo]); omax = fmax(omax, fabs(ref[s * O + oo])); } }
    const double us = ms * 1e3 / reps;
    printf("%-28s grid %6d x %4d  %8.1f us  %.3e evals/s  %.1f cyc@2.4/pair-row/SIMD  max|err| %.2e (out max %.1f)  max err/sum|terms| %.2e\n",
           name, grid, 64 * nw, us, B / (us * 1e-6), us * 1e-6 * 2.4e9 * 1024 / ((double)B * N / 64), emax, omax, erel);
  };
  auto run2 = [&](auto kern, int S, int QG, const char* name) {
    const int grid = (int)((B + 32 * QG - 1) / (32 * QG));
    const size_t ldsb = (size_t)S * QG * 2048;
    CHECK(hipMemset(dout, 0, (size_t)B * O * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * S * QG), ldsb, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * S * QG), ldsb, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> o((size_t)B * O);
    CHECK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, erel = 0, omax = 0;
    for (int s = 0; s < NS; ++s) { const long b = (long)s * (B / NS); for (int oo = 0; oo < O; ++oo) {
      const double e = fabs(o[b * O + oo] - ref[s * O + oo]); emax = fmax(emax, e); erel = fmax(erel, e / mag[s * O + oo]); omax = fmax(omax, fabs(ref[s * O + oo])); } }
    const double us = ms * 1e3 / reps;
    printf("%-28s grid %6d x %4d  %8.1f us  %.3e evals/s  %.1f cyc@2.4/pair-row/SIMD  max|err| %.2e (out max %.1f)  max err/sum|terms| %.2e\n",
           name, grid, 64 * S * QG, us, B / (us * 1e-6), us * 1e-6 * 2.4e9 * 1024 / ((double)B * N / 64), emax, omax, erel);
  };
  auto run3 = [&](auto kern, int S, int QG, int TL, const char* name) {
    const int grid = (int)((B + 16 * TL * QG - 1) / (16 * TL * QG));
    size_t ldsb = (size_t)S * QG * 2048;
    const size_t redb = (size_t)S * QG * TL * 4 * 64 * 4;
    if (redb > ldsb) ldsb = redb;
    CHECK(hipMemset(dout, 0, (size_t)B * O * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * S * QG), ldsb, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * S * QG), ldsb, 0, dx, dimg, dos, dbias, dout, B, nchunks, O);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> o((size_t)B * O);
    CHECK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, erel = 0, omax = 0;
    for (int s = 0; s < NS; ++s) { const long b = (long)s * (B / NS); for (int oo = 0; oo < O; ++oo) {
      const double e = fabs(o[b * O + oo] - ref[s * O + oo]); emax = fmax(emax, e); erel = fmax(erel, e / mag[s * O + oo]); omax = fmax(omax, fabs(ref[s * O + oo])); } }
    const double us = ms * 1e3 / reps;
    printf("%-28s grid %6d x %4d  %8.1f us  %.3e evals/s  %.1f cyc@2.4/pair-row/SIMD  max|err| %.2e  max err/sum|terms| %.2e\n",
           name, grid, 64 * S * QG, us, B / (us * 1e-6), us * 1e-6 * 2.4e9 * 1024 / ((double)B * N / 64), emax, erel);
  };
  run3(fwd_f16split_v3<8, 1, 4, 2>, 8, 1, 2, "v3 S8 TL2 4mfma");
  run3(fwd_f16split_v3<8, 1, 3, 2>, 8, 1, 2, "v3 S8 TL2 3mfma");
  run3(fwd_f16split_v3<4, 2, 3, 2>, 4, 2, 2, "v3 S4 QG2 TL2 3mfma");
  run3(fwd_f16split_v3<8, 1, 3, 4>, 8, 1, 4, "v3 S8 TL4 3mfma");
  run3(fwd_f16split_v3<8, 2, 3, 4>, 8, 2, 4, "v3 S8 QG2 TL4 3mfma");
  run3(fwd_f16split_v3<16, 1, 3, 4>, 16, 1, 4, "v3 S16 TL4 3mfma");
  run3(fwd_f16split_v3<16, 1, 4, 4>, 16, 1, 4, "v3 S16 TL4 4mfma");
  run3(fwd_f16split_v3<4, 2, 3, 4>, 4, 2, 4, "v3 S4 QG2 TL4 3mfma");
  run2(fwd_f16split_v2<8, 1, 4, true, 0>, 8, 1, "v2 S8 full (ref)");
  return 0;
}
