// Wave specialisation for the wide K1h forward (BASELINE config 4: d = 7, O = 100 -> NT = 7 column tiles), measured on the
// kernel's own instruction streams before rewriting the kernel:
//   I  "interleaved"  today's structure: every one of the block's 8 waves owns 32 queries (2 tiles of 16) and runs, per
//                     32-centre step, the distances / basis / hi-lo split of its tiles on the VALU AND their 6 NT = 42 MFMAs
//                     (A operands in registers, W operands from LDS);
//   S  "specialised"  waves 0-3 are producers: VALU work of 64 queries (4 tiles) per step, A operands (hi, lo) written to a
//                     double-buffered LDS tile; waves 4-7 (their SIMD partners: a block's waves go to the SIMDs in cyclic
//                     order, so wave w and w + 4 share one) are consumers: A operands and W from LDS, 84 MFMAs per step, the
//                     accumulators of 4 tiles x 7 column tiles (224 VGPRs).  One s_barrier per step in both.
// Same total work per block and step; records / W operands come from an LDS image filled once (the LDS-DMA refill of the
// real kernel is the same in both forms and is left out).  Prints microseconds per launch, A/B interleaved on one lease.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I irbfn_amd/csrc tools/ubench_wavespec.hip -o tools/_bin/ubench_wavespec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "rbf_forward_f16_wide.h"

using namespace irbfn;

constexpr int DC = 7, RF = 8, NT = 7, BC = BC_GAUSS;
constexpr int RECB = kF16Chunk * RF * 4;                     // 1 KiB of records
constexpr int CB = RECB + NT * 2 * kF16WBytes;               // + 14 KiB of W operands
constexpr int ABYTES = 4 * 2 * 64 * 16;                      // a producer's A operands of one step: 4 tiles x (hi, lo) x 64 lanes x 16 B

struct UArgs {
  const float* x;
  const unsigned char* img;   // one chunk image
  float* out;
  int nsteps;
};

__device__ __forceinline__ void fill_image(unsigned char* lds, const unsigned char* img, int tid, int nthreads) {
  for (int i = tid * 16; i < CB; i += nthreads * 16) *reinterpret_cast<f4_t*>(lds + i) = *reinterpret_cast<const f4_t*>(img + i);
}

// distances / basis argument of the lane's 8 centres for NTILE query tiles, 16 transcendentals per 2 tiles, hi-lo split
template <int NTILE>
__device__ __forceinline__ void valu_step(const unsigned char* cur, const float (&xq)[NTILE][DC], int g, h8_t (&ah)[NTILE], h8_t (&al)[NTILE]) {
#pragma unroll
  for (int tp = 0; tp < NTILE; tp += 2) {
    float t16[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* rp = reinterpret_cast<const float*>(cur) + (8 * g + j) * RF;
      float r[RF];
#pragma unroll
      for (int v = 0; v < RF / 4; ++v) {
        const f4_t rr = *reinterpret_cast<const f4_t*>(rp + 4 * v);
        r[4 * v] = rr.x; r[4 * v + 1] = rr.y; r[4 * v + 2] = rr.z; r[4 * v + 3] = rr.w;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float r2 = 0.0f;
#pragma unroll
        for (int d = 0; d < DC; ++d) {
          const float df = xq[tp + t][d] - r[d];
          r2 = __builtin_fmaf(df, df, r2);
        }
        t16[t * 8 + j] = f16_arg<BC>(r2, r[RF - 1]);
      }
    }
    trans_block<BC, 16>(t16);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned wh[4], wl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) split_pair_f16<3, true>(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
      ah[tp + t] = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
      al[tp + t] = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
    }
  }
}

template <int NTILE>
__device__ __forceinline__ void mfma_step(const unsigned char* prv, int lane, const h8_t (&ah)[NTILE], const h8_t (&al)[NTILE],
                                          f4_t (&acc)[NTILE][NT], f4_t (&acl)[NTILE][NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const h8_t bh = *reinterpret_cast<const h8_t*>(prv + RECB + j * 2 * kF16WBytes + lane * 16);
    const h8_t bl = *reinterpret_cast<const h8_t*>(prv + RECB + j * 2 * kF16WBytes + kF16WBytes + lane * 16);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][j], 0, 0, 0);
      acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acl[t][j], 0, 0, 0);
      acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acl[t][j], 0, 0, 0);
    }
  }
}

// MODE 0: both pipes; 1: VALU work only; 2: MFMAs only (ablations of the interleaved form)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_interleaved(const UArgs a) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  fill_image(lds, a.img, tid, 512);
  float xq[2][DC];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int d = 0; d < DC; ++d) xq[t][d] = a.x[((blockIdx.x * 8 + (tid >> 6)) * 32 + t * 16 + n) * DC + d];
  f4_t acc[2][NT], acl[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc[t][j] = f4_t{0, 0, 0, 0}; acl[t][j] = f4_t{0, 0, 0, 0}; }
  h8_t ah[2], al[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ah[0][j] = 1; ah[1][j] = 1; al[0][j] = 1; al[1][j] = 1; }
  __syncthreads();
  for (int i = 0; i < a.nsteps; ++i) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    h8_t nh[2] = {ah[0], ah[1]}, nl[2] = {al[0], al[1]};
    if constexpr (MODE != 2) valu_step<2>(lds, xq, g, nh, nl);
    if constexpr (MODE != 1) mfma_step<2>(lds, lane, ah, al, acc, acl);     // the previous step's operands (deferred, as in the kernel)
#pragma unroll
    for (int t = 0; t < 2; ++t) { ah[t] = nh[t]; al[t] = nl[t]; }
  }
  float s = 0.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) s += acc[t][j][0] + acl[t][j][1];
  s += (float)ah[0][0] + (float)al[1][3];
  a.out[blockIdx.x * 512 + tid] = s;
}

__global__ __launch_bounds__(512, 2) void k_specialised(const UArgs a) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  const int pair = wave & 3;
  unsigned char* abuf = lds + CB + pair * 2 * ABYTES;        // [2][4 tiles][hi, lo][64 lanes][16 B]
  fill_image(lds, a.img, tid, 512);
  float s = 0.0f;
  if (producer) {
    float xq[4][DC];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int d = 0; d < DC; ++d) xq[t][d] = a.x[((blockIdx.x * 4 + pair) * 64 + t * 16 + n) * DC + d];
    __syncthreads();
    for (int i = 0; i < a.nsteps; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      h8_t ah[4], al[4];
      valu_step<4>(lds, xq, g, ah, al);
      unsigned char* dst = abuf + (i & 1) * ABYTES + lane * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        *reinterpret_cast<h8_t*>(dst + (t * 2 + 0) * 1024) = ah[t];
        *reinterpret_cast<h8_t*>(dst + (t * 2 + 1) * 1024) = al[t];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    s = xq[0][0];
  } else {
    f4_t acc[4][NT], acl[4][NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) { acc[t][j] = f4_t{0, 0, 0, 0}; acl[t][j] = f4_t{0, 0, 0, 0}; }
    __syncthreads();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // step 0: nothing to multiply yet
    for (int i = 1; i <= a.nsteps; ++i) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // 224 accumulator registers leave no room for the A operands of all four tiles (the first version spilled 13 dwords):
      // column tile outer, the tile's A operands re-read from LDS behind it (conflict-free 16-byte reads)
      const unsigned char* src = abuf + ((i - 1) & 1) * ABYTES + lane * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const h8_t bh = *reinterpret_cast<const h8_t*>(lds + RECB + j * 2 * kF16WBytes + lane * 16);
        const h8_t bl = *reinterpret_cast<const h8_t*>(lds + RECB + j * 2 * kF16WBytes + kF16WBytes + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const h8_t ah = *reinterpret_cast<const h8_t*>(src + (t * 2 + 0) * 1024);
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t][j], 0, 0, 0);
          acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acl[t][j], 0, 0, 0);
          const h8_t al = *reinterpret_cast<const h8_t*>(src + (t * 2 + 1) * 1024);
          acl[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acl[t][j], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < NT; ++j) s += acc[t][j][0] + acl[t][j][1];
  }
  a.out[blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, nsteps = argc > 2 ? atoi(argv[2]) : 64, reps = 20;
  const int only = argc > 3 ? atoi(argv[3]) : -1;            // run ONE variant (for rocprofv3 --pmc passes): 0 I, 1 S, 2 VALU-only, 3 MFMA-only
  float *x, *out;
  unsigned char* img;
  hipMalloc(&x, (size_t)blocks * 256 * DC * 4 + 4096);
  hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipMalloc(&img, CB);
  {
    float* hx = (float*)malloc((size_t)blocks * 256 * DC * 4);
    for (size_t i = 0; i < (size_t)blocks * 256 * DC; ++i) hx[i] = (float)(rand() % 1000) * 7e-3f;
    hipMemcpy(x, hx, (size_t)blocks * 256 * DC * 4, hipMemcpyHostToDevice);
    unsigned char* hi = (unsigned char*)malloc(CB);
    float* rec = (float*)hi;
    for (int c = 0; c < 32; ++c) {
      for (int d = 0; d < 7; ++d) rec[c * 8 + d] = (float)(rand() % 1000) * 7e-3f;
      rec[c * 8 + 7] = -0.05f;
    }
    _Float16* w = (_Float16*)(hi + RECB);
    for (int i = 0; i < NT * 2 * kF16WBytes / 2; ++i) w[i] = (_Float16)((rand() % 2000 - 1000) * 1e-3f);
    hipMemcpy(img, hi, CB, hipMemcpyHostToDevice);
  }
  UArgs a{x, img, out, nsteps};
  const size_t ldsI = CB, ldsS = CB + 4 * 2 * ABYTES;
  hipFuncSetAttribute((const void*)k_specialised, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsS);
  auto launch = [&](int v) {
    if (v == 0) hipLaunchKernelGGL(k_interleaved<0>, dim3(blocks), dim3(512), ldsI, 0, a);
    else if (v == 1) hipLaunchKernelGGL(k_specialised, dim3(blocks), dim3(512), ldsS, 0, a);
    else if (v == 2) hipLaunchKernelGGL(k_interleaved<1>, dim3(blocks), dim3(512), ldsI, 0, a);
    else hipLaunchKernelGGL(k_interleaved<2>, dim3(blocks), dim3(512), ldsI, 0, a);
  };
  const char* names[4] = {"interleaved (today)", "specialised (producer / consumer waves)", "interleaved, VALU work only", "interleaved, MFMAs only"};
  if (only >= 0) {
    for (int r = 0; r < 30; ++r) launch(only);
    hipDeviceSynchronize();
    printf("ran 30 x %s\n", names[only]);
    return 0;
  }
  for (int v = 0; v < 4; ++v) { launch(v); }
  hipDeviceSynchronize();
  printf("blocks %d x 512 threads, %d steps of 32 centres, NT = %d column tiles (= %d queries against %d centres, O = 100)\n", blocks, nsteps,
         NT, blocks * 256, nsteps * 32);
  for (int round = 0; round < 3; ++round) {
    for (int v = 0; v < 4; ++v) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) launch(v);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("round %d  %-42s %8.1f us per launch\n", round, names[v], ms / reps * 1e3);
    }
  }
  return 0;
}
