// Do f32 / f64 MFMAs co-execute with VALU work on gfx950?  Times MFMA-only, VALU-only and an interleaved
// mix at 8 waves per SIMD; overlap => t(mix) ~ max, no overlap => t(mix) ~ sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define ITERS 3000

template <int MODE, int NV>   // MODE 0: f32 16x16x4, 1: f64 16x16x4, 2: bf16 16x16x32 ; NV = VALU fmacs per MFMA
__global__ __launch_bounds__(1024) void k(float* out, float sv, int do_mfma, int do_valu) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  f64x4 dac0 = {0, 0, 0, 0}, dac1 = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  double da = a, db = b;
  typedef short s8 __attribute__((ext_vector_type(8)));
  s8 ha = {1, 2, 3, 4, 5, 6, 7, 8}, hb = {1, 1, 1, 1, 1, 1, 1, 1};
  float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
  for (int it = 0; it < ITERS; ++it) {
    if (do_mfma) {
      if (MODE == 0) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); }
      if (MODE == 1) { dac0 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, dac0, 0, 0, 0); }
      if (MODE == 2) { acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc0, 0, 0, 0); }
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < NV / 8; ++r)
        asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                     "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a));
    }
    if (do_mfma) {
      if (MODE == 0) { acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0); }
      if (MODE == 1) { dac1 = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, dac1, 0, 0, 0); }
      if (MODE == 2) { acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc1, 0, 0, 0); }
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < NV / 8; ++r)
        asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                     "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(b), "v"(a));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + (float)(dac0[0] + dac1[1]) + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int MODE, int NV>
void run(const char* name, float* out) {
  float t[3];
  for (int c = 0; c < 3; ++c) {
    int dm = c != 1, dv = c != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV><<<512, 1024>>>(out, 1.0f, dm, dv); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE, NV><<<512, 1024>>>(out, 1.0f, dm, dv); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&t[c], e0, e1);
  }
  double per = 1e-3 * 2.4e9 / ((double)ITERS * 2 * 8);   // cycles@2.4 per (1 MFMA + NV VALU) per SIMD
  printf("%-28s NV=%2d: mfma-only %6.1f  valu-only %6.1f  mix %6.1f cyc@2.4 per group (sum %6.1f, max %6.1f)\n", name, NV,
         t[0] * per, t[1] * per, t[2] * per, (t[0] + t[1]) * per, (t[0] > t[1] ? t[0] : t[1]) * per);
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0, 8>("f32 mfma 16x16x4", out);
  run<0, 16>("f32 mfma 16x16x4", out);
  run<1, 16>("f64 mfma 16x16x4", out);
  run<1, 32>("f64 mfma 16x16x4", out);
  run<2, 8>("bf16 mfma 16x16x32", out);
  run<2, 16>("bf16 mfma 16x16x32", out);
  return 0;
}
