// Issue cost of v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16 on gfx950, and of the packed f32 VALU forms, one wave
// per SIMD and four: cycles per instruction from s_memtime.  hipcc --offload-arch=gfx950 -O2 -o tools/_ubench_mfma_rate tools/ubench_mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  const int l = threadIdx.x;
  h8_t a8, b8; h4_t a4, b4;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(l * 0.01f + j); b8[j] = (_Float16)(1.0f + j); }
  for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
  f4_t c[8];
  for (int i = 0; i < 8; ++i) c[i] = f4_t{0, 0, 0, 0};
  f2_t p[8];
  for (int i = 0; i < 8; ++i) p[i] = f2_t{l * 1.0f + i, l * 2.0f + i};
  float s[16];
  for (int i = 0; i < 16; ++i) s[i] = l + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[i], 0, 0, 0);
    } else if (MODE == 2) {          // 8 packed multiplies (16 values)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    } else if (MODE == 3) {          // 16 plain multiplies
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
    } else if (MODE == 4) {          // 16 exps back to back
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
    } else if (MODE == 5) {          // 8 MFMA x32 with 16 plain multiplies interleaved (two per MFMA)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[2 * i]) : "v"(s[(2 * i + 1) & 15]));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[2 * i + 1]) : "v"(s[(2 * i + 2) & 15]));
      }
    } else if (MODE == 6) {          // 8 MFMA x32 with 32 plain multiplies interleaved (four per MFMA)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[i], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[(4 * i + m) & 15]) : "v"(s[(4 * i + m + 1) & 15]));
      }
    } else if (MODE == 7) {          // v_cvt_pkrtz x 16
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
    } else if (MODE == 8) {          // v_fma_mix_f32 x 16
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_mix_f32 %0, %0, %1, %0 op_sel_hi:[1,0,0]" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][1] + c[i][2] + c[i][3] + p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) acc += s[i];
  out[blockIdx.x * blockDim.x + l] = acc;
  if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int block) {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, out, cyc, iters);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s block %4d: %.2f clk per instruction (s_memtime clk = 100 MHz ticks? raw %lld)\n", name, block, (double)h / iters / per_iter, h);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int block : {64, 256, 512}) {      // 1, 1 (x4 SIMDs), 2 waves per SIMD
    run<0>("mfma_f32_16x16x32_f16 (8 independent)", 8, block);
    run<1>("mfma_f32_16x16x16_f16 (8 independent)", 8, block);
    run<2>("v_pk_mul_f32", 8, block);
    run<3>("v_mul_f32", 16, block);
    run<4>("v_exp_f32 back to back", 16, block);
    run<5>("mfma x32 + 2 v_mul each (per mfma)", 8, block);
    run<6>("mfma x32 + 4 v_mul each (per mfma)", 8, block);
    run<7>("v_cvt_pkrtz_f16_f32", 16, block);
    run<8>("v_fma_mix_f32", 16, block);
  }
  return 0;
}
