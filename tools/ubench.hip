// VALU issue-rate microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32 (+ SGPR operand
// forms), at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 64
#define ITERS 2000

template <int MODE>
__global__ void k(float* out, float sv, unsigned long long* cyc) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (MODE == 0) {   // v_fma_f32, VGPR operands, 8 independent chains
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 1) {   // v_fma_f32 with an SGPR operand
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sv));
      } else if (MODE == 2) {   // v_pk_fma_f32 (4 independent register pairs)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                     "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));
      } else if (MODE == 3) {   // v_exp_f32
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if (MODE == 4) {   // v_sub_f32 with SGPR + v_fmac (the distance pattern)
        asm volatile("v_subrev_f32 %0, %8, %1\n v_fmac_f32 %2, %0, %0\n v_subrev_f32 %3, %8, %4\n v_fmac_f32 %5, %3, %3\n"
                     "v_subrev_f32 %0, %8, %6\n v_fmac_f32 %2, %0, %0\n v_subrev_f32 %3, %8, %7\n v_fmac_f32 %5, %3, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sv));
      } else if (MODE == 5) {   // v_pk_add_f32 + v_pk_fma_f32 (packed distance pattern), SGPR pair operand
        asm volatile("v_pk_add_f32 %0, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %2, %0, %0, %2\n"
                     "v_pk_add_f32 %0, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %2, %0, %0, %2\n"
                     "v_pk_add_f32 %0, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %2, %0, %0, %2\n"
                     "v_pk_add_f32 %0, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_fma_f32 %2, %0, %0, %2\n"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "s"(*(double*)&sv));
      } else if (MODE == 7) {   // v_fmac_f32_e32 VOP2, VGPR only
        asm volatile("v_fmac_f32_e32 %0, %8, %8\n v_fmac_f32_e32 %1, %8, %8\n v_fmac_f32_e32 %2, %8, %8\n v_fmac_f32_e32 %3, %8, %8\n"
                     "v_fmac_f32_e32 %4, %8, %8\n v_fmac_f32_e32 %5, %8, %8\n v_fmac_f32_e32 %6, %8, %8\n v_fmac_f32_e32 %7, %8, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 8) {   // v_fmac_f32_e32 VOP2, SGPR src0 (same SGPR)
        asm volatile("v_fmac_f32_e32 %0, %9, %8\n v_fmac_f32_e32 %1, %9, %8\n v_fmac_f32_e32 %2, %9, %8\n v_fmac_f32_e32 %3, %9, %8\n"
                     "v_fmac_f32_e32 %4, %9, %8\n v_fmac_f32_e32 %5, %9, %8\n v_fmac_f32_e32 %6, %9, %8\n v_fmac_f32_e32 %7, %9, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sv));
      } else if (MODE == 9) {   // v_fmac_f32_e32 VOP2, 4 different SGPRs
        asm volatile("v_fmac_f32_e32 %0, %9, %8\n v_fmac_f32_e32 %1, %10, %8\n v_fmac_f32_e32 %2, %11, %8\n v_fmac_f32_e32 %3, %12, %8\n"
                     "v_fmac_f32_e32 %4, %9, %8\n v_fmac_f32_e32 %5, %10, %8\n v_fmac_f32_e32 %6, %11, %8\n v_fmac_f32_e32 %7, %12, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(sv), "s"(sv*2), "s"(sv*3), "s"(sv*4));
      } else if (MODE == 10) {   // v_subrev_f32_e32 VOP2 with SGPR
        asm volatile("v_subrev_f32_e32 %0, %8, %0\n v_subrev_f32_e32 %1, %8, %1\n v_subrev_f32_e32 %2, %8, %2\n v_subrev_f32_e32 %3, %8, %3\n"
                     "v_subrev_f32_e32 %4, %8, %4\n v_subrev_f32_e32 %5, %8, %5\n v_subrev_f32_e32 %6, %8, %6\n v_subrev_f32_e32 %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sv));
      } else if (MODE == 11) {   // v_add_f32_e32 VOP2 VGPR only
        asm volatile("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n"
                     "v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 12) {   // v_pk_fma_f32 with SGPR pair broadcast operand
        asm volatile("v_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %5, %4, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %5, %4, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, %4, %3 op_sel_hi:[0,1,1]\n"
                     "v_pk_fma_f32 %0, %5, %4, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %5, %4, %1 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %5, %4, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %5, %4, %3 op_sel_hi:[0,1,1]\n"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "s"(*(double*)&sv), "v"(*(double*)&b));
      } else if (MODE == 13) {   // v_mul_f32 + v_exp_f32 + 2 fmac mix
        asm volatile("v_mul_f32_e32 %0, %8, %0\n v_exp_f32_e32 %1, %0\n v_fmac_f32_e32 %2, %1, %1\n v_fmac_f32_e32 %3, %1, %1\n"
                     "v_mul_f32_e32 %4, %8, %4\n v_exp_f32_e32 %5, %4\n v_fmac_f32_e32 %6, %5, %5\n v_fmac_f32_e32 %7, %5, %5\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
      } else if (MODE == 6) {   // v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[4096 + blockIdx.x] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int lanes_per_instr_ops) {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 8 * 1024 * sizeof(float) * 4);
  hipMalloc(&cyc, 8192 * sizeof(unsigned long long));
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD: block = 256*wps threads, 1 block per CU
    int threads = 256 * wps;
    if (threads > 1024) { threads = 1024; }
    int blocks = 256 * (256 * wps / threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8192);
    hipMemcpy(h.data(), cyc, 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0, rt = 0; for (int i = 0; i < blocks; ++i) { avg += h[i]; rt += h[4096 + i]; } avg /= blocks; rt /= blocks;
    double mhz = avg / (rt / 100.0);   // memrealtime = 100 MHz
    double instr_per_wave = (double)ITERS * REP;
    // memtime ticks at 100 MHz constant? report both ticks per instr and wall-derived cycles at 2.4 GHz
    double waves_per_simd = wps;
    double wall_cycles = ms * 1e-3 * 2.4e9;
    printf("%-30s wps=%d wall %.3f ms  %.2f cyc@2.4/instr/SIMD  memtime/instr/wave %.2f  clk(memtime/realtime) %.0f MHz\n", name, wps, ms,
           wall_cycles / (instr_per_wave * waves_per_simd), avg / instr_per_wave, mhz);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_fma_f32 vgpr", 1);
  run<1>("v_fma_f32 sgpr-operand", 1);
  run<2>("v_pk_fma_f32", 2);
  run<6>("v_pk_mul_f32", 2);
  run<3>("v_exp_f32", 1);
  run<4>("v_subrev(sgpr)+v_fmac", 1);
  run<5>("v_pk_add(sgpr)+v_pk_fma", 2);
  run<7>("v_fmac_e32 vgpr", 1);
  run<8>("v_fmac_e32 sgpr src0 (same)", 1);
  run<9>("v_fmac_e32 sgpr src0 (4 diff)", 1);
  run<10>("v_subrev_e32 sgpr", 1);
  run<11>("v_add_e32 vgpr", 1);
  run<12>("v_pk_fma sgprpair bcast", 2);
  run<13>("mul+exp+2fmac mix", 1);
  return 0;
}
