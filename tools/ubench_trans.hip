// Transcendental-op hazard probe for gfx950: what does ONE v_exp_f32 cost inside a stream of independent
// v_fmac_f32, as a function of the distance to its producer and to its consumer?  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 500
#define U8(X) X X X X X X X X
#define F0 "v_fmac_f32_e32 v100, v105, v110\n"
#define F1 "v_fmac_f32_e32 v101, v106, v111\n"
#define F2 "v_fmac_f32_e32 v102, v107, v108\n"
#define F3 "v_fmac_f32_e32 v103, v104, v109\n"
#define F4 F0 F1 F2 F3
#define F8 F4 F4
#define F16 F8 F8
#define MUL "v_mul_f32_e32 v121, v105, v110\n"
#define EXPI "v_exp_f32_e32 v120, v105\n"          /* independent input */
#define EXPD "v_exp_f32_e32 v120, v121\n"          /* input = MUL's result */
#define USE "v_fmac_f32_e32 v116, v120, v110\n"    /* consumer of the exp */
#define RCPI "v_rcp_f32_e32 v120, v105\n"
#define CLOB "v100","v101","v102","v103","v116","v120","v121"

template <int M>
__global__ __launch_bounds__(1024) void k(float* out) {
  asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n"
               "v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
               "v_mov_b32 v108, 0.25\n v_mov_b32 v109, 0.25\n v_mov_b32 v110, 0.25\n v_mov_b32 v111, 0.25\n"
               "v_mov_b32 v116, 1.0\n v_mov_b32 v120, 1.0\n v_mov_b32 v121, 1.0\n"
               ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v116","v120","v121");
  for (int it = 0; it < ITERS; ++it) {
    if (M == 0) asm volatile(U8(F16) ::: CLOB);
    if (M == 1) asm volatile(U8(F8 EXPI F8) ::: CLOB);
    if (M == 2) asm volatile(U8(F8 MUL EXPD F8) ::: CLOB);
    if (M == 3) asm volatile(U8(F4 MUL F4 EXPD F8) ::: CLOB);
    if (M == 4) asm volatile(U8(F8 EXPI USE F8) ::: CLOB);
    if (M == 5) asm volatile(U8(F4 EXPI F4 USE F8) ::: CLOB);
    if (M == 6) asm volatile(U8(F8 MUL EXPD USE F8) ::: CLOB);
    if (M == 7) asm volatile(U8(F4 MUL F4 EXPD F4 USE F4) ::: CLOB);
    if (M == 8) asm volatile(U8(F8 EXPI EXPI F8) ::: CLOB);
    if (M == 9) asm volatile(U8(F4 EXPI F4 EXPI F4 EXPI F4 EXPI) ::: CLOB);
    if (M == 10) asm volatile(U8(F8 RCPI F8) ::: CLOB);
    if (M == 11) asm volatile(U8(F4 MUL F8 EXPD F4) ::: CLOB);
    if (M == 12) asm volatile(U8(F4 EXPI F8 USE F4) ::: CLOB);
    if (M == 13) asm volatile(U8(EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI EXPI) ::: CLOB);
  }
  float r;
  asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102\n v_add_f32 %0, %0, v103\n v_add_f32 %0, %0, v116\n v_add_f32 %0, %0, v120" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int M>
void run(const char* name, float* out, int blocks, double base) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<M><<<blocks, 1024>>>(out); hipDeviceSynchronize();
  hipEventRecord(e0); k<M><<<blocks, 1024>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wps = blocks * 16.0 / 1024.0;       // waves per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * 8 * wps);
  printf("%-52s wps=%2.0f %7.3f ms  %6.1f cyc@2.4 per body per SIMD  (extra over 16 fillers: %5.1f)\n", name, wps, ms, cyc, cyc - base);
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  for (int blocks : {512, 256}) {
    const double b = blocks == 512 ? 38.9 : 0;     // printed relative to the 8-wps filler baseline measured first
    run<0>("16 fillers", out, blocks, 0);
    run<1>("+ exp (independent)", out, blocks, b);
    run<2>("+ mul -> exp adjacent", out, blocks, b);
    run<3>("+ mul, 4 fillers, exp", out, blocks, b);
    run<11>("+ mul, 8 fillers, exp", out, blocks, b);
    run<4>("+ exp -> use adjacent", out, blocks, b);
    run<5>("+ exp, 4 fillers, use", out, blocks, b);
    run<12>("+ exp, 8 fillers, use", out, blocks, b);
    run<6>("+ mul -> exp -> use adjacent", out, blocks, b);
    run<7>("+ mul, 4, exp, 4, use", out, blocks, b);
    run<8>("+ 2 exps adjacent", out, blocks, b);
    run<9>("+ 4 exps spread", out, blocks, b);
    run<10>("+ rcp (independent)", out, blocks, b);
    run<13>("16 exps only", out, blocks, 0);
  }
  return 0;
}
