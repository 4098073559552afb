// What does one v_mfma_f32_16x16x32_{f16,bf16} cost inside a stream of plain VALU (large unrolled body)?
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
#define MH0 "v_mfma_f32_16x16x32_f16 a[0:3], v[112:115], v[116:119], a[0:3]\n"
#define MH1 "v_mfma_f32_16x16x32_f16 a[4:7], v[112:115], v[116:119], a[4:7]\n"
#define MB0 "v_mfma_f32_16x16x32_bf16 a[0:3], v[112:115], v[116:119], a[0:3]\n"
#define MB1 "v_mfma_f32_16x16x32_bf16 a[4:7], v[112:115], v[116:119], a[4:7]\n"
#define MV0 "v_mfma_f32_16x16x32_f16 v[60:63], v[112:115], v[116:119], v[60:63]\n"
#define MV1 "v_mfma_f32_16x16x32_f16 v[64:67], v[112:115], v[116:119], v[64:67]\n"
#define CLOB "v100","v101","v102","v103","a0","a1","a2","a3","a4","a5","a6","a7","v60","v61","v62","v63","v64","v65","v66","v67"

template <int M>
__global__ __launch_bounds__(1024) void k(float* out) {
  asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n"
               "v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
               "v_mov_b32 v108, 0.25\n v_mov_b32 v109, 0.25\n v_mov_b32 v110, 0.25\n v_mov_b32 v111, 0.25\n"
               "v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n"
               "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n"
               "v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
               "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"
               ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119",CLOB);
  for (int it = 0; it < ITERS; ++it) {
    if (M == 0) asm volatile(U8(F16) ::: CLOB);
    if (M == 1) asm volatile(U8(MH0 MH1) ::: CLOB);                 // 2 MFMAs per body, nothing else
    if (M == 2) asm volatile(U8(F8 MH0 F8) ::: CLOB);               // 16 fmacs + 1 f16 MFMA
    if (M == 3) asm volatile(U8(F4 MH0 F8 MH1 F4) ::: CLOB);        // 16 fmacs + 2
    if (M == 4) asm volatile(U8(F16 F8 MH0 F8) ::: CLOB);           // 32 fmacs + 1
    if (M == 5) asm volatile(U8(F8 MB0 F8) ::: CLOB);               // bf16
    if (M == 6) asm volatile(U8(MB0 MB1) ::: CLOB);
    if (M == 7) asm volatile(U8(F8 MV0 F8) ::: CLOB);               // accumulator in VGPRs
    if (M == 8) asm volatile(U8(MH0 MH0) ::: CLOB);                 // dependent chain on one accumulator
    if (M == 9) asm volatile(U8(F4 MH0 F4 MH0 F4 MH1 F4 MH1) ::: CLOB);  // 16 fmacs + 4 (2 chains)
  }
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a0\n v_add_f32 %0, %0, v100\n v_add_f32 %0, %0, v101\n v_add_f32 %0, %0, v102\n v_add_f32 %0, %0, v103\n v_add_f32 %0, %0, v60" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int M>
void run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<M><<<blocks, 1024>>>(out); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<M><<<blocks, 1024>>>(out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wps = blocks * 16.0 / 1024.0;
  printf("%-52s wps=%2.0f %7.3f ms  %6.1f cyc@2.4 per body per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 8 * wps));
}

int main() {
  float* out; (void)hipMalloc(&out, 512 * 1024 * 4);
  for (int blocks : {512, 256}) {
    run<0>("16 fmacs", out, blocks);
    run<1>("2 f16 MFMA 16x16x32 only (2 accumulators)", out, blocks);
    run<8>("2 f16 MFMA only, one accumulator", out, blocks);
    run<6>("2 bf16 MFMA only", out, blocks);
    run<2>("16 fmacs + 1 f16 MFMA", out, blocks);
    run<5>("16 fmacs + 1 bf16 MFMA", out, blocks);
    run<7>("16 fmacs + 1 f16 MFMA (VGPR accumulator)", out, blocks);
    run<3>("16 fmacs + 2 f16 MFMA", out, blocks);
    run<9>("16 fmacs + 4 f16 MFMA", out, blocks);
    run<4>("32 fmacs + 1 f16 MFMA", out, blocks);
  }
  return 0;
}
