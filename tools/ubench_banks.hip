// VGPR bank-conflict probe for gfx950: v_fmac / v_fma with operands placed in chosen registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4000
#define R8(X) X X X X X X X X

template <int M>
__global__ __launch_bounds__(1024) void k(float* out, float sv) {
  // all registers named explicitly and clobbered; initialise them first
  asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n"
               "v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
               "v_mov_b32 v108, 0.25\n v_mov_b32 v109, 0.25\n v_mov_b32 v110, 0.25\n v_mov_b32 v111, 0.25\n"
               "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 1.0\n v_mov_b32 v114, 1.0\n v_mov_b32 v115, 1.0\n"
               ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115");
  for (int it = 0; it < ITERS; ++it) {
    if (M == 0) asm volatile(R8("v_fmac_f32_e32 v100, v104, v108\n v_fmac_f32_e32 v112, v104, v108\n") ::: "v100","v112");          // all bank 0
    if (M == 1) asm volatile(R8("v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v112, v105, v110\n") ::: "v100","v112");          // banks 0,1,2
    if (M == 2) asm volatile(R8("v_fmac_f32_e32 v100, v105, v105\n v_fmac_f32_e32 v112, v105, v105\n") ::: "v100","v112");          // src0 == src1, other bank
    if (M == 3) asm volatile(R8("v_fmac_f32_e32 v100, v104, v104\n v_fmac_f32_e32 v112, v104, v104\n") ::: "v100","v112");          // src0 == src1, same bank as dst
    if (M == 4) asm volatile(R8("v_fmac_f32_e32 v100, %0, v105\n v_fmac_f32_e32 v112, %1, v105\n") :: "s"(sv), "s"(sv * 2) : "v100","v112");   // sgpr, dst bank 0, src bank 1
    if (M == 5) asm volatile(R8("v_fmac_f32_e32 v100, %0, v104\n v_fmac_f32_e32 v112, %1, v104\n") :: "s"(sv), "s"(sv * 2) : "v100","v112");   // sgpr, dst/src same bank
    if (M == 6) asm volatile(R8("v_fmac_f32_e32 v100, %0, v105\n v_fmac_f32_e32 v101, %1, v105\n") :: "s"(sv), "s"(sv * 2) : "v100","v101");   // sgpr, dsts banks 0/1, src bank 1
    if (M == 7) asm volatile(R8("v_subrev_f32_e32 v100, %0, v105\n v_subrev_f32_e32 v112, %1, v106\n") :: "s"(sv), "s"(sv * 2) : "v100","v112"); // sub with sgpr
    if (M == 8) asm volatile(R8("v_subrev_f32_e32 v100, %0, v105\n v_fmac_f32_e32 v113, v100, v100\n") :: "s"(sv) : "v100","v113");              // sub -> dependent square (banks 0 ->1)
    if (M == 9) asm volatile(R8("v_subrev_f32_e32 v100, %0, v105\n v_fmac_f32_e32 v113, v102, v102\n") :: "s"(sv) : "v100","v113");              // sub + independent square
    if (M == 10) asm volatile(R8("v_mul_f32_e32 v100, v105, v105\n v_mul_f32_e32 v112, v106, v106\n") ::: "v100","v112");            // squares by mul
    if (M == 11) asm volatile(R8("v_add_f32_e32 v100, v105, v110\n v_add_f32_e32 v112, v105, v110\n") ::: "v100","v112");            // plain add 3 banks
  }
  float r;
  asm volatile("v_add_f32 %0, v100, v112\n v_add_f32 %0, %0, v113\n v_add_f32 %0, %0, v101" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int M>
void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<M><<<512, 1024>>>(out, 1.0001f); hipDeviceSynchronize();
  hipEventRecord(e0); k<M><<<512, 1024>>>(out, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %7.3f ms  %5.2f cyc@2.4 per instr per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 16 * 8));
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0>("fmac dst,src0,src1 all bank 0", out);
  run<1>("fmac banks 0,1,2", out);
  run<2>("fmac src0==src1 (bank 1), dst bank 0", out);
  run<3>("fmac src0==src1 same bank as dst", out);
  run<4>("fmac sgpr src0, dst bank 0, src1 bank 1", out);
  run<5>("fmac sgpr src0, dst & src1 bank 0", out);
  run<6>("fmac sgpr, alternating dst banks 0/1", out);
  run<7>("subrev sgpr", out);
  run<8>("subrev sgpr -> dependent square fmac", out);
  run<9>("subrev sgpr + independent square fmac", out);
  run<10>("mul squares", out);
  run<11>("add 3 banks", out);
  return 0;
}
