// The exact bank-aware 2-centre asm body of rbf_fwd_qlane (D=7, O=10), no memory traffic: where do the
// cycles go?  Parts: DIST (28 instr), EXP (2 mul + 2 exp), W (20 fmac with SGPR operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 3000
struct SA { float a[18]; float b[18]; };
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v45","v46"
#define DIM(j, xa, da, db) "v_subrev_f32_e32 " da ", %[a" #j "], " xa "\n v_subrev_f32_e32 " db ", %[b" #j "], " xa "\n v_fmac_f32_e32 v28, " da ", " da "\n v_fmac_f32_e32 v36, " db ", " db "\n"
#define SOPS [a0] "s"(s.a[0]), [a1] "s"(s.a[1]), [a2] "s"(s.a[2]), [a3] "s"(s.a[3]), [a4] "s"(s.a[4]), [a5] "s"(s.a[5]), [a6] "s"(s.a[6]), [a7] "s"(s.a[7]), \
             [b0] "s"(s.b[0]), [b1] "s"(s.b[1]), [b2] "s"(s.b[2]), [b3] "s"(s.b[3]), [b4] "s"(s.b[4]), [b5] "s"(s.b[5]), [b6] "s"(s.b[6]), [b7] "s"(s.b[7])
#define WOPS [a0] "s"(s.a[8]), [a1] "s"(s.a[9]), [a2] "s"(s.a[10]), [a3] "s"(s.a[11]), [a4] "s"(s.a[12]), [a5] "s"(s.a[13]), [a6] "s"(s.a[14]), [a7] "s"(s.a[15]), [a8] "s"(s.a[16]), [a9] "s"(s.a[17]), \
             [b0] "s"(s.b[8]), [b1] "s"(s.b[9]), [b2] "s"(s.b[10]), [b3] "s"(s.b[11]), [b4] "s"(s.b[12]), [b5] "s"(s.b[13]), [b6] "s"(s.b[14]), [b7] "s"(s.b[15]), [b8] "s"(s.b[16]), [b9] "s"(s.b[17])
template <int M>
__global__ __launch_bounds__(1024) void k(float* out, SA s) {
  asm volatile("v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n"
               "v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v41, 0\n"
               "v_mov_b32 v28, 0\n v_mov_b32 v36, 0\n v_mov_b32 v32, 0.5\n v_mov_b32 v40, 0.5\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n" ::: CLOB);
  for (int it = 0; it < ITERS; ++it) {
    if (M & 1) asm volatile("v_subrev_f32_e32 v42, %[a0], v20\n v_subrev_f32_e32 v45, %[b0], v20\n v_mul_f32_e32 v28, v42, v42\n v_mul_f32_e32 v36, v45, v45\n"
        DIM(1, "v21", "v43", "v46") DIM(2, "v22", "v42", "v45") DIM(3, "v23", "v43", "v46") DIM(4, "v24", "v42", "v45") DIM(5, "v25", "v43", "v46") DIM(6, "v26", "v42", "v45")
        "v_mul_f32_e32 v28, %[a7], v28\n v_mul_f32_e32 v36, %[b7], v36\n" :: SOPS : CLOB);
    if (M & 2) asm volatile("v_exp_f32_e32 v32, v28\n v_exp_f32_e32 v40, v36\n" ::: CLOB);
    if (M & 4) asm volatile(
        "v_fmac_f32_e32 v29, %[a0], v32\n v_fmac_f32_e32 v30, %[a1], v32\n v_fmac_f32_e32 v31, %[a2], v32\n v_fmac_f32_e32 v33, %[a3], v32\n v_fmac_f32_e32 v34, %[a4], v32\n"
        "v_fmac_f32_e32 v35, %[a5], v32\n v_fmac_f32_e32 v37, %[a6], v32\n v_fmac_f32_e32 v38, %[a7], v32\n v_fmac_f32_e32 v39, %[a8], v32\n v_fmac_f32_e32 v41, %[a9], v32\n"
        "v_fmac_f32_e32 v29, %[b0], v40\n v_fmac_f32_e32 v30, %[b1], v40\n v_fmac_f32_e32 v31, %[b2], v40\n v_fmac_f32_e32 v33, %[b3], v40\n v_fmac_f32_e32 v34, %[b4], v40\n"
        "v_fmac_f32_e32 v35, %[b5], v40\n v_fmac_f32_e32 v37, %[b6], v40\n v_fmac_f32_e32 v38, %[b7], v40\n v_fmac_f32_e32 v39, %[b8], v40\n v_fmac_f32_e32 v41, %[b9], v40\n" :: WOPS : CLOB);
    if (M & 8) asm volatile(   // W part with only 2 distinct SGPRs (is it the SGPR variety?)
        "v_fmac_f32_e32 v29, %[a0], v32\n v_fmac_f32_e32 v30, %[a1], v32\n v_fmac_f32_e32 v31, %[a0], v32\n v_fmac_f32_e32 v33, %[a1], v32\n v_fmac_f32_e32 v34, %[a0], v32\n"
        "v_fmac_f32_e32 v35, %[a1], v32\n v_fmac_f32_e32 v37, %[a0], v32\n v_fmac_f32_e32 v38, %[a1], v32\n v_fmac_f32_e32 v39, %[a0], v32\n v_fmac_f32_e32 v41, %[a1], v32\n"
        "v_fmac_f32_e32 v29, %[b0], v40\n v_fmac_f32_e32 v30, %[b1], v40\n v_fmac_f32_e32 v31, %[b0], v40\n v_fmac_f32_e32 v33, %[b1], v40\n v_fmac_f32_e32 v34, %[b0], v40\n"
        "v_fmac_f32_e32 v35, %[b1], v40\n v_fmac_f32_e32 v37, %[b0], v40\n v_fmac_f32_e32 v38, %[b1], v40\n v_fmac_f32_e32 v39, %[b0], v40\n v_fmac_f32_e32 v41, %[b1], v40\n" :: WOPS : CLOB);
    if (M & 16) asm volatile(  // W part with VGPR weights (v20..v26 as stand-ins), banks: acc 1-3, phi 0, w any
        "v_fmac_f32_e32 v29, v22, v32\n v_fmac_f32_e32 v30, v23, v32\n v_fmac_f32_e32 v31, v21, v32\n v_fmac_f32_e32 v33, v22, v32\n v_fmac_f32_e32 v34, v23, v32\n"
        "v_fmac_f32_e32 v35, v21, v32\n v_fmac_f32_e32 v37, v22, v32\n v_fmac_f32_e32 v38, v23, v32\n v_fmac_f32_e32 v39, v21, v32\n v_fmac_f32_e32 v41, v22, v32\n"
        "v_fmac_f32_e32 v29, v22, v40\n v_fmac_f32_e32 v30, v23, v40\n v_fmac_f32_e32 v31, v21, v40\n v_fmac_f32_e32 v33, v22, v40\n v_fmac_f32_e32 v34, v23, v40\n"
        "v_fmac_f32_e32 v35, v21, v40\n v_fmac_f32_e32 v37, v22, v40\n v_fmac_f32_e32 v38, v23, v40\n v_fmac_f32_e32 v39, v21, v40\n v_fmac_f32_e32 v41, v22, v40\n" ::: CLOB);
  }
  float r; asm volatile("v_add_f32 %0, v29, v41\n v_add_f32 %0, %0, v32\n v_add_f32 %0, %0, v28" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int M> void run(const char* name, float* out, int ninstr) {
  SA s; for (int i = 0; i < 18; ++i) { s.a[i] = 0.01f * (i + 1); s.b[i] = 0.02f * (i + 1); } s.a[7] = -0.01f; s.b[7] = -0.02f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<M><<<512, 1024>>>(out, s); hipDeviceSynchronize();
  hipEventRecord(e0); k<M><<<512, 1024>>>(out, s); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * 8);
  printf("%-44s %7.3f ms  %6.1f cyc@2.4 per 2 pair-waves  (%2d instr -> %.2f cyc/instr)\n", name, ms, cyc, ninstr, cyc / ninstr);
}
int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<1>("DIST only (30 instr)", out, 30);
  run<2>("EXP only (2)", out, 2);
  run<4>("W only, 20 fmac distinct SGPRs", out, 20);
  run<8>("W only, 20 fmac 2 SGPRs each", out, 20);
  run<16>("W only, 20 fmac VGPR weights", out, 20);
  run<3>("DIST + EXP", out, 32);
  run<7>("DIST + EXP + W (full)", out, 52);
  run<19>("DIST + EXP + W(VGPR)", out, 52);
  return 0;
}
