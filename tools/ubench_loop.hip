// Inner-loop instruction-sequence microbenchmark for the fused RBF forward (gfx950): the exact
// per-(query, centre) VALU sequence with SGPR operands, no memory traffic, 8 waves per SIMD.
// Reports cycles(@2.4 GHz nominal) per pair-wave per SIMD for variants of the sequence.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITERS 4000

struct SArgs { float c[7]; float sc; float w[10]; };

// two pairs per asm block (two different "centres" = the same SGPRs, distinct temporaries)
template <int V>
__global__ __launch_bounds__(1024) void k(float* out, SArgs s) {
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + .1f, x2 = x0 + .2f, x3 = x0 + .3f, x4 = x0 + .4f, x5 = x0 + .5f, x6 = x0 + .6f;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, a9 = 0;
  float d, e, r, ph;
  for (int it = 0; it < ITERS; ++it) {
#define DIST_FMAC                                                                                       \
    "v_subrev_f32_e32 %[d], %[c0], %[x0]\n v_mul_f32_e32 %[r], %[d], %[d]\n"                            \
    "v_subrev_f32_e32 %[e], %[c1], %[x1]\n v_fmac_f32_e32 %[r], %[e], %[e]\n"                           \
    "v_subrev_f32_e32 %[d], %[c2], %[x2]\n v_fmac_f32_e32 %[r], %[d], %[d]\n"                           \
    "v_subrev_f32_e32 %[e], %[c3], %[x3]\n v_fmac_f32_e32 %[r], %[e], %[e]\n"                           \
    "v_subrev_f32_e32 %[d], %[c4], %[x4]\n v_fmac_f32_e32 %[r], %[d], %[d]\n"                           \
    "v_subrev_f32_e32 %[e], %[c5], %[x5]\n v_fmac_f32_e32 %[r], %[e], %[e]\n"                           \
    "v_subrev_f32_e32 %[d], %[c6], %[x6]\n v_fmac_f32_e32 %[r], %[d], %[d]\n"
#define DIST_FMA3                                                                                       \
    "v_subrev_f32_e32 %[d], %[c0], %[x0]\n v_mul_f32_e32 %[r], %[d], %[d]\n"                            \
    "v_subrev_f32_e32 %[e], %[c1], %[x1]\n v_fma_f32 %[r], %[e], %[e], %[r]\n"                          \
    "v_subrev_f32_e32 %[d], %[c2], %[x2]\n v_fma_f32 %[r], %[d], %[d], %[r]\n"                          \
    "v_subrev_f32_e32 %[e], %[c3], %[x3]\n v_fma_f32 %[r], %[e], %[e], %[r]\n"                          \
    "v_subrev_f32_e32 %[d], %[c4], %[x4]\n v_fma_f32 %[r], %[d], %[d], %[r]\n"                          \
    "v_subrev_f32_e32 %[e], %[c5], %[x5]\n v_fma_f32 %[r], %[e], %[e], %[r]\n"                          \
    "v_subrev_f32_e32 %[d], %[c6], %[x6]\n v_fma_f32 %[r], %[d], %[d], %[r]\n"
#define DIST_DPP                                                                                        \
    "v_subrev_f32_e32 %[d], %[c0], %[x0]\n v_mul_f32_e32 %[r], %[d], %[d]\n"                            \
    "v_subrev_f32_e32 %[e], %[c1], %[x1]\n v_fmac_f32_dpp %[r], %[e], %[e] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n" \
    "v_subrev_f32_e32 %[d], %[c2], %[x2]\n v_fmac_f32_dpp %[r], %[d], %[d] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n" \
    "v_subrev_f32_e32 %[e], %[c3], %[x3]\n v_fmac_f32_dpp %[r], %[e], %[e] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n" \
    "v_subrev_f32_e32 %[d], %[c4], %[x4]\n v_fmac_f32_dpp %[r], %[d], %[d] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n" \
    "v_subrev_f32_e32 %[e], %[c5], %[x5]\n v_fmac_f32_dpp %[r], %[e], %[e] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n" \
    "v_subrev_f32_e32 %[d], %[c6], %[x6]\n v_fmac_f32_dpp %[r], %[d], %[d] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
#define EXPO "v_mul_f32_e32 %[r], %[sc], %[r]\n v_exp_f32_e32 %[ph], %[r]\n"
#define W_FMAC                                                                                          \
    "v_fmac_f32_e32 %[a0], %[w0], %[ph]\n v_fmac_f32_e32 %[a1], %[w1], %[ph]\n v_fmac_f32_e32 %[a2], %[w2], %[ph]\n" \
    "v_fmac_f32_e32 %[a3], %[w3], %[ph]\n v_fmac_f32_e32 %[a4], %[w4], %[ph]\n v_fmac_f32_e32 %[a5], %[w5], %[ph]\n" \
    "v_fmac_f32_e32 %[a6], %[w6], %[ph]\n v_fmac_f32_e32 %[a7], %[w7], %[ph]\n v_fmac_f32_e32 %[a8], %[w8], %[ph]\n" \
    "v_fmac_f32_e32 %[a9], %[w9], %[ph]\n"
#define OPS : [d] "=&v"(d), [e] "=&v"(e), [r] "=&v"(r), [ph] "=&v"(ph), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2),  \
      [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6), [a7] "+v"(a7), [a8] "+v"(a8), [a9] "+v"(a9)        \
    : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x4] "v"(x4), [x5] "v"(x5), [x6] "v"(x6),             \
      [c0] "s"(s.c[0]), [c1] "s"(s.c[1]), [c2] "s"(s.c[2]), [c3] "s"(s.c[3]), [c4] "s"(s.c[4]), [c5] "s"(s.c[5]),   \
      [c6] "s"(s.c[6]), [sc] "s"(s.sc), [w0] "s"(s.w[0]), [w1] "s"(s.w[1]), [w2] "s"(s.w[2]), [w3] "s"(s.w[3]),     \
      [w4] "s"(s.w[4]), [w5] "s"(s.w[5]), [w6] "s"(s.w[6]), [w7] "s"(s.w[7]), [w8] "s"(s.w[8]), [w9] "s"(s.w[9])
    if (V == 0) asm volatile(DIST_FMAC EXPO W_FMAC OPS);
    if (V == 1) asm volatile(DIST_FMA3 EXPO W_FMAC OPS);
    if (V == 2) asm volatile(DIST_DPP EXPO W_FMAC OPS);
    if (V == 3) asm volatile(DIST_FMAC EXPO "v_fmac_f32_e32 %[a0], %[w0], %[ph]\n" OPS);        // no W part
    if (V == 4) asm volatile("v_mov_b32 %[ph], %[x0]\n" W_FMAC OPS);                              // W part only
    if (V == 5) asm volatile(EXPO "v_fmac_f32_e32 %[a0], %[w0], %[ph]\n" OPS);                   // exp only
    if (V == 6) asm volatile("v_mov_b32 %[ph], %[x0]\n"
        "v_fmac_f32_e32 %[a0], %[x1], %[ph]\n v_fmac_f32_e32 %[a1], %[x2], %[ph]\n v_fmac_f32_e32 %[a2], %[x3], %[ph]\n"
        "v_fmac_f32_e32 %[a3], %[x4], %[ph]\n v_fmac_f32_e32 %[a4], %[x5], %[ph]\n v_fmac_f32_e32 %[a5], %[x6], %[ph]\n"
        "v_fmac_f32_e32 %[a6], %[x1], %[ph]\n v_fmac_f32_e32 %[a7], %[x2], %[ph]\n v_fmac_f32_e32 %[a8], %[x3], %[ph]\n"
        "v_fmac_f32_e32 %[a9], %[x4], %[ph]\n" OPS);                                              // W part, VGPR weights
    if (V == 7) asm volatile(DIST_FMAC EXPO
        "v_fmac_f32_e32 %[a0], %[x1], %[ph]\n v_fmac_f32_e32 %[a1], %[x2], %[ph]\n v_fmac_f32_e32 %[a2], %[x3], %[ph]\n"
        "v_fmac_f32_e32 %[a3], %[x4], %[ph]\n v_fmac_f32_e32 %[a4], %[x5], %[ph]\n v_fmac_f32_e32 %[a5], %[x6], %[ph]\n"
        "v_fmac_f32_e32 %[a6], %[x1], %[ph]\n v_fmac_f32_e32 %[a7], %[x2], %[ph]\n v_fmac_f32_e32 %[a8], %[x3], %[ph]\n"
        "v_fmac_f32_e32 %[a9], %[x4], %[ph]\n" OPS);                                              // full, VGPR weights
    if (V == 8) asm volatile(                                                                      // distance with d*e (no same-reg square)
        "v_subrev_f32_e32 %[d], %[c0], %[x0]\n v_subrev_f32_e32 %[e], %[c1], %[x1]\n v_mul_f32_e32 %[r], %[d], %[e]\n"
        "v_subrev_f32_e32 %[d], %[c2], %[x2]\n v_fmac_f32_e32 %[r], %[e], %[d]\n"
        "v_subrev_f32_e32 %[e], %[c3], %[x3]\n v_fmac_f32_e32 %[r], %[d], %[e]\n"
        "v_subrev_f32_e32 %[d], %[c4], %[x4]\n v_fmac_f32_e32 %[r], %[e], %[d]\n"
        "v_subrev_f32_e32 %[e], %[c5], %[x5]\n v_fmac_f32_e32 %[r], %[d], %[e]\n"
        "v_subrev_f32_e32 %[d], %[c6], %[x6]\n v_fmac_f32_e32 %[r], %[e], %[d]\n v_fmac_f32_e32 %[r], %[d], %[e]\n"
        EXPO "v_fmac_f32_e32 %[a0], %[w0], %[ph]\n" OPS);
    if (V == 9) asm volatile("v_mov_b32 %[ph], %[x0]\n"
        "v_pk_fma_f32 %[p0], %[ph2], %[wp0], %[p0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %[p1], %[ph2], %[wp1], %[p1] op_sel_hi:[0,1,1]\n"
        "v_pk_fma_f32 %[p2], %[ph2], %[wp2], %[p2] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %[p3], %[ph2], %[wp3], %[p3] op_sel_hi:[0,1,1]\n"
        "v_pk_fma_f32 %[p4], %[ph2], %[wp4], %[p4] op_sel_hi:[0,1,1]\n"
        : [ph] "=&v"(ph), [p0] "+v"(*(double*)&a0), [p1] "+v"(*(double*)&a2), [p2] "+v"(*(double*)&a4), [p3] "+v"(*(double*)&a6), [p4] "+v"(*(double*)&a8)
        : [x0] "v"(x0), [ph2] "v"(*(double*)&x2), [wp0] "s"(*(double*)&s.w[0]), [wp1] "s"(*(double*)&s.w[2]), [wp2] "s"(*(double*)&s.w[4]),
          [wp3] "s"(*(double*)&s.w[6]), [wp4] "s"(*(double*)&s.w[8]));
    if (V == 10) asm volatile(                                                                    // interleave W fmacs (SGPR) between distance fmacs (VGPR-only)
        "v_subrev_f32_e32 %[d], %[c0], %[x0]\n v_mul_f32_e32 %[r], %[d], %[d]\n v_fmac_f32_e32 %[a0], %[w0], %[x5]\n"
        "v_subrev_f32_e32 %[e], %[c1], %[x1]\n v_fmac_f32_e32 %[r], %[e], %[e]\n v_fmac_f32_e32 %[a1], %[w1], %[x5]\n"
        "v_subrev_f32_e32 %[d], %[c2], %[x2]\n v_fmac_f32_e32 %[r], %[d], %[d]\n v_fmac_f32_e32 %[a2], %[w2], %[x5]\n"
        "v_subrev_f32_e32 %[e], %[c3], %[x3]\n v_fmac_f32_e32 %[r], %[e], %[e]\n v_fmac_f32_e32 %[a3], %[w3], %[x5]\n"
        "v_subrev_f32_e32 %[d], %[c4], %[x4]\n v_fmac_f32_e32 %[r], %[d], %[d]\n v_fmac_f32_e32 %[a4], %[w4], %[x5]\n"
        "v_subrev_f32_e32 %[e], %[c5], %[x5]\n v_fmac_f32_e32 %[r], %[e], %[e]\n v_fmac_f32_e32 %[a5], %[w5], %[x5]\n"
        "v_subrev_f32_e32 %[d], %[c6], %[x6]\n v_fmac_f32_e32 %[r], %[d], %[d]\n v_fmac_f32_e32 %[a6], %[w6], %[x5]\n"
        EXPO "v_fmac_f32_e32 %[a7], %[w7], %[ph]\n v_fmac_f32_e32 %[a8], %[w8], %[ph]\n v_fmac_f32_e32 %[a9], %[w9], %[ph]\n" OPS);
    x0 += 1e-7f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9;
}

template <int V>
void run(const char* name, float* out, int ninstr) {
  SArgs s; for (int i = 0; i < 7; ++i) s.c[i] = 0.1f * i; s.sc = -0.01f; for (int i = 0; i < 10; ++i) s.w[i] = 0.01f * (i + 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<512, 1024>>>(out, s); hipDeviceSynchronize();
  hipEventRecord(e0); k<V><<<512, 1024>>>(out, s); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * 8);   // 8 waves per SIMD, each ITERS pair-waves
  printf("%-44s %7.3f ms  %6.1f cyc@2.4/pair-wave  (%d VALU instr -> %.2f cyc/instr)\n", name, ms, cyc, ninstr, cyc / ninstr);
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  run<0>("full: sub+fmac(VOP2) exp 10 fmac(sgpr)", out, 26);
  run<1>("full: sub+fma(VOP3) exp 10 fmac(sgpr)", out, 26);
  run<2>("full: sub+fmac_dpp exp 10 fmac(sgpr)", out, 26);
  run<3>("distance + exp only", out, 17);
  run<4>("W part only (10 fmac sgpr)", out, 11);
  run<5>("mul+exp only", out, 3);
  run<6>("W part only, VGPR weights", out, 11);
  run<7>("full, VGPR weights", out, 26);
  run<8>("distance d*e (distinct regs) + exp", out, 18);
  run<9>("W part only, 5 pk_fma sgpr-pair", out, 6);
  run<10>("full, W fmacs interleaved in distance", out, 26);
  return 0;
}
