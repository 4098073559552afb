// Dependent MFMAs issued back to back with NO wait states between them, as hipcc emits them in rbf_vjp_f16gram: the second reads the
// first's result as SrcC and writes (1) the same registers, (2) other registers; first = 16x16x16 or 16x16x32, second = 16x16x32.
// Does the hardware interlock, or does the second read a stale accumulator?  Reference: the builtins (compiler-scheduled).
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/probe_mfma_dependent tools/probe_mfma_dependent.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f2_t __attribute__((ext_vector_type(2)));

#define SETUP                                                                                                                   \
  "v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n v_mov_b32 v42, %6\n v_mov_b32 v43, %7\n"                                                \
  "v_mov_b32 v44, %8\n v_mov_b32 v45, %9\n v_mov_b32 v46, %10\n v_mov_b32 v47, %11\n"                                              \
  "v_mov_b32 v48, %12\n v_mov_b32 v49, %13\n v_mov_b32 v50, %14\n v_mov_b32 v51, %15\n"                                            \
  "v_mov_b32 v52, %16\n v_mov_b32 v53, %17\n v_mov_b32 v54, %18\n v_mov_b32 v55, %19\n"                                            \
  "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n s_nop 7\n s_nop 7\n"
#define READ(R) "s_nop 7\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v" #R "0\n v_mov_b32 %1, v" #R "1\n v_mov_b32 %2, v" #R "2\n v_mov_b32 %3, v" #R "3\n"
#define OPS                                                                                                                     \
  : "=v"(o[0]), "=v"(o[1]), "=v"(o[2]), "=v"(o[3])                                                                              \
  : "v"(a4f[0]), "v"(a4f[1]), "v"(b4f[0]), "v"(b4f[1]), "v"(bf[0]), "v"(bf[1]), "v"(bf[2]), "v"(bf[3]), "v"(a2f[0]), "v"(a2f[1]),   \
    "v"(a2f[2]), "v"(a2f[3]), "v"(af[0]), "v"(af[1]), "v"(af[2]), "v"(af[3])                                                      \
  : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v60", "v61", \
    "v62", "v63", "v64", "v65", "v66", "v67", "v70", "v71", "v72", "v73"

__global__ void probe(const _Float16* A, const _Float16* Bm, float* out) {
  const int l = threadIdx.x, g = l >> 4, n = l & 15;
  h8_t a, b, a2;
  h4_t a4, b4;
  for (int j = 0; j < 8; ++j) { a[j] = A[n * 32 + 8 * g + j]; b[j] = Bm[(8 * g + j) * 16 + n]; a2[j] = A[512 + n * 32 + 8 * g + j]; }
  for (int j = 0; j < 4; ++j) { a4[j] = A[1024 + n * 16 + 4 * g + j]; b4[j] = Bm[512 + (4 * g + j) * 16 + n]; }
  const f4_t z = {0, 0, 0, 0};
  const f4_t r16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, z, 0, 0, 0);
  const f4_t refA = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, r16, 0, 0, 0);
  const f4_t r32 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, z, 0, 0, 0);
  const f4_t refC = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, r32, 0, 0, 0);
  const f2_t a4f = __builtin_bit_cast(f2_t, a4), b4f = __builtin_bit_cast(f2_t, b4);
  const f4_t af = __builtin_bit_cast(f4_t, a), bf = __builtin_bit_cast(f4_t, b), a2f = __builtin_bit_cast(f4_t, a2);
  float res[6][4];
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n v_mfma_f32_16x16x32_f16 v[60:63], v[48:51], v[44:47], v[60:63]\n" READ(6) OPS);
    for (int q = 0; q < 4; ++q) res[0][q] = o[q] - refA[q]; }
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n v_mfma_f32_16x16x32_f16 v[70:73], v[48:51], v[44:47], v[60:63]\n" READ(7) OPS);
    for (int q = 0; q < 4; ++q) res[1][q] = o[q] - refA[q]; }
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[60:63], v[52:55], v[44:47], 0\n v_mfma_f32_16x16x32_f16 v[60:63], v[48:51], v[44:47], v[60:63]\n" READ(6) OPS);
    for (int q = 0; q < 4; ++q) res[2][q] = o[q] - refC[q]; }
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[60:63], v[52:55], v[44:47], 0\n v_mfma_f32_16x16x32_f16 v[70:73], v[48:51], v[44:47], v[60:63]\n" READ(7) OPS);
    for (int q = 0; q < 4; ++q) res[3][q] = o[q] - refC[q]; }
  // an unrelated MFMA in between (what K1g's order has)
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[60:63], v[52:55], v[44:47], 0\n v_mfma_f32_16x16x32_f16 v[64:67], v[52:55], v[44:47], v[64:67]\n v_mfma_f32_16x16x32_f16 v[70:73], v[48:51], v[44:47], v[60:63]\n" READ(7) OPS);
    for (int q = 0; q < 4; ++q) res[4][q] = o[q] - refC[q]; }
  // VALU read of an MFMA result with only 2 wait states (the hazard the compiler pads with s_nop 5 / 6)
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x32_f16 v[60:63], v[52:55], v[44:47], 0\n s_nop 1\n v_mov_b32 v70, v60\n v_mov_b32 v71, v61\n v_mov_b32 v72, v62\n v_mov_b32 v73, v63\n" READ(7) OPS);
    for (int q = 0; q < 4; ++q) res[5][q] = o[q] - r32[q]; }
  for (int k = 0; k < 6; ++k)
    for (int q = 0; q < 4; ++q) out[k * 256 + (4 * g + q) * 16 + n] = res[k][q];
  // how many wait states / independent MFMAs between a 16x16x16 and the 16x16x32 that accumulates onto it?
  float w[12][4];
#define GAP(K, FILL)                                                                                                              \
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n" FILL                                   \
                 "v_mfma_f32_16x16x32_f16 v[60:63], v[48:51], v[44:47], v[60:63]\n" READ(6) OPS);                                   \
    for (int q = 0; q < 4; ++q) w[K][q] = o[q] - refA[q]; }
  GAP(0, "s_nop 0\n") GAP(1, "s_nop 1\n") GAP(2, "s_nop 2\n") GAP(3, "s_nop 3\n") GAP(4, "s_nop 4\n") GAP(5, "s_nop 5\n")
  GAP(6, "s_nop 7\n") GAP(7, "s_nop 7\n s_nop 3\n") GAP(8, "s_nop 7\n s_nop 7\n")
  GAP(9, "v_mfma_f32_16x16x32_f16 v[64:67], v[52:55], v[44:47], v[64:67]\n")
  GAP(10, "v_mfma_f32_16x16x32_f16 v[64:67], v[52:55], v[44:47], v[64:67]\n v_mfma_f32_16x16x32_f16 v[70:73], v[52:55], v[44:47], 0\n")
  GAP(11, "v_mfma_f32_16x16x16_f16 v[64:67], v[40:41], v[42:43], v[64:67]\n")
#undef GAP
  for (int k = 0; k < 12; ++k)
    for (int q = 0; q < 4; ++q) out[(6 + k) * 256 + (4 * g + q) * 16 + n] = w[k][q];
  // the other consumers of a 16x16x16 result: a 16x16x16 that accumulates onto it (K2h's chains), a VALU instruction
  const f4_t ref1616 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, r16, 0, 0, 0);
  float y[8][4];
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], v[60:63]\n" READ(6) OPS);
    for (int q = 0; q < 4; ++q) y[0][q] = o[q] - ref1616[q]; }
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n v_mfma_f32_16x16x16_f16 v[70:73], v[40:41], v[42:43], v[60:63]\n" READ(7) OPS);
    for (int q = 0; q < 4; ++q) y[1][q] = o[q] - ref1616[q]; }
#define VR(K, FILL)                                                                                                               \
  { float o[4]; asm volatile(SETUP "v_mfma_f32_16x16x16_f16 v[60:63], v[40:41], v[42:43], 0\n" FILL                                   \
                 "v_mov_b32 v70, v60\n v_mov_b32 v71, v61\n v_mov_b32 v72, v62\n v_mov_b32 v73, v63\n" READ(7) OPS);                 \
    for (int q = 0; q < 4; ++q) y[K][q] = o[q] - r16[q]; }
  VR(2, "") VR(3, "s_nop 1\n") VR(4, "s_nop 3\n") VR(5, "s_nop 5\n") VR(6, "s_nop 7\n") VR(7, "s_nop 7\n s_nop 3\n")
#undef VR
  for (int k = 0; k < 8; ++k)
    for (int q = 0; q < 4; ++q) out[(18 + k) * 256 + (4 * g + q) * 16 + n] = y[k][q];
}

int main() {
  static _Float16 hA[1024 + 256], hB[512 + 256]; static float hO[26 * 256];
  srand(2);
  for (auto& v : hA) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  for (auto& v : hB) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  _Float16 *dA, *dB; float* dO;
  (void)hipMalloc(&dA, sizeof(hA)); (void)hipMalloc(&dB, sizeof(hB)); (void)hipMalloc(&dO, sizeof(hO));
  (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  const char* names[26] = {"16x16x16 -> 16x16x32, same dst (in-place chain)", "16x16x16 -> 16x16x32, other dst", "16x16x32 -> 16x16x32, same dst",
                          "16x16x32 -> 16x16x32, other dst", "16x16x32 -> (unrelated MFMA) -> 16x16x32, other dst", "VALU reads an MFMA result after 2 wait states",
                          "16x16x16 -> s_nop 0 -> 16x16x32 in place", "16x16x16 -> s_nop 1 -> ...", "16x16x16 -> s_nop 2 -> ...", "16x16x16 -> s_nop 3 -> ...",
                          "16x16x16 -> s_nop 4 -> ...", "16x16x16 -> s_nop 5 -> ...", "16x16x16 -> s_nop 7 -> ...", "16x16x16 -> s_nop 7, s_nop 3 -> ...",
                          "16x16x16 -> s_nop 7, s_nop 7 -> ...", "16x16x16 -> one unrelated 16x16x32 -> ...", "16x16x16 -> two unrelated 16x16x32 -> ...",
                          "16x16x16 -> one unrelated 16x16x16 -> ...", "16x16x16 -> 16x16x16 in place, back to back", "16x16x16 -> 16x16x16 other dst, back to back",
                          "16x16x16 -> VALU read, 0 wait states", "16x16x16 -> VALU read, s_nop 1", "16x16x16 -> VALU read, s_nop 3", "16x16x16 -> VALU read, s_nop 5",
                          "16x16x16 -> VALU read, s_nop 7", "16x16x16 -> VALU read, s_nop 7 + 3"};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    (void)hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
    for (int k = 0; k < 26; ++k) {
      float mx = 0; int bad = 0;
      for (int i = 0; i < 256; ++i) { const float d = fabsf(hO[k * 256 + i]); if (!(d <= mx)) mx = d; if (!(d == 0.0f)) ++bad; }
      printf("run %d  %-58s max |diff| %.3g, %d of 256 differ\n", rep, names[k], mx, bad);
    }
  }
  return 0;
}
