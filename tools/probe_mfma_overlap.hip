// Does v_mfma_f32_16x16x32_f16 / 16x16x16 tolerate a destination that overlaps a source operand?  Patterns the register allocator
// produced in rbf_vjp_f16gram: dst == SrcA with SrcC = 0, dst == SrcA with SrcC elsewhere, dst == SrcB, each followed at once by a
// second MFMA that uses the first one's sources again (a pipelined reader).  Random operands; reference = the builtin with separate
// registers.   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/probe_mfma_overlap tools/probe_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* A, const _Float16* Bm, const float* Cin, float* out) {
  const int l = threadIdx.x, g = l >> 4, n = l & 15;
  h8_t a, b, a2, b2;
  for (int j = 0; j < 8; ++j) { a[j] = A[n * 32 + 8 * g + j]; b[j] = Bm[(8 * g + j) * 16 + n]; a2[j] = A[512 + n * 32 + 8 * g + j]; b2[j] = Bm[512 + (8 * g + j) * 16 + n]; }
  f4_t c;
  for (int r = 0; r < 4; ++r) c[r] = Cin[(4 * g + r) * 16 + n];
  const f4_t ref0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, f4_t{0, 0, 0, 0}, 0, 0, 0);
  const f4_t refc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  const f4_t ref2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b, c, 0, 0, 0);     // a second product sharing b
  f4_t r[6];
  {  // 0: dst == SrcA, SrcC = 0
    f4_t t = __builtin_bit_cast(f4_t, a);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %1, 0\n s_nop 7\n s_nop 7" : "+v"(t) : "v"(b));
    r[0] = t;
  }
  {  // 1: dst == SrcA, SrcC in other registers
    f4_t t = __builtin_bit_cast(f4_t, a);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %1, %2\n s_nop 7\n s_nop 7" : "+v"(t) : "v"(b), "v"(c));
    r[1] = t;
  }
  {  // 2: dst == SrcB, SrcC in other registers
    f4_t t = __builtin_bit_cast(f4_t, b);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %0, %2\n s_nop 7\n s_nop 7" : "+v"(t) : "v"(a), "v"(c));
    r[2] = t;
  }
  {  // 3: dst == SrcA (C elsewhere), a SECOND MFMA right behind it reads the same B (and its own A): is the second one right?
    f4_t t = __builtin_bit_cast(f4_t, a), t2 = c;
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %2, %3\n v_mfma_f32_16x16x32_f16 %1, %4, %2, %1\n s_nop 7\n s_nop 7"
                 : "+v"(t), "+v"(t2) : "v"(b), "v"(c), "v"(a2));
    r[3] = t; r[4] = t2;
  }
  {  // 5: first MFMA reads A, the next one OVERWRITES A as its destination (dst == SrcA of the previous, still in flight)
    f4_t acc = c, t = __builtin_bit_cast(f4_t, a);
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %1, %1, %3, %4\n s_nop 7\n s_nop 7"
                 : "+v"(acc), "+v"(t) : "v"(b), "v"(b2), "v"(c));
    r[5] = acc;                                              // must equal refc
  }
  const f4_t refs[6] = {ref0, refc, refc, refc, ref2, refc};
  for (int k = 0; k < 6; ++k)
    for (int q = 0; q < 4; ++q) out[k * 256 + (4 * g + q) * 16 + n] = r[k][q] - refs[k][q];
  // k = 16 form (A, B: two registers each): the destination's four registers CONTAIN a source's two (fixed registers: the
  // allocator's sub-register overlaps -- v[62:65] <- v[64:65] -- cannot be written with operand constraints)
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  typedef float f2v __attribute__((ext_vector_type(2)));
  h4_t a4, b4;
  for (int j = 0; j < 4; ++j) { a4[j] = A[n * 32 + 4 * g + j]; b4[j] = Bm[(4 * g + j) * 16 + n]; }
  const f4_t ref16 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c, 0, 0, 0);
  const f2v af = __builtin_bit_cast(f2v, a4), bf = __builtin_bit_cast(f2v, b4);
  float o[4][4];
#define IRBFN_OV16(K, AREG, BREG)                                                                                                  \
  asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n v_mov_b32 v42, %4\n v_mov_b32 v43, %5\n"                                    \
               "v_mov_b32 v44, %6\n v_mov_b32 v45, %7\n v_mov_b32 v46, %6\n v_mov_b32 v47, %7\n"                                    \
               "v_mov_b32 v48, %8\n v_mov_b32 v49, %9\n v_mov_b32 v50, %10\n v_mov_b32 v51, %11\n s_nop 4\n"                       \
               "v_mfma_f32_16x16x16_f16 v[40:43], " AREG ", " BREG ", v[48:51]\n s_nop 7\n s_nop 7\n"                                 \
               "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43"                                       \
               : "=v"(o[K][0]), "=v"(o[K][1]), "=v"(o[K][2]), "=v"(o[K][3])                                                       \
               : "v"(af[0]), "v"(af[1]), "v"(bf[0]), "v"(bf[1]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3])                          \
               : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
  IRBFN_OV16(0, "v[40:41]", "v[44:45]")                       // A = dst[0:1]
  IRBFN_OV16(1, "v[42:43]", "v[44:45]")                       // A = dst[2:3]
  IRBFN_OV16(2, "v[44:45]", "v[40:41]")                       // B = dst[0:1]  (A, B swapped roles: operands are a4 in v44.., b4 in v40..)
  IRBFN_OV16(3, "v[44:45]", "v[42:43]")                       // B = dst[2:3]
#undef IRBFN_OV16
  // cases 2, 3 multiply (b4 as A) x (a4 as B): reference with the roles swapped
  const f4_t ref16s = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, c, 0, 0, 0);
  for (int k = 0; k < 4; ++k)
    for (int q = 0; q < 4; ++q) out[(6 + k) * 256 + (4 * g + q) * 16 + n] = o[k][q] - (k < 2 ? ref16[q] : ref16s[q]);
}

int main() {
  static _Float16 hA[1024], hB[1024]; static float hC[256], hO[10 * 256];
  srand(1);
  for (auto& v : hA) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  for (auto& v : hB) v = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  for (auto& v : hC) v = (rand() % 2001 - 1000) / 10.0f;
  _Float16 *dA, *dB; float *dC, *dO;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dO, sizeof(hO));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
  hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
  const char* names[10] = {"dst == SrcA, C = 0", "dst == SrcA, C elsewhere", "dst == SrcB, C elsewhere", "dst == SrcA then a 2nd MFMA on the same B: first",
                          "                                             ...: second", "MFMA reads A, next MFMA's dst == that A: first", "16x16x16: SrcA = dst[0:1], C elsewhere", "16x16x16: SrcA = dst[2:3], C elsewhere",
                          "16x16x16: SrcB = dst[0:1], C elsewhere", "16x16x16: SrcB = dst[2:3], C elsewhere"};
  for (int k = 0; k < 10; ++k) {
    float mx = 0; int bad = 0;
    for (int i = 0; i < 256; ++i) { const float d = fabsf(hO[k * 256 + i]); if (!(d <= mx)) mx = d; if (!(d == 0.0f)) ++bad; }
    printf("%-62s max |diff| %.3g, %d of 256 differ\n", names[k], mx, bad);
  }
  return 0;
}
