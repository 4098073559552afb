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
}

int main() {
  static _Float16 hA[1024], hB[1024]; static float hC[256], hO[6 * 256];
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
  const char* names[6] = {"dst == SrcA, C = 0", "dst == SrcA, C elsewhere", "dst == SrcB, C elsewhere", "dst == SrcA then a 2nd MFMA on the same B: first",
                          "                                             ...: second", "MFMA reads A, next MFMA's dst == that A: first"};
  for (int k = 0; k < 6; ++k) {
    float mx = 0; int bad = 0;
    for (int i = 0; i < 256; ++i) { const float d = fabsf(hO[k * 256 + i]); if (!(d <= mx)) mx = d; if (!(d == 0.0f)) ++bad; }
    printf("%-62s max |diff| %.3g, %d of 256 differ\n", names[k], mx, bad);
  }
  return 0;
}
