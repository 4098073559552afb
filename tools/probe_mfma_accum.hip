// How does v_mfma_f32_16x16x32_f16 accumulate its 32 products?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/probe_mfma_accum.hip)
// Row 0 of A carries +big at slot i, -big at slot j and `small` at slot k, B is all ones (exact products): the exact sum is
// `small`.  An adder tree that keeps W bits below the largest product returns small rounded to 2^-W big.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* A, const _Float16* Bm, const float* Cin, float* out) {
  const int l = threadIdx.x, g = l >> 4, n = l & 15;
  h8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[n * 32 + 8 * g + j]; b[j] = Bm[(8 * g + j) * 16 + n]; }
  f4_t c;
  for (int r = 0; r < 4; ++r) c[r] = Cin[(4 * g + r) * 16 + n];
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + n] = c[r];
}

int main() {
  _Float16 hA[16 * 32], hB[32 * 16]; float hC[256], hO[256];
  _Float16 *dA, *dB; float *dC, *dO;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dO, sizeof(hO));
  auto run = [&](int i, int j, int k, float big, float smallA, float smallB, float cin) {
    for (auto& v : hA) v = 0; for (auto& v : hB) v = 0; for (auto& v : hC) v = 0;
    for (int s = 0; s < 32; ++s) hB[s * 16 + 0] = (_Float16)1.0f;
    hA[i] = (_Float16)big; hA[j] = (_Float16)(-big); hA[k] = (_Float16)smallA; hB[k * 16 + 0] = (_Float16)smallB;
    hB[i * 16] = (_Float16)big; hB[j * 16] = (_Float16)big;
    hC[0] = cin;
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
    return hO[0];
  };
  const float sA = 1.0f + ldexpf(1.0f, -10), sB = 1.0f + ldexpf(1.0f, -10);       // product 1 + 2^-9 + 2^-20
  const double exact = (double)sA * sB;
  printf("exact small product %.10f\n", exact);
  const int pos[][3] = {{0, 1, 2}, {0, 2, 1}, {0, 4, 8}, {0, 8, 16}, {0, 16, 31}, {1, 30, 15}, {0, 31, 16}, {2, 3, 0}};
  for (auto& p : pos) {
    printf("slots +big %2d  -big %2d  small %2d :", p[0], p[1], p[2]);
    for (int e = 0; e <= 15; e += 1) {
      const float r = run(p[0], p[1], p[2], ldexpf(1.0f, e), sA, sB, 0.0f);         // products +-2^(2e)
      printf(" %d:%.3g", 2 * e, (r - exact));
    }
    printf("\n");
  }
  // C input: big in C, -big as a product, small as a product
  printf("C = +big, product -big, small product (slots 0, 5):");
  for (int e = 0; e <= 15; ++e) {
    for (auto& v : hA) v = 0; for (auto& v : hB) v = 0; for (auto& v : hC) v = 0;
    hA[0] = (_Float16)(-ldexpf(1.0f, e)); hB[0] = (_Float16)ldexpf(1.0f, e); hA[5] = (_Float16)sA; hB[5 * 16] = (_Float16)sB;
    hC[0] = ldexpf(1.0f, 2 * e);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
    printf(" %d:%.3g", 2 * e, hO[0] - exact);
  }
  printf("\n");
  // rounding of the final sum: 32 products of (1 + 2^-10)^2 each -> exact 32 (1 + 2^-9 + 2^-20) = 32 + 2^-4 + 2^-15
  {
    for (auto& v : hA) v = 0; for (auto& v : hB) v = 0; for (auto& v : hC) v = 0;
    for (int s = 0; s < 32; ++s) { hA[s] = (_Float16)sA; hB[s * 16] = (_Float16)sB; }
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
    hipMemcpy(hO, dO, sizeof(hO), hipMemcpyDeviceToHost);
    printf("32 equal products: got %.10f exact %.10f\n", hO[0], 32.0 * exact);
  }
  // subnormal f16 inputs honoured?
  {
    const float r = run(0, 1, 2, 1.0f, ldexpf(1.0f, -20), 1.0f, 0.0f);
    printf("subnormal f16 input 2^-20 x 1: got %.6g (2^-20 = %.6g)\n", r, ldexp(1.0, -20));
  }
  return 0;
}
