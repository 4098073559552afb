// Store/load pattern microbenchmark for gfx950: aligned vs unaligned 16-byte accesses, contiguous vs
// row-run patterns (the roll-out kernel's output layout).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// MODE 0: contiguous float4 stores, aligned.  1: same but base shifted by 4 bytes (unaligned).
// 2: row-run pattern, 7 lanes x 16 B per 112-B run, row stride 1400 B, misaligned as in the roll-out.
// 3: row-run with dword stores (28 lanes per run).  4: contiguous dword stores.  5: row-run dwordx2 8-B aligned.
template <int MODE>
__global__ void st(float* out, long nfloats, int iters) {
  const long wave = (long)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  float* base = out + wave * 22528;   // one wave-tile = 64 rows x 352 floats (MODE 6/7) or 350
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 1) {
      for (int k = 0; k < 87; ++k) {   // 87 x 1 KB ~ 89 KB
        float* p = base + (MODE == 1 ? 1 : 0) + k * 256 + lane * 4;
        *reinterpret_cast<f4u*>(p) = f4u{1.f, 2.f, 3.f, (float)it};
      }
    } else if (MODE == 2) {
      const int rsub = lane / 7, part = lane % 7;
      if (rsub < 9)
        for (int c = 0; c < 12; ++c)
          for (int j = 0; j < 8; ++j) {
            const int r = j * 9 + rsub;
            if (r < 64) *reinterpret_cast<f4u*>(base + r * 350 + c * 28 + 4 * part) = f4u{1.f, 2.f, 3.f, (float)it};
          }
    } else if (MODE == 3) {
      for (int c = 0; c < 12; ++c)
        for (int k = 0; k < 28; ++k) {
          const int idx = lane + 64 * k;
          const int r = idx / 28, cc = idx % 28;
          base[r * 350 + c * 28 + cc] = (float)it;
        }
    } else if (MODE == 4) {
      for (int k = 0; k < 350; ++k) base[k * 64 + lane] = (float)it;
    } else if (MODE == 6 || MODE == 7) {   // 8 lanes x 16 B aligned per row, 8 rows per instruction, row stride 1408 B (11 lines)
      const int rsub = lane >> 3, part = lane & 7;
      const int run = MODE == 6 ? 28 : 32;          // 112-byte runs (partial lines) vs 128-byte runs (whole lines)
      if (part * 4 < run)
        for (int c = 0; c < 11; ++c)
          for (int j = 0; j < 8; ++j) {
            const int r = j * 8 + rsub;
            *reinterpret_cast<float4*>(base + r * 352 + c * run + 4 * part) = float4{1.f, 2.f, 3.f, (float)it};
          }
    } else if (MODE == 5) {   // 14 lanes x 8 B per run, 4 rows per instruction; rows 8-B aligned (1400 B stride)
      const int rsub = lane / 14, part = lane % 14;
      if (rsub < 4)
        for (int c = 0; c < 12; ++c)
          for (int j = 0; j < 16; ++j) {
            const int r = j * 4 + rsub;
            *reinterpret_cast<float2*>(base + r * 350 + c * 28 + 2 * part) = float2{1.f, (float)it};
          }
    }
  }
}

template <int MODE>
void run(const char* name, float* buf, long nwaves) {
  const int iters = 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  st<MODE><<<dim3((unsigned)(nwaves / 4)), dim3(256)>>>(buf, 0, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  st<MODE><<<dim3((unsigned)(nwaves / 4)), dim3(256)>>>(buf, 0, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = MODE == 6 ? (double)nwaves * 64 * 11 * 28 * 4 * iters : MODE == 7 ? (double)nwaves * 64 * 11 * 32 * 4 * iters
               : (MODE == 2 || MODE == 3 || MODE == 5) ? (double)nwaves * 64 * 12 * 28 * 4 * iters
               : (MODE == 4 ? (double)nwaves * 350 * 64 * 4 * iters : (double)nwaves * 87 * 1024 * iters);
  printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
  const long nwaves = 4096;
  float* buf; hipMalloc(&buf, (nwaves * 22528 + 1024) * sizeof(float));
  run<0>("contiguous float4 aligned", buf, nwaves);
  run<1>("contiguous float4 +4B (unaligned)", buf, nwaves);
  run<4>("contiguous dword", buf, nwaves);
  run<2>("row-run float4 (mixed alignment)", buf, nwaves);
  run<3>("row-run dword", buf, nwaves);
  run<5>("row-run dwordx2 (8-B aligned)", buf, nwaves);
  run<6>("row-run float4 aligned, 112-B runs", buf, nwaves);
  run<7>("row-run float4 aligned, 128-B lines", buf, nwaves);
  return 0;
}
