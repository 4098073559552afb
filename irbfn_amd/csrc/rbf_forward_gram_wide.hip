// K1g for wide outputs (16 < O <= 128, d = 7 or 8): rbf_fwd_f16gram_wide -- see rbf_forward_gram_wide.h (the body) and
// rbf_forward_gram.hip (the expansion, its accuracy argument, the pack).  Same mathematics as K1 / K1h
// (src/irbfn_mpc/model.py:169-198; RBF stage flax_rbf.py:258-285).
#include <stdio.h>

#include "rbf_forward_gram_wide.h"

namespace irbfn {

template <int DC, int BC, int NT>
__global__ __launch_bounds__(512) void rbf_fwd_f16gram_wide(const GramArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  wide_gram_body<DC, BC, NT, -1>(ga, F16Roll{}, lds);
}

template <int DC, int NT>
static int launch_gw_bc(const GramArgs& a, int bc, int grid, int block, size_t lds, hipStream_t s) {
#define IRBFN_GWCASE(BCV)                                                                                     \
  case BCV: {                                                                                                 \
    auto k = rbf_fwd_f16gram_wide<DC, BCV, NT>;                                                                \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a);                                                \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_GWCASE(BC_GAUSS)
    IRBFN_GWCASE(BC_IQ)
    IRBFN_GWCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_GWCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// block geometry of the wide K1g kernels: SW centre slices x QG query groups of 32, the slices' rings (three chunk images each)
// within the 160 KB of LDS
void gram_wide_geometry(const irbfn_net* net, int64_t B, int* SW_out, int* QG_out) {
  const int NT = (net->O + 15) / 16;
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  const long groups = (B + 31) / 32;
  int SW = 1;
  while (SW < 4 && (groups * SW + 7) / 8 < 256) SW *= 2;
  if (net->opt[IRBFN_OPT_FWD_F16_S] > 0) SW = net->opt[IRBFN_OPT_FWD_F16_S];
  if (SW != 1 && SW != 2 && SW != 4) SW = 1;
  while (SW > 1 && ((size_t)SW * 3 * gram_chunk_bytes(NT) > 160 * 1024 || nchunks / SW < 2)) SW /= 2;
  // query groups per block, measured at config 4 (us; B = 8192 / 32768 / 262144): SW=2 QG=2: 88 / 173 / 1359; SW=2 QG=4: 109 / 117 / 863;
  // SW=1 QG=4: 133 / 136 / 698; SW=1 QG=8: 183 / 186 / 742 (K1h's wide kernel: 102 / 132 / 921)
  int QG = groups <= 384 ? 2 : (SW == 1 ? 4 : 8 / SW);
  if (net->opt[IRBFN_OPT_FWD_F16_QG] > 0) QG = net->opt[IRBFN_OPT_FWD_F16_QG];
  if (QG < 1 || SW * QG > 8) QG = 8 / SW;
  *SW_out = SW; *QG_out = QG;
}

size_t gram_wide_lds_bytes(const irbfn_net* net, int SW, int QG, size_t extra_red_floats) {
  const int NT = (net->O + 15) / 16;
  const size_t ring = (size_t)SW * 3 * gram_chunk_bytes(NT);
  const size_t red = ((size_t)SW * QG * 2 * 4 * 64 + (size_t)QG * 32 + extra_red_floats) * sizeof(float);
  return ring > red ? ring : red;
}

void gram_fill_args(const irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, GramArgs* a) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  a->f.x = x; a->f.img = net->f16_img; a->f.oscale = net->f16_oscale; a->f.bias = net->bias; a->f.out = out; a->f.gate = net->gate();
  a->f.B = (long)B; a->f.Dreal = net->D; a->f.O = net->O; a->f.nchunks = nchunks; a->f.S = S; a->f.QG = QG;
  a->gimg = net->gram_img;
  a->hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
}

int launch_forward_gram_wide(irbfn_net* net, const float* x, float* out, int64_t B, int SW, int QG, hipStream_t s) {
  if (!net->gram_img || !net->f16_img || (net->DC != 7 && net->DC != 8) || net->O <= 16 || net->O > 128) return IRBFN_ERR_UNSUPPORTED;
  const int NT = (net->O + 15) / 16;
  gram_wide_geometry(net, B, &SW, &QG);
  GramArgs a;
  gram_fill_args(net, x, out, B, SW, QG, &a);
  const size_t lds = gram_wide_lds_bytes(net, SW, QG, 0);
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  int rc;
#define IRBFN_GW_NT(DCV)                                                                      \
  switch (NT) {                                                                               \
    case 2: rc = launch_gw_bc<DCV, 2>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 3: rc = launch_gw_bc<DCV, 3>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 4: rc = launch_gw_bc<DCV, 4>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 5: rc = launch_gw_bc<DCV, 5>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 6: rc = launch_gw_bc<DCV, 6>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 7: rc = launch_gw_bc<DCV, 7>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    case 8: rc = launch_gw_bc<DCV, 8>(a, net->bclass, grid, SW * QG * 64, lds, s); break;      \
    default: rc = IRBFN_ERR_UNSUPPORTED;                                                      \
  }
  if (net->DC == 7) { IRBFN_GW_NT(7) } else { IRBFN_GW_NT(8) }
#undef IRBFN_GW_NT
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_f16gram_wide<D=%d,BC=%d,NT=%d,SW=%d,QG=%d>", net->DC, net->bclass, NT, SW, QG);
    net->last_grid = grid;
    net->last_block = SW * QG * 64;
  }
  return rc;
}

}  // namespace irbfn
