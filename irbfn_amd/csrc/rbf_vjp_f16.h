// K2h (rbf_vjp_f16.hip): interface towards the VJP launcher in rbf_vjp.hip.
#pragma once
#include "common.h"

namespace irbfn {

constexpr int vjph_rfq(int DC) { return DC <= 3 ? 4 : (DC <= 7 ? 8 : 12); }   // floats per packed query row (x, gamma)

// d phi / d(d2) from phi for the fast basis classes
template <int BC>
__device__ __forceinline__ float dphi_dd2_h(float phi, float gscale) {
  if constexpr (BC == BC_GAUSS) return -gscale * phi;
  else if constexpr (BC == BC_IQ) return -(phi * phi);
  else return -0.5f * phi * phi * phi;
}

bool vjph_eligible(const irbfn_net* net);
size_t vjph_block_bytes(const irbfn_net* net);
// run_if: null, or a device word -- the kernels return at once unless it holds run_gen, the generation number K2g's pre-pass
// stores there when a query of this call lies outside the box (K2h as the fallback behind K2g; no reset between calls)
int launch_vjp_f16(irbfn_net* net, const float* x, const float* gout, int64_t B, unsigned char* qblk, const float* bmax,
                   int nbmax, float* scales, float* part, int QSB, int Npad, int CT, hipStream_t s, const int* run_if = nullptr, int run_gen = 0);

// K2g (rbf_vjp_gram.hip): u and the centre gradients on the matrix cores as well
bool vjpg_eligible(const irbfn_net* net);
size_t vjpg_block_bytes();
int launch_vjp_gram(irbfn_net* net, const float* x, const float* gout, int64_t B, unsigned char* qblk, const float* bmax, int nbmax,
                    float* scales, int* flag, int gen, float* part, int QSB, int Npad, hipStream_t s);

}  // namespace irbfn
