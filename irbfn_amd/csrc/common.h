// Shared declarations of libirbfn_hip.so (gfx950 only; see include/irbfn_hip.h for the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "irbfn_hip.h"

namespace irbfn {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMaxSplit = 8;       // nsplit <= 8 (gate dims)
constexpr int kMaxD = 8;
// blocks (= partial sums) of every two-stage loss / norm reduction: the size of the caller's `partials` buffer
// (irbfn_train_loss_partials) -- train_step.hip and the cross-entropy of the cluster gate (rbf_vjp.hip) share it
constexpr int kRedBlocks = 256;
// the loss / seed kernels (one roll-out + adjoint per row) run one row per thread up to this many blocks of 256: the
// caller's `partials` buffer holds kSeedBlocksMax floats
constexpr int kSeedBlocksMax = 1024;

// basis classes the hot loops are specialised on
enum BasisClass : int { BC_GAUSS = 0, BC_IQ = 1, BC_IMQ = 2, BC_GENERIC = 3 };

inline int basis_class(int basis) {
  switch (basis) {
    case IRBFN_GAUSSIAN:
    case IRBFN_GAUSSIAN_WIDE:
    case IRBFN_GAUSSIAN_WIDER: return BC_GAUSS;
    case IRBFN_INVERSE_QUADRATIC: return BC_IQ;
    case IRBFN_INVERSE_MULTIQUADRIC: return BC_IMQ;
    default: return BC_GENERIC;
  }
}

// Gaussian family exponent scale a in exp(-a d^2) (flax_rbf.py:35-47)
inline float gauss_scale(int basis) {
  return basis == IRBFN_GAUSSIAN_WIDE ? 0.1f : (basis == IRBFN_GAUSSIAN_WIDER ? 0.01f : 1.0f);
}

extern thread_local int g_last_hip_error;

#define IRBFN_HIP_CHECK(expr)                         \
  do {                                                \
    hipError_t e__ = (expr);                          \
    if (e__ != hipSuccess) {                          \
      ::irbfn::g_last_hip_error = (int)e__;           \
      return IRBFN_ERR_HIP;                           \
    }                                                 \
  } while (0)

struct GateTables {        // device pointers owned by the descriptor
  const float* lo;         // [nsplit][max_ranges]
  const float* hi;         // [nsplit][max_ranges]
  const float* delta;      // [nsplit]
  const int* dim_ranges;   // [n_ranges][nsplit]
  int nsplit, max_ranges, n_ranges;
};

__host__ __device__ constexpr int mfma_cw(int D) { return (D + 1 + 3) & ~3; }   // K1m record: c[D], scale, pad

struct DynParams {         // dynamics.py:24-36
  float p[13];
};

}  // namespace irbfn

// The descriptor behind the opaque handle of the ABI.
struct irbfn_net {
  int D, R, K, O, basis, bclass;
  int DC;       // D padded to a compiled width (padding coordinates are 0 in x and c)
  int N;        // R*K centres
  int OP;       // O padded to a compiled accumulator width
  int S;        // floats per packed centre record: c[D], scale, W[OP], padded to a multiple of 4
  float* rec;   // [N][S]   packed records (device)
  float* bias;  // [OP]     (device, zero padded)
  float* sig2;  // [N]      exp(-2 log_sig) (device) -- VJP
  float* recm;  // [Npad][CW + 16*NT] records of the MFMA forward (K1m); NULL if not eligible
  int Npad;     // N rounded up to a multiple of 16 (MFMA chunk)
  unsigned char* f16_img;   // chunk images of the f16-split matrix-core forward (K1h); NULL if not eligible
  float* f16_oscale;        // [128] per-output power-of-two scale of the K1h weight split
  unsigned char* gram_img;  // chunk images of K1g (distances as a Gram expansion on the matrix cores); NULL if not eligible
  void* gram_hdr;           // K1g: origin and exponents of the expansion (device, written by the pack)
  int gram_ok;              // K1g: the parameters fit the expansion's exactness budget (read back by set_params)
  int gram_exp[5];          // K1g: exponents ex, ec, eq, ea, e2 (diagnostics)
  int gram_checked;         // K1g: gram_ok has been read back at least once (IRBFN_OPT_GRAM_STICKY)
  float* pack_part;         // K1g: partial statistics of the pack, one row per 1024 centres (pack_all.hip)
  int* vjp_flags;           // K2g: ring of 64 hand-over words (rbf_vjp.hip); zero at creation, never reset
  int vjp_gen;              // K2g: generation number of the last VJP call
  float* small_part;            // K1s workspace part[NB][B][OP] (small-batch latency kernel)
  unsigned int* small_ticket;   // K1s arrival counters [64], zero between launches
  // raw parameter pointers are NOT kept: set_params copies what it needs
  float* gate_lo;
  float* gate_hi;
  float* gate_delta;
  int* gate_ranges;
  int nsplit, max_ranges, n_ranges;
  // region-sparse path (rbf_sparse.hip): tables of the card, built by irbfn_net_create where the net is eligible
  float* sp_img;            // the net as the region-sparse kernels hold it in LDS (rbf_sparse.hip, sp_img_layout): centre
                            // table [n_ranges][RS], region index words, region masks, factor entries, Dense weight rows
  int sp_ok, sp_E, sp_cap, sp_RS, sp_EW, sp_OPS;
  float sp_mean_active;     // expected number of regions with gamma != 0 for a query uniform over the card's bounds
  bool has_params;
  int opt[IRBFN_OPT_COUNT];     // irbfn_net_set_option
  char last_name[96];
  int last_grid, last_block;

  irbfn::GateTables gate() const {
    return irbfn::GateTables{gate_lo, gate_hi, gate_delta, gate_ranges, nsplit, max_ranges, n_ranges};
  }
};

namespace irbfn {
// launchers implemented per translation unit
// K0 (pack_all.hip): every image of the net in two launches
int launch_pack_all(irbfn_net* net, const float* centers, const float* log_sigs, const float* kernel, const float* bias,
                    hipStream_t s);
size_t pack_partials_bytes(const irbfn_net* net);
int launch_forward(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s);
int launch_forward_gamma(irbfn_net* net, const float* x, const float* gamma, float* out, int64_t B, hipStream_t s);
int launch_cluster_gate(const float* x, const float* wc, const float* bc, float* logits, float* gamma, int64_t B, int D,
                        int R, hipStream_t s);
bool mfma_eligible(const irbfn_net* net);
size_t mfma_record_floats(int D, int O);
int launch_forward_mfma(irbfn_net* net, const float* x, float* out, int64_t B, int QJ, int nw, hipStream_t s);
bool f16_eligible(const irbfn_net* net);
size_t f16_image_bytes(const irbfn_net* net);
int launch_forward_f16(irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, int terms, hipStream_t s);
bool gram_eligible(const irbfn_net* net);
bool gram_preferred(const irbfn_net* net, int64_t B);
bool gram_wide_preferred(const irbfn_net* net, int64_t B);
void gram_geometry(const irbfn_net* net, int64_t B, int* S, int* QG);
size_t gram_image_bytes(const irbfn_net* net);
size_t gram_header_bytes();
int launch_forward_gram(irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, hipStream_t s);
int launch_tick_gram_narrow(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0, const DynParams& dp,
                            float* controls, float* states, int64_t B, int T, hipStream_t s);
bool tick_through_controls(const irbfn_net* net, int64_t B);
bool f16_narrow_geometry(const irbfn_net* net, int64_t B, int* S, int* QG);
bool tick_f16_narrow_available(const irbfn_net* net, int mode, int64_t B, int T);
int launch_tick_f16_narrow(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                           const DynParams& dp, float* controls, float* states, int64_t B, int T, hipStream_t s);
bool f16_wide_geometry(const irbfn_net* net, int64_t B, int* SW, int* QG);
void f16_wide_normalize(const irbfn_net* net, int* SW, int* QG, bool* pipe);
bool tick_f16_wide_available(const irbfn_net* net, int mode, int64_t B, int T);
int launch_tick_f16_wide(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                         const DynParams& dp, float* controls, float* states, int64_t B, int T, hipStream_t s);
size_t small_workspace_floats(int OP);
int small_ticket_count();
bool small_eligible(const irbfn_net* net, int64_t B);
int launch_forward_small(irbfn_net* net, const float* x, float* out, int64_t B, hipStream_t s);
int launch_gate(irbfn_net* net, const float* x, float* gamma, int64_t B, hipStream_t s);
int64_t vjp_workspace_bytes(const irbfn_net* net, int64_t B);
int launch_vjp(irbfn_net* net, const float* x, const float* gout, float* g_centers, float* g_log_sigs,
               float* g_kernel, float* g_bias, int64_t B, void* ws, int64_t ws_bytes, hipStream_t s,
               const float* gamma_ext = nullptr);
int launch_dgamma(irbfn_net* net, const float* x, const float* gout, float* dgamma, int64_t B, hipStream_t s);
int64_t cluster_gate_vjp_workspace_bytes(int D, int R);
int launch_cluster_gate_vjp(const float* x, const float* gamma, const float* dgamma, const float* glogits, float* dlogits,
                            float* g_wc, float* g_bc, int64_t B, int D, int R, float* ws, hipStream_t s);
int launch_softmax_xent(const float* logits, const float* labels, float* glogits, float* loss, float* partials, int accumulate,
                        int64_t B, int R, hipStream_t s);
int launch_rollout_forward(int mode, const float* x0u, const DynParams& dp, float* states, int64_t B,
                           int T, hipStream_t s);
int launch_rollout_forward_split(int mode, const float* state0, const float* controls, const DynParams& dp,
                                 float* states, int64_t B, int T, hipStream_t s);
bool prefer_mfma(const irbfn_net* net);
int launch_rollout_vjp(int mode, const float* x0u, const DynParams& dp, const float* gstates,
                       float* g_x0u, int64_t B, int T, float clip_tie, hipStream_t s);
int launch_unmirror(float* controls, const int* mirror, int64_t B, int O, int sv0, hipStream_t s);
int launch_forward_rollout(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0,
                           const DynParams& dp, float* controls, float* states, int64_t B, int T,
                           hipStream_t s);
int padded_O(int O);
// region-sparse evaluation of multi-region nets (rbf_sparse.hip)
int sparse_setup(irbfn_net* net, const float* lo, const float* hi, const float* delta, const int* dim_ranges);
void sparse_free(irbfn_net* net);
bool sparse_preferred(const irbfn_net* net, int64_t B);
void sparse_pack_tables(const irbfn_net* net, float** ctab, float** wtab, int* wp);   // K1r / K2r tables packed by pack_all.hip (role P)
bool sparse_vjp_eligible(const irbfn_net* net);
int sparse_vjp_slices(const irbfn_net* net, int64_t B);
size_t sparse_vjp_workspace_bytes(const irbfn_net* net, int64_t B);
int launch_vjp_sparse(irbfn_net* net, const float* x, const float* gout, int64_t B, void* spws, float* part, int SL, int Npad,
                      hipStream_t s);
int launch_forward_sparse(irbfn_net* net, const float* x, float* out, int64_t B, const int* mirror, int sv0, int mode,
                          const float* state0, const DynParams* dp, float* states, int T, hipStream_t s);
int rollout_state_dim(int mode);
int rollout_input_dim(int mode, int T);
}  // namespace irbfn
