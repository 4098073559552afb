// extern "C" entry points of libirbfn_hip.so (declared in include/irbfn_hip.h).
#include <new>
#include <string.h>

#include "common.h"

namespace irbfn {
thread_local int g_last_hip_error = 0;
int padded_D(int D);
}  // namespace irbfn

using namespace irbfn;

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" {

int irbfn_abi_version(void) { return IRBFN_ABI_VERSION; }

int irbfn_last_hip_error(void) { return g_last_hip_error; }

int irbfn_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return 0;
  }
  return n;
}

const char* irbfn_strerror(int status) {
  switch (status) {
    case IRBFN_OK: return "ok";
    case IRBFN_ERR_BAD_ARG: return "bad argument (null pointer, negative size or unknown enum)";
    case IRBFN_ERR_UNSUPPORTED: return "shape or mode outside the compiled kernel set";
    case IRBFN_ERR_HIP: return "HIP runtime error (see irbfn_last_hip_error)";
    case IRBFN_ERR_NO_PARAMS: return "irbfn_net_set_params has not been called";
    case IRBFN_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

int irbfn_net_create(irbfn_net** out_net, int D, int R, int K, int O, int basis, int nsplit,
                     int max_ranges, const float* lo_tab_host, const float* hi_tab_host,
                     const float* delta_host, const int* dim_ranges_host, int n_ranges) {
  if (!out_net) return IRBFN_ERR_BAD_ARG;
  *out_net = nullptr;
  if (D < 1 || R < 1 || K < 1 || O < 1 || nsplit < 0 || max_ranges < 0 || n_ranges < 0)
    return IRBFN_ERR_BAD_ARG;
  if (basis < IRBFN_GAUSSIAN || basis > IRBFN_MATERN52) return IRBFN_ERR_BAD_ARG;
  if (nsplit > 0 && (!lo_tab_host || !hi_tab_host || !delta_host || max_ranges < 1)) return IRBFN_ERR_BAD_ARG;
  if (n_ranges > 0 && nsplit > 0 && !dim_ranges_host) return IRBFN_ERR_BAD_ARG;
  if (nsplit > D || nsplit > kMaxSplit) return IRBFN_ERR_UNSUPPORTED;
  if ((long)R * K > (1L << 30)) return IRBFN_ERR_UNSUPPORTED;
  const int DC = padded_D(D), OP = padded_O(O);
  if (DC < 0 || OP < 0) return IRBFN_ERR_UNSUPPORTED;
  for (int i = 0; i < n_ranges * nsplit; ++i)
    if (dim_ranges_host[i] < 0 || dim_ranges_host[i] >= max_ranges) return IRBFN_ERR_BAD_ARG;
  if (n_ranges > R) n_ranges = R;   // jnp .at[:, i].set beyond num_regions is dropped (model.py:93)

  irbfn_net* net = new (std::nothrow) irbfn_net();
  if (!net) return IRBFN_ERR_BAD_ARG;
  memset(net, 0, sizeof(*net));
  net->D = D; net->R = R; net->K = K; net->O = O; net->basis = basis;
  net->bclass = basis_class(basis);
  net->DC = DC; net->OP = OP; net->N = R * K;
  net->S = (DC + 1 + OP + 3) & ~3;
  net->nsplit = nsplit; net->max_ranges = max_ranges > 0 ? max_ranges : 1; net->n_ranges = n_ranges;
  net->opt[IRBFN_OPT_FWD_SMALL] = 1;
  net->opt[IRBFN_OPT_FWD_F16_TERMS] = 3;
  net->opt[IRBFN_OPT_FWD_WIDE_PIPE] = 1;
  net->opt[IRBFN_OPT_TICK_FUSED] = 1;

  const size_t tab = (size_t)(nsplit > 0 ? nsplit : 1) * net->max_ranges;
  const size_t nr = (size_t)(n_ranges > 0 ? n_ranges : 1) * (nsplit > 0 ? nsplit : 1);
  hipError_t e = hipSuccess;
  auto alloc = [&](void** p, size_t bytes) {
    if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 4);
  };
  alloc((void**)&net->rec, (size_t)net->N * net->S * sizeof(float));
  alloc((void**)&net->bias, (size_t)OP * sizeof(float));
  alloc((void**)&net->sig2, (size_t)net->N * sizeof(float));
  net->Npad = (net->N + 15) & ~15;
  if (mfma_eligible(net)) alloc((void**)&net->recm, (size_t)net->Npad * mfma_record_floats(D, O) * sizeof(float));
  if (f16_eligible(net)) {
    alloc((void**)&net->f16_img, f16_image_bytes(net));
    alloc((void**)&net->f16_oscale, 128 * sizeof(float));
  }
  if (gram_eligible(net)) {
    alloc((void**)&net->gram_img, gram_image_bytes(net));
    alloc((void**)&net->gram_hdr, gram_header_bytes());
    alloc((void**)&net->pack_part, pack_partials_bytes(net));
    alloc((void**)&net->vjp_flags, 64 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(net->vjp_flags, 0, 64 * sizeof(int));
  }
  alloc((void**)&net->small_part, small_workspace_floats(OP) * sizeof(float));
  alloc((void**)&net->small_ticket, small_ticket_count() * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMemset(net->small_ticket, 0, small_ticket_count() * sizeof(unsigned int));
  alloc((void**)&net->gate_lo, tab * sizeof(float));
  alloc((void**)&net->gate_hi, tab * sizeof(float));
  alloc((void**)&net->gate_delta, (size_t)(nsplit > 0 ? nsplit : 1) * sizeof(float));
  alloc((void**)&net->gate_ranges, nr * sizeof(int));
  if (e == hipSuccess && nsplit > 0) {
    e = hipMemcpy(net->gate_lo, lo_tab_host, tab * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(net->gate_hi, hi_tab_host, tab * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = hipMemcpy(net->gate_delta, delta_host, (size_t)nsplit * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_ranges > 0)
      e = hipMemcpy(net->gate_ranges, dim_ranges_host, (size_t)n_ranges * nsplit * sizeof(int),
                    hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    irbfn_net_destroy(net);
    return IRBFN_ERR_HIP;
  }
  if (nsplit > 0 && n_ranges > 0) {
    const int rcs = sparse_setup(net, lo_tab_host, hi_tab_host, delta_host, dim_ranges_host);
    if (rcs != IRBFN_OK) {
      irbfn_net_destroy(net);
      return rcs;
    }
  }
  *out_net = net;
  return IRBFN_OK;
}

int irbfn_net_destroy(irbfn_net* net) {
  if (!net) return IRBFN_OK;
  void* bufs[] = {net->rec, net->bias, net->sig2, net->recm, net->f16_img, net->f16_oscale, net->gram_img, net->gram_hdr, net->pack_part, net->vjp_flags, net->small_part, net->small_ticket, net->gate_lo, net->gate_hi, net->gate_delta, net->gate_ranges};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  sparse_free(net);
  delete net;
  return IRBFN_OK;
}

int irbfn_net_set_params(irbfn_net* net, const float* centers_dev, const float* log_sigs_dev,
                         const float* kernel_dev, const float* bias_dev, void* stream) {
  if (!net || !centers_dev || !log_sigs_dev || !kernel_dev || !bias_dev) return IRBFN_ERR_BAD_ARG;
  int rc = launch_pack_all(net, centers_dev, log_sigs_dev, kernel_dev, bias_dev, as_stream(stream));
  if (rc == IRBFN_OK) net->has_params = true;
  return rc;
}

int irbfn_net_set_option(irbfn_net* net, int option, int value) {
  if (!net || option < 0 || option >= IRBFN_OPT_COUNT || value < 0) return IRBFN_ERR_BAD_ARG;
  switch (option) {
    case IRBFN_OPT_FWD_KERNEL: if (value > IRBFN_FWD_K1G) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_VJP_KERNEL: if (value > IRBFN_VJP_K2G) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_SMALL: if (value > 1) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_F16_TERMS: if (value != 1 && value != 2 && value != 3) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_Q: if (value > 2) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_NW: if (value > 16) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_QJ: if (value != 0 && value != 1 && value != 2 && value != 4) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_FWD_F16_S:
    case IRBFN_OPT_FWD_F16_QG: if (value > 16) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_VJP_F16_CT: if (value != 0 && value != 2 && value != 4) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_LDS_PAD: if (value > 128 * 1024) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_GRAM_STICKY: if (value > 1) return IRBFN_ERR_BAD_ARG; break;
    case IRBFN_OPT_VJP_QSB: if (value > 4096) return IRBFN_ERR_BAD_ARG; break;
    default: break;
  }
  net->opt[option] = value;
  return IRBFN_OK;
}

int irbfn_net_get_option(const irbfn_net* net, int option, int* value_out) {
  if (!net || !value_out || option < 0 || option >= IRBFN_OPT_COUNT) return IRBFN_ERR_BAD_ARG;
  *value_out = net->opt[option];
  return IRBFN_OK;
}

int irbfn_net_forward(irbfn_net* net, const float* x_dev, float* out_dev, int64_t B, void* stream) {
  if (!net || B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || !out_dev) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  return launch_forward(net, x_dev, out_dev, B, as_stream(stream));
}

int irbfn_net_gate(irbfn_net* net, const float* x_dev, float* gamma_dev, int64_t B, void* stream) {
  if (!net || B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || !gamma_dev) return IRBFN_ERR_BAD_ARG;
  return launch_gate(net, x_dev, gamma_dev, B, as_stream(stream));
}

int64_t irbfn_net_vjp_workspace_bytes(const irbfn_net* net, int64_t B) {
  if (!net || B < 0) return IRBFN_ERR_BAD_ARG;
  return vjp_workspace_bytes(net, B);
}

int irbfn_net_vjp(irbfn_net* net, const float* x_dev, const float* gout_dev, float* g_centers_dev,
                  float* g_log_sigs_dev, float* g_kernel_dev, float* g_bias_dev, int64_t B,
                  void* workspace_dev, int64_t workspace_bytes, void* stream) {
  if (!net || B < 0 || !g_centers_dev || !g_log_sigs_dev || !g_kernel_dev || !g_bias_dev)
    return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!x_dev || !gout_dev)) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  if (workspace_bytes < vjp_workspace_bytes(net, B) || (!workspace_dev && vjp_workspace_bytes(net, B) > 0))
    return IRBFN_ERR_BAD_ARG;
  return launch_vjp(net, x_dev, gout_dev, g_centers_dev, g_log_sigs_dev, g_kernel_dev, g_bias_dev, B,
                    workspace_dev, workspace_bytes, as_stream(stream));
}

int irbfn_net_vjp_gamma(irbfn_net* net, const float* x_dev, const float* gamma_dev, const float* gout_dev,
                        float* g_centers_dev, float* g_log_sigs_dev, float* g_kernel_dev, float* g_bias_dev, float* dgamma_dev,
                        int64_t B, void* workspace_dev, int64_t workspace_bytes, void* stream) {
  if (!net || B < 0 || !g_centers_dev || !g_log_sigs_dev || !g_kernel_dev || !g_bias_dev) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!x_dev || !gout_dev || !gamma_dev)) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  if (workspace_bytes < vjp_workspace_bytes(net, B) || (!workspace_dev && vjp_workspace_bytes(net, B) > 0))
    return IRBFN_ERR_BAD_ARG;
  int rc = launch_vjp(net, x_dev, gout_dev, g_centers_dev, g_log_sigs_dev, g_kernel_dev, g_bias_dev, B, workspace_dev,
                      workspace_bytes, as_stream(stream), B > 0 ? gamma_dev : nullptr);
  if (rc == IRBFN_OK && dgamma_dev) rc = launch_dgamma(net, x_dev, gout_dev, dgamma_dev, B, as_stream(stream));
  return rc;
}

int64_t irbfn_cluster_gate_vjp_workspace_bytes(int D, int R) {
  if (D < 1 || R < 1) return IRBFN_ERR_BAD_ARG;
  return cluster_gate_vjp_workspace_bytes(D, R);
}

int irbfn_cluster_gate_vjp(const float* x_dev, const float* gamma_dev, const float* dgamma_dev, const float* glogits_dev,
                           float* dlogits_dev, float* g_wc_dev, float* g_bc_dev, int64_t B, int D, int R,
                           void* workspace_dev, int64_t workspace_bytes, void* stream) {
  if (B < 0 || D < 1 || R < 1 || !g_wc_dev || !g_bc_dev) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!x_dev || !gamma_dev || !dgamma_dev || !dlogits_dev)) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!workspace_dev || workspace_bytes < cluster_gate_vjp_workspace_bytes(D, R))) return IRBFN_ERR_BAD_ARG;
  return launch_cluster_gate_vjp(x_dev, gamma_dev, dgamma_dev, glogits_dev, dlogits_dev, g_wc_dev, g_bc_dev, B, D, R,
                                 static_cast<float*>(workspace_dev), as_stream(stream));
}

int irbfn_softmax_xent(const float* logits_dev, const float* labels_dev, float* glogits_dev, float* loss_dev,
                       float* partials_dev, int accumulate, int64_t B, int R, void* stream) {
  if (B < 1 || R < 1 || !logits_dev || !labels_dev || !glogits_dev || !loss_dev || !partials_dev) return IRBFN_ERR_BAD_ARG;
  return launch_softmax_xent(logits_dev, labels_dev, glogits_dev, loss_dev, partials_dev, accumulate, B, R, as_stream(stream));
}

int irbfn_rollout_state_dim(int mode) {
  const int s = rollout_state_dim(mode);
  return s < 0 ? IRBFN_ERR_BAD_ARG : s;
}

int irbfn_rollout_input_dim(int mode, int T) {
  if (T < 0) return IRBFN_ERR_BAD_ARG;
  const int l = rollout_input_dim(mode, T);
  return l < 0 ? IRBFN_ERR_BAD_ARG : l;
}

static int load_dyn(int mode, const float* dyn_params_host, DynParams* dp) {
  memset(dp, 0, sizeof(*dp));
  const bool needs = mode == IRBFN_ROLLOUT_ST_SELECT || mode == IRBFN_ROLLOUT_ST_KS ||
                     mode == IRBFN_ROLLOUT_FRENET_LS;
  if (needs && !dyn_params_host) return IRBFN_ERR_BAD_ARG;
  if (dyn_params_host) memcpy(dp->p, dyn_params_host, sizeof(dp->p));
  return IRBFN_OK;
}

int irbfn_rollout_forward(int mode, const float* x0u_dev, const float* dyn_params_host,
                          float* states_dev, int64_t B, int T, void* stream) {
  if (rollout_state_dim(mode) < 0 || B < 0 || T < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0 || T == 0) return IRBFN_OK;
  if (!x0u_dev || !states_dev) return IRBFN_ERR_BAD_ARG;
  DynParams dp;
  int rc = load_dyn(mode, dyn_params_host, &dp);
  if (rc != IRBFN_OK) return rc;
  return launch_rollout_forward(mode, x0u_dev, dp, states_dev, B, T, as_stream(stream));
}

int irbfn_rollout_vjp(int mode, const float* x0u_dev, const float* dyn_params_host,
                      const float* gstates_dev, float* g_x0u_dev, int64_t B, int T, float clip_tie,
                      void* stream) {
  if (rollout_state_dim(mode) < 0 || B < 0 || T < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x0u_dev || !g_x0u_dev || (T > 0 && !gstates_dev)) return IRBFN_ERR_BAD_ARG;
  DynParams dp;
  int rc = load_dyn(mode, dyn_params_host, &dp);
  if (rc != IRBFN_OK) return rc;
  return launch_rollout_vjp(mode, x0u_dev, dp, gstates_dev, g_x0u_dev, B, T, clip_tie, as_stream(stream));
}

int irbfn_net_forward_rollout(irbfn_net* net, int mode, const float* x_dev, const float* state0_dev,
                              const float* dyn_params_host, float* controls_dev, float* states_dev,
                              int64_t B, int T, void* stream) {
  if (!net || rollout_state_dim(mode) < 0 || B < 0 || T < 1) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || !state0_dev || !states_dev) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  DynParams dp;
  int rc = load_dyn(mode, dyn_params_host, &dp);
  if (rc != IRBFN_OK) return rc;
  return launch_forward_rollout(net, mode, x_dev, nullptr, state0_dev, dp, controls_dev, states_dev, B, T,
                                as_stream(stream));
}

int irbfn_net_tick_needs_controls(irbfn_net* net, int mode, int64_t B, int T) {
  if (!net || rollout_state_dim(mode) < 0) return IRBFN_ERR_BAD_ARG;
  if (tick_f16_wide_available(net, mode, B, T) || tick_f16_narrow_available(net, mode, B, T)) return 0;    // one launch, controls stay in LDS
  return tick_through_controls(net, B) ? 1 : 0;              // forward + split-row roll-out through the caller's buffer
}

int irbfn_net_forward_gamma(irbfn_net* net, const float* x_dev, const float* gamma_dev, float* out_dev, int64_t B,
                            void* stream) {
  if (!net || B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || !gamma_dev || !out_dev) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  return launch_forward_gamma(net, x_dev, gamma_dev, out_dev, B, as_stream(stream));
}

int irbfn_cluster_gate(const float* x_dev, const float* wc_dev, const float* bc_dev, float* logits_dev, float* gamma_dev,
                       int64_t B, int D, int R, void* stream) {
  if (B < 0) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || !wc_dev || !bc_dev || !logits_dev || !gamma_dev) return IRBFN_ERR_BAD_ARG;
  return launch_cluster_gate(x_dev, wc_dev, bc_dev, logits_dev, gamma_dev, B, D, R, as_stream(stream));
}

int irbfn_plan_tick(irbfn_net* net, int mode, const float* x_dev, const int32_t* mirror_dev, const float* state0_dev,
                    const float* dyn_params_host, float* controls_dev, float* states_dev, int64_t B, int T,
                    void* stream) {
  if (!net || rollout_state_dim(mode) < 0 || B < 0 || T < 1) return IRBFN_ERR_BAD_ARG;
  if (B == 0) return IRBFN_OK;
  if (!x_dev || (states_dev && !state0_dev) || (!states_dev && !controls_dev)) return IRBFN_ERR_BAD_ARG;
  if (!net->has_params) return IRBFN_ERR_NO_PARAMS;
  if (net->O != 2 * T) return IRBFN_ERR_BAD_ARG;
  DynParams dp;
  memset(&dp, 0, sizeof(dp));
  int rc = states_dev ? load_dyn(mode, dyn_params_host, &dp) : IRBFN_OK;
  if (rc != IRBFN_OK) return rc;
  return launch_forward_rollout(net, mode, x_dev, mirror_dev, state0_dev, dp, controls_dev, states_dev, B, T,
                                as_stream(stream));
}

int irbfn_net_last_launch(const irbfn_net* net, char* name_buf, int name_len, int* grid, int* block) {
  if (!net) return IRBFN_ERR_BAD_ARG;
  if (name_buf && name_len > 0) {
    strncpy(name_buf, net->last_name, (size_t)name_len - 1);
    name_buf[name_len - 1] = 0;
  }
  if (grid) *grid = net->last_grid;
  if (block) *block = net->last_block;
  return IRBFN_OK;
}

}  // extern "C"
