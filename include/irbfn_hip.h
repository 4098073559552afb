/* irbfn_hip.h -- C ABI of libirbfn_hip.so: the MI355X (gfx950) implementation of the IRBFN hot path.
 *
 * Each entry point names the reference interface (hzheng40/irbfn @ 2024_10_08, path:line) it replaces.
 * Conventions
 *   - every function returns an int status: 0 = IRBFN_OK, negative = irbfn_status below; nothing
 *     throws or aborts across the ABI; irbfn_strerror() turns a status into text.
 *   - `_dev` pointers are device (HBM) pointers owned by the caller; `_host` pointers are host memory
 *     that is copied during the call.  The library never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work of a call is
 *     enqueued on that stream; the call does not synchronise (no hipMalloc/hipFree/sync on the
 *     launch path, so calls can be captured into a hipGraph once a descriptor exists).
 *   - all floating point data is IEEE binary32, row-major, densely packed.
 *   - NaN/Inf inputs propagate as IEEE arithmetic dictates (no clamping), like the reference.
 */
#ifndef IRBFN_HIP_H_
#define IRBFN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRBFN_ABI_VERSION 1

typedef enum irbfn_status {
  IRBFN_OK = 0,
  IRBFN_ERR_BAD_ARG = -1,       /* NULL pointer, negative size, unknown enum */
  IRBFN_ERR_UNSUPPORTED = -2,   /* shape outside the compiled kernel set (see irbfn_net_create) */
  IRBFN_ERR_HIP = -3,           /* a HIP runtime call failed; irbfn_last_hip_error() has the code */
  IRBFN_ERR_NO_PARAMS = -4,     /* forward/vjp called before irbfn_net_set_params */
  IRBFN_ERR_NO_DEVICE = -5      /* no gfx950 device visible */
} irbfn_status;

/* Radial basis functions of flax_rbf (deprecated/f1tenth_gym/examples/flax_rbf/flax_rbf/flax_rbf.py:34-111),
 * selected in the reference by `basis_func: <name>` in the YAML model card (irbfn_planner.py:72). */
typedef enum irbfn_basis {
  IRBFN_GAUSSIAN = 0,             /* exp(-d^2)            flax_rbf.py:35-37 */
  IRBFN_GAUSSIAN_WIDE = 1,        /* exp(-0.1 d^2)        :40-42 */
  IRBFN_GAUSSIAN_WIDER = 2,       /* exp(-0.01 d^2)       :45-47 */
  IRBFN_INVERSE_QUADRATIC = 3,    /* 1/(1+d^2)            :50-52 */
  IRBFN_LINEAR = 4,               /* d                    :55-57 */
  IRBFN_QUADRATIC = 5,            /* d^2                  :61-63 */
  IRBFN_MULTIQUADRIC = 6,         /* sqrt(1+d^2)          :67-69 */
  IRBFN_INVERSE_MULTIQUADRIC = 7, /* 1/sqrt(1+d^2)        :73-75 */
  IRBFN_SPLINE = 8,               /* d^2 log(d+1)         :79-81 */
  IRBFN_POISSON_ONE = 9,          /* (d-1) exp(-d)        :85-87 */
  IRBFN_POISSON_TWO = 10,         /* ((d-2)/2) d exp(-d)  :91-97 */
  IRBFN_MATERN32 = 11,            /* :101-103 */
  IRBFN_MATERN52 = 12             /* :107-111 */
} irbfn_basis;

/* Roll-out models (one lane per trajectory). */
typedef enum irbfn_rollout_mode {
  IRBFN_ROLLOUT_ST_SELECT = 0, /* integrate_st_mult: scan of dynamic_st_onestep, select(V>3, f, f_ks)
                                  src/irbfn_mpc/dynamics.py:9-100.        state 7, x0u[B,7+2T] -> [B,T,7] */
  IRBFN_ROLLOUT_ST_KS = 1,     /* scan of the kinematic one-step map dynamic_st_onestep_aux
                                  src/irbfn_mpc/dynamics.py:103-187 (T=1 is that function itself).
                                                                          state 7, x0u[B,7+2T] -> [B,T,7] */
  IRBFN_ROLLOUT_FULLINT = 2,   /* inline kinematic bicycle of train_step_fullint
                                  scripts/train_nmpc.py:306-374.          state 5, x0u[B,1+2T]=[v0,u] -> [B,T,5] */
  IRBFN_ROLLOUT_FRENET_LS = 3, /* integrate_frenet_mult (low-speed RHS)
                                  src/irbfn_mpc/dynamics.py:190-290.      state 8, x0u[B,8+2T] -> [B,T,8] */
  IRBFN_ROLLOUT_SPIRAL = 4     /* integrate_path_mult: cubic-spiral path, T = number of samples N
                                  src/irbfn_mpc/planner_utils.py:8-77.    state 6, x0u[B,5]=(k0..k3,s) -> [B,N,6] */
} irbfn_rollout_mode;

/* ---------------------------------------------------------------------------------------------
 * Network descriptor.  Replaces the Flax module object `WCRBFNet(**cfg)` (src/irbfn_mpc/model.py:98-167)
 * plus the bound parameter pytree.  The static part is the YAML model card
 * (scripts/train_nmpc.py:431-450): in_features D, out_features O, num_kernels K, num_regions R,
 * basis_func, lower_bounds / upper_bounds / delta / dimension_ranges of the smooth region gate
 * `_region_activation` (model.py:42-95).
 *
 *   lo_tab_host, hi_tab_host : [nsplit][max_ranges] -- row d holds lower_bounds[d][:] / upper_bounds[d][:]
 *                              (ragged rows padded with anything; padding is never indexed)
 *   delta_host               : [nsplit]
 *   dim_ranges_host          : [n_ranges][nsplit] ints; regions r >= n_ranges have gamma = 0 (model.py:70)
 *   nsplit = len(activation_idx) (model.py:128); the gate reads x[:, d] for d < nsplit (model.py:74-81).
 *
 * Compiled kernel set: D in 1..8; any O >= 1 (O is padded internally to a compiled width);
 * R*K >= 1; nsplit <= 8.  Outside it: IRBFN_ERR_UNSUPPORTED.
 * The descriptor owns device buffers for the packed centre records, the gate tables and a small
 * scratch area of the small-batch latency kernel; calls that use the same descriptor must be ordered
 * on one stream (or serialised by the caller) -- use one descriptor per concurrent stream.
 */
typedef struct irbfn_net irbfn_net;

int irbfn_net_create(irbfn_net** out_net, int D, int R, int K, int O, int basis, int nsplit,
                     int max_ranges, const float* lo_tab_host, const float* hi_tab_host,
                     const float* delta_host, const int* dim_ranges_host, int n_ranges);
int irbfn_net_destroy(irbfn_net* net);

/* Binds the parameter pytree {"rbf_list": {"centers"[R,K,D], "log_sigs"[R,K]},
 * "linear": {"kernel"[K,O], "bias"[O]}} (checkpoint layout).  Device pointers; the
 * data is re-packed on `stream` into the descriptor's own images (two launches, pack_all.hip), so the
 * caller may overwrite its arrays afterwards.  Call again after every optimiser step.  For nets the
 * matrix-core kernels K1g / K2g can take (one region, d <= 8, fast basis) the call ends with ONE small
 * synchronous read-back on `stream` (64 bytes: do the bound parameters fit K1g's expansion?) -- the
 * kernel choice of later forwards is a property of the parameters; every other net returns without
 * synchronising. */
int irbfn_net_set_params(irbfn_net* net, const float* centers_dev, const float* log_sigs_dev,
                         const float* kernel_dev, const float* bias_dev, void* stream);

/* Per-descriptor options: kernel selection and launch geometry.  The defaults are what the library ships
 * with; everything else exists for A/B measurements and for the reduced-precision report of BASELINE config 5.
 * Nothing on the launch path reads the process environment.  value 0 of a geometry option = automatic. */
typedef enum irbfn_option {
  IRBFN_OPT_FWD_KERNEL = 0,    /* irbfn_fwd_kernel below; default IRBFN_FWD_AUTO */
  IRBFN_OPT_FWD_SMALL = 1,     /* 1 (default): B <= 64 runs on the latency kernel K1s; 0: never */
  IRBFN_OPT_FWD_F16_TERMS = 2, /* MFMA operands of K1h.  3 (default): (hi, lo) f16 pairs, float32-grade result;
                                  1: plain f16 operands (~1e-4 relative), 2: plain bf16 operands (~4e-3; O <= 16 only)
                                  -- the two reduced-precision variants BASELINE config 5 asks to report */
  IRBFN_OPT_FWD_F16_MINB = 3,  /* smallest batch K1h takes in automatic mode (default 65) */
  IRBFN_OPT_FWD_Q = 4,         /* K1: queries per lane (1, 2) */
  IRBFN_OPT_FWD_NW = 5,        /* K1 / K1m: waves per workgroup */
  IRBFN_OPT_FWD_QJ = 6,        /* K1m: query tiles per wave (1, 2, 4) */
  IRBFN_OPT_FWD_F16_S = 7,     /* K1h: centre slices per query group */
  IRBFN_OPT_FWD_F16_QG = 8,    /* K1h: query groups of 32 per workgroup */
  IRBFN_OPT_VJP_KERNEL = 9,    /* irbfn_vjp_kernel below; default IRBFN_VJP_AUTO */
  IRBFN_OPT_VJP_F16_CT = 10,   /* K2h: 16-centre tiles per wave (2 default, 4) */
  IRBFN_OPT_LDS_PAD = 11,      /* diagnosis: extra dynamic LDS bytes per workgroup of K1h / K2h (lowers occupancy) */
  IRBFN_OPT_FWD_WIDE_PIPE = 12,/* K1h, 16 < O <= 128: 1 (default) pipelined kernel (deferred MFMAs, LDS-DMA ring of three), 0: two-buffer kernel */
  IRBFN_OPT_TICK_FUSED = 13,   /* planning tick on the matrix-core kernels: 1 (default) one launch where the instance exists (wide: d = 7,
                                  O = 2T in (96, 112], single-track modes; narrow: O = 2T <= 16, d = 7 single-track / inline bicycle,
                                  d = 8 Frenet), 0: forward + sign flip + roll-out launches */
  IRBFN_OPT_GRAM_STICKY = 14,  /* K1g / K2g: 0 (default) every irbfn_net_set_params reads the pack's verdict back (one small synchronous copy);
                                  1: only the first one does and later calls keep its verdict -- for training loops, which re-bind every step.
                                  The kernels test the device-side verdict themselves and fall back to the VALU distances / to K2h, so a stale
                                  host verdict costs speed, never correctness */
  IRBFN_OPT_VJP_QSB = 15,      /* K2g / K2h: query slices of the VJP grid (slabs summed in fixed order); 0 (default): automatic.  Never more than
                                  the workspace was sized for (the automatic number of the all-float32 kernel) */
  IRBFN_OPT_COUNT = 16
} irbfn_option;
typedef enum irbfn_fwd_kernel {
  IRBFN_FWD_AUTO = 0, /* B <= 64: K1s; sparse multi-region gate: K1r; one region + fast basis: K1g (d <= 8, parameters inside its budget; O <= 16:
                         B >= 12288; 16 < O <= 128: d = 7 or 8, B >= 2048, >= 256 centres) else K1h (O <= 128); O > 16: K1m; otherwise K1 */
  IRBFN_FWD_K1 = 1,   /* rbf_fwd_qlane: all-float32 VALU kernel (any net) */
  IRBFN_FWD_K1M = 2,  /* rbf_fwd_mfma: Phi x W on the f32-input matrix cores */
  IRBFN_FWD_K1H = 3,  /* rbf_fwd_f16mfma[_wide]: Phi x W on the f16 matrix cores, hi/lo operand pairs */
  IRBFN_FWD_K1R = 4,  /* rbf_fwd_sparse: several regions, every query visits only the regions whose gamma != 0 */
  IRBFN_FWD_K1G = 5   /* rbf_fwd_f16gram[_wide]: one region, d <= 8 (O <= 16) / d = 7 or 8 (16 < O <= 128): the squared distances as an exactly-cancelling Gram expansion
                         on the f16 matrix cores in front of K1h's Phi x W; IRBFN_ERR_UNSUPPORTED when the bound parameters
                         do not fit the expansion (widths of 1e-3 of the centres' spread, non-finite values) */
} irbfn_fwd_kernel;
typedef enum irbfn_vjp_kernel { IRBFN_VJP_AUTO = 0, IRBFN_VJP_K2 = 1, IRBFN_VJP_K2H = 2, IRBFN_VJP_K2R = 3, IRBFN_VJP_K2G = 4 } irbfn_vjp_kernel;
/* K2: all-float32 VALU; K2H: hbar and dW on the f16 matrix cores; K2R: region-sparse pair lists; K2G: the squared distances (K1g's
 * expansion) and the centre gradients on the matrix cores as well -- automatic from 16384 queries where K1g's expansion fits.
 * A forced kernel that cannot take the net answers IRBFN_ERR_UNSUPPORTED at the call that would launch it. */
int irbfn_net_set_option(irbfn_net* net, int option, int value);
int irbfn_net_get_option(const irbfn_net* net, int option, int* value_out);

/* Forward: replaces `WCRBFNet.apply(params, x)` (model.py:169-198; called as
 * `state.apply_fn(state.params, x)` in pred_step, src/irbfn_mpc/irbfn_planner.py:29-32).
 * x_dev[B,D] -> out_dev[B,O].  B = 0 is a no-op. */
int irbfn_net_forward(irbfn_net* net, const float* x_dev, float* out_dev, int64_t B, void* stream);

/* Region gate alone: `_region_activation` (model.py:42-95).  x_dev[B,D] -> gamma_dev[B,R]. */
int irbfn_net_gate(irbfn_net* net, const float* x_dev, float* gamma_dev, int64_t B, void* stream);

/* Bytes of scratch irbfn_net_vjp needs for batch B (caller allocates; may be reused across calls). */
int64_t irbfn_net_vjp_workspace_bytes(const irbfn_net* net, int64_t B);

/* Parameter VJP: replaces `jax.value_and_grad(loss_fn)(state.params)` restricted to the network
 * (scripts/train_nmpc.py:297-298, scripts/train_nmpc_frenet.py:388-389,416-417).
 * Cotangent gout_dev[B,O] -> g_centers[R,K,D], g_log_sigs[R,K], g_kernel[K,O], g_bias[O]
 * (overwritten, not accumulated).  Gradients w.r.t. x are never taken by the reference and are
 * not produced.  Deterministic (no float atomics). */
int irbfn_net_vjp(irbfn_net* net, const float* x_dev, const float* gout_dev, float* g_centers_dev,
                  float* g_log_sigs_dev, float* g_kernel_dev, float* g_bias_dev, int64_t B,
                  void* workspace_dev, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Roll-outs.  Replace integrate_st_mult (dynamics.py:94-100), dynamic_st_onestep_aux (:103-187),
 * integrate_frenet_mult (:284-290), the inline bicycle of train_step_fullint
 * (scripts/train_nmpc.py:329-374) and integrate_path_mult (planner_utils.py:62-77).
 * Control layout is the reference's column-major pairs u = [a_0..a_{T-1}, sv_0..sv_{T-1}]
 * (dynamics.py:98: reshape(5, 2, order="F")); T is a run-time parameter (the reference hard-codes
 * T = 5 and N = 9).  dyn_params_host[13] = [mu, m, I, lf, lr, C_Sf, C_Sr, h, dt, sv_max, a_max,
 * s_max, v_max] (dynamics.py:24-36); ignored (may be NULL) for FULLINT and SPIRAL.
 */
int irbfn_rollout_state_dim(int mode);              /* 7, 7, 5, 8, 6 */
int irbfn_rollout_input_dim(int mode, int T);       /* columns of x0u */
int irbfn_rollout_forward(int mode, const float* x0u_dev, const float* dyn_params_host,
                          float* states_dev, int64_t B, int T, void* stream);

/* VJP of the roll-out w.r.t. its input row: gstates_dev[B,T,S] -> g_x0u_dev[B, input_dim].
 * Replaces JAX's transpose of the scans under value_and_grad (scripts/train_nmpc.py:275-276,
 * :356-374; scripts/train_nmpc_frenet.py:408-409; deprecated/train_newlut.py:194-199).
 * clip() passes gradient 1 strictly inside its bounds, 0 strictly outside and `clip_tie` (0, 0.5 or
 * 1) exactly on a bound.  ST_SELECT returns IRBFN_ERR_UNSUPPORTED: the reference never
 * differentiates integrate_st_mult, and jax.grad through its lax.select is NaN at V = 0. */
int irbfn_rollout_vjp(int mode, const float* x0u_dev, const float* dyn_params_host,
                      const float* gstates_dev, float* g_x0u_dev, int64_t B, int T, float clip_tie,
                      void* stream);

/* Fused planning tick: net forward + roll-out in one launch, the predicted controls never leave
 * the chip.  Batched form of IRBFNPlanner.plan's `pred_step` -> `hstack` -> `integrate_st_mult`
 * (src/irbfn_mpc/irbfn_planner.py:205-212).  x_dev[B,D] queries, state0_dev[B,S] initial states,
 * requires O = 2T.  controls_dev may be NULL (then only states are written). */
int irbfn_net_forward_rollout(irbfn_net* net, int mode, const float* x_dev, const float* state0_dev,
                              const float* dyn_params_host, float* controls_dev, float* states_dev,
                              int64_t B, int T, void* stream);
/* 1 if irbfn_net_forward_rollout / irbfn_plan_tick with a states output run forward -> roll-out through the caller's
 * controls buffer for this net, mode, batch and horizon (wide outputs need it, narrow ones on the matrix-core kernel are
 * faster with it), 0 if the tick is one launch and controls_dev may be NULL, < 0 on a bad argument. */
int irbfn_net_tick_needs_controls(irbfn_net* net, int mode, int64_t B, int T);

/* ---------------------------------------------------------------------------------------------
 * Training-step pieces: the loss compositions that define the VJP seeds, and the
 * optimiser chain, on device -- a step needs no host round trip (the reference pulls the loss with
 * jax.device_get every step, scripts/train_nmpc.py:477-479).  Caller-allocated buffers only.
 *
 * irbfn_train_seeds_oneint: loss_fn of train_step_oneint (scripts/train_nmpc.py:268-295):
 *   loss = mean(l2(y_pred, y)) + mean(l2(onestep(x_pred_u)[:, [0,1,3,4]], onestep(x_u)[:, [0,1,3,4]])),
 *   initial state [0,0,0, x[:,0], 0, x[:,6], x[:,5]] (:260-266), one step = dynamic_st_onestep_aux.
 *   Writes gy = d loss / d y_pred [B,O] and the scalar loss (device).  D >= 7, O >= 2.
 * irbfn_train_seeds_fullint: loss_fn of train_step_fullint (scripts/train_nmpc.py:306-390):
 *   loss = mean|y_pred[:, [0,T]] - y[:, [0,T]]| + mean|final_pred - final_actual| (T-step inline bicycle), O = 2T.
 * partials_dev: scratch of irbfn_train_loss_partials() floats.
 * irbfn_adam_clip_step: optax.chain(clip_by_global_norm(max_grad_norm), adam(lr)) + apply_gradients
 *   (scripts/train_nmpc.py:231-233, :299) on flat float32 buffers of n elements; step_dev is the
 *   device-resident update count (incremented by the call).  max_grad_norm <= 0 disables clipping.
 */
int irbfn_train_loss_partials(void);
int irbfn_train_seeds_oneint(const float* x_dev, const float* y_pred_dev, const float* y_dev,
                             const float* dyn_params_host, float clip_tie, float* gy_dev, float* loss_dev,
                             float* partials_dev, int64_t B, int D, int O, void* stream);
int irbfn_train_seeds_fullint(const float* x_dev, const float* y_pred_dev, const float* y_dev, float clip_tie,
                              float* gy_dev, float* loss_dev, float* partials_dev, int64_t B, int D, int T,
                              void* stream);
/* irbfn_train_seeds_frenet_fullint: loss_fn of the Frenet train_step_fullint (scripts/train_nmpc_frenet.py:394-421):
 *   x [B,8] = [ey, delta, vx_car, vy_car, vx_goal, wz, epsi, curv], initial_state = x[:, [0,0,1,2,3,5,6,7]] (:398),
 *   loss = mean|y_pred - y| + mean|integrate_frenet_mult([init, y_pred]) - integrate_frenet_mult([init, y])|, O = 2T,
 *   T <= 16 (the reference: 5).  Writes gy = d loss / d y_pred [B,2T] and the scalar loss. */
int irbfn_train_seeds_frenet_fullint(const float* x_dev, const float* y_pred_dev, const float* y_dev,
                                     const float* dyn_params_host, float clip_tie, float* gy_dev, float* loss_dev,
                                     float* partials_dev, int64_t B, int D, int T, void* stream);
int irbfn_adam_clip_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, int64_t n,
                         int* step_dev, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                         float* partials_dev, void* stream);

/* ---- Batched planner front / back end ------------------------------------------------------
 * Query construction + mirror trick of IRBFNPlanner.plan (src/irbfn_mpc/irbfn_planner.py:181-201):
 * pose [B,7] = [x, y, delta, v, theta, angv, beta] (:240), goal [B,4] = ref_point [x, y, theta, v] (:170-171),
 * float64 like the NumPy host code; -> x [B,7] = [v, x_g, y_g, t_g, v_g, beta, angv] float32, mirror [B]
 * (goal_local[1] < 0), optional state0 [B,7] = float32(pose). */
int irbfn_plan_queries_cartesian(const double* pose_dev, const double* goal_dev, float* x_dev, float* state0_dev,
                                 int32_t* mirror_dev, int64_t B, void* stream);
/* IRBFNFrenetPlanner.plan (irbfn_planner.py:456-502): frenet [B,8] = [s, ey, delta, vx, vy, wz, epsi, curv],
 * vx_goal [B] -> x [B,8] = [+-ey, delta, vx, +-vy, vx_goal, +-wz, +-epsi, curv], mirror = ey < -0.05,
 * optional state0 [B,8] = float32(frenet). */
int irbfn_plan_queries_frenet(const double* frenet_dev, const double* vx_goal_dev, float* x_dev, float* state0_dev,
                              int32_t* mirror_dev, int64_t B, void* stream);
/* One planning tick: pred_step -> negate the steer-velocity controls [T, 2T) of mirrored rows
 * (irbfn_planner.py:203-204, :487-488) -> roll-out from state0 (:205-212).  mirror_dev may be NULL (then
 * identical to irbfn_net_forward_rollout); states_dev may be NULL (controls only; state0/dyn unused). */
int irbfn_plan_tick(irbfn_net* net, int mode, const float* x_dev, const int32_t* mirror_dev, const float* state0_dev,
                    const float* dyn_params_host, float* controls_dev, float* states_dev, int64_t B, int T,
                    void* stream);
/* Explicit-MPC table look-up, grid form (src/irbfn_mpc/explicit_planner.py:165-175): per axis
 * idx_d = min(shape_d - 1, searchsorted(keys_d, x_d, side="right")); keys_dev = the D sorted key arrays
 * concatenated (float64), key_offsets_host[D+1], shape_host[D]; table_dev [prod(shape), OW] float32 (may
 * be NULL with out_dev NULL).  -> flat row index idx_dev [B] and, if out_dev, the gathered rows [B,OW]. */
int irbfn_lut_grid_lookup(const double* keys_dev, const int32_t* key_offsets_host, const int32_t* shape_host,
                          const float* table_dev, const double* x_dev, int64_t* idx_dev, float* out_dev, int64_t B,
                          int D, int OW, void* stream);
/* Explicit-MPC table look-up, nearest-neighbour form (scipy KDTree.query at explicit_planner.py:383):
 * idx = argmin_n ||inputs[n] - x_b||_2 over inputs_dev [N,D] (float32; ties -> lowest n), dist_dev [B]
 * (optional), out_dev [B,OW] = table_dev[idx] (optional).  D in {3,4,7,8}. */
int64_t irbfn_lut_nearest_workspace_bytes(int64_t N, int64_t B);
int irbfn_lut_nearest(const float* inputs_dev, const float* table_dev, const float* x_dev, int64_t* idx_dev,
                      float* dist_dev, float* out_dev, int64_t N, int64_t B, int D, int OW, void* ws_dev,
                      int64_t ws_bytes, void* stream);

/* Way-point geometry of the planners' pure-pursuit front end, batched over B query points against ONE piecewise-linear
 * trajectory [N,2] (float64 device arrays as the NumPy callers hold them; one wave per point):
 * irbfn_nearest_point = nearest_point (src/irbfn_mpc/planner_utils.py:109-146): projection [B,2], distance, t in [0,1]
 *   and segment index of the closest point (first minimum, as np.argmin);
 * irbfn_intersect_point = intersect_point (:149-233): first point one `radius` away along the trajectory from
 *   t_start (= i + t, may be NULL = 0), with wrap-around; found[b] = 0 where the reference returns (None, None, None)
 *   (then first_p / first_t are NaN); first_i may be -1 (the closing segment, :206). */
int irbfn_nearest_point(const double* points_dev, const double* trajectory_dev, double* proj_dev, double* dist_dev,
                        double* t_dev, int32_t* seg_dev, int64_t B, int N, void* stream);
int irbfn_intersect_point(const double* points_dev, const double* trajectory_dev, const double* t_start_dev, float radius,
                          int wrap, float* first_p_dev, int32_t* first_i_dev, float* first_t_dev, int32_t* found_dev,
                          int64_t B, int N, void* stream);

/* ClusterWCRBFNet (src/irbfn_mpc/model.py:341-414): the region weights are a learned softmax gate instead of the
 * tanh indicator.  irbfn_cluster_gate: logits = x Wc + bc [B,R] (second output of the reference module) and
 * gamma = softmax(logits) [B,R]; wc [D,R], bc [R] device pointers.  irbfn_net_forward_gamma: the fused
 * sum_r gamma[b,r] phi[b,r,k] -> Dense forward (model.py:405-412) with caller-provided gamma_dev [B,R], on a
 * descriptor created with R regions (its own gate tables are ignored). */
int irbfn_cluster_gate(const float* x_dev, const float* wc_dev, const float* bc_dev, float* logits_dev, float* gamma_dev,
                       int64_t B, int D, int R, void* stream);
int irbfn_net_forward_gamma(irbfn_net* net, const float* x_dev, const float* gamma_dev, float* out_dev, int64_t B,
                            void* stream);

/* VJP of ClusterWCRBFNet (the reference trains it: scripts/train_nmpc_frenet.py:424-453).
 * irbfn_net_vjp_gamma: irbfn_net_vjp with caller-provided region weights gamma_dev [B,R] (the softmax gate) instead of
 *   the tanh tables, plus -- if dgamma_dev is not NULL -- the cotangent of those weights,
 *   dgamma[b,r] = sum_k (gout W^T)[b,k] phi[b,r,k]   (O <= 16).
 * irbfn_cluster_gate_vjp: softmax + Dense backward of the gate (model.py:402-404): dlogits = gamma * (dgamma -
 *   <gamma, dgamma>) [+ glogits_dev, the direct cotangent of the logits; may be NULL] -> g_wc [D,R], g_bc [R];
 *   dlogits_dev [B,R] is scratch / output; workspace_dev: irbfn_cluster_gate_vjp_workspace_bytes(D, R) bytes (per-block
 *   partial sums, added in block order: deterministic).
 * irbfn_softmax_xent: optax.softmax_cross_entropy(logits, labels).mean() (train_nmpc_frenet.py:431) -> loss (added to
 *   *loss_dev if accumulate != 0) and glogits = d loss / d logits; partials_dev: irbfn_train_loss_partials() floats. */
int irbfn_net_vjp_gamma(irbfn_net* net, const float* x_dev, const float* gamma_dev, const float* gout_dev,
                        float* g_centers_dev, float* g_log_sigs_dev, float* g_kernel_dev, float* g_bias_dev, float* dgamma_dev,
                        int64_t B, void* workspace_dev, int64_t workspace_bytes, void* stream);
int64_t irbfn_cluster_gate_vjp_workspace_bytes(int D, int R);
int irbfn_cluster_gate_vjp(const float* x_dev, const float* gamma_dev, const float* dgamma_dev, const float* glogits_dev,
                           float* dlogits_dev, float* g_wc_dev, float* g_bc_dev, int64_t B, int D, int R,
                           void* workspace_dev, int64_t workspace_bytes, void* stream);
int irbfn_softmax_xent(const float* logits_dev, const float* labels_dev, float* glogits_dev, float* loss_dev,
                       float* partials_dev, int accumulate, int64_t B, int R, void* stream);

/* Dense head of DeeperWCRBFNet (src/irbfn_mpc/model.py:201-289; the model of IRBFNFrenetPlanner with
 * deeper=True, src/irbfn_mpc/irbfn_planner.py:286-298):  out = linear(relu(linear_pre2(relu(h1)))) with
 * h1 = linear_pre1(rbf_out) [B,H1] produced by irbfn_net_forward on a descriptor whose Dense layer is
 * linear_pre1.  w2[H1,H2], b2[H2], w3[H2,O], b3[O] device pointers; H1 = H2 = 64 (hard-coded in the
 * reference).  The VJP of the head is irbfn_mlp_head_vjp below. */
int irbfn_mlp_head_forward(const float* h1_dev, const float* w2_dev, const float* b2_dev, const float* w3_dev,
                           const float* b3_dev, float* out_dev, int64_t B, int H1, int H2, int O, void* stream);

/* VJP of that head (the reference trains the model: scripts/train_nmpc_frenet.py:339-421): gout [B,O] ->
 * gh1 [B,H1] (cotangent of linear_pre1's output = the seed of irbfn_net_vjp on the stage descriptor) and the
 * gradients of linear_pre2 (gw2 [H1,H2], gb2 [H2]) and linear (gw3 [H2,O], gb3 [O]).  O <= 16.  Deterministic. */
int64_t irbfn_mlp_head_vjp_workspace_bytes(int H1, int H2, int O);
int irbfn_mlp_head_vjp(const float* h1_dev, const float* w2_dev, const float* b2_dev, const float* w3_dev,
                       const float* gout_dev, float* gh1_dev, float* gw2_dev, float* gb2_dev, float* gw3_dev,
                       float* gb3_dev, int64_t B, int H1, int H2, int O, void* ws_dev, int64_t ws_bytes, void* stream);

/* float64 mode.  The reference trains and evaluates in float64 under --use_float64 (scripts/train_nmpc.py:41-42:
 * jax.config.update("jax_enable_x64", True); some of its checkpoints hold float64 centers / log_sigs).  These entry points
 * evaluate WCRBFNet.apply (model.py:169-198) and its parameter VJP (scripts/train_nmpc.py:297-298) in float64 on the f64
 * vector pipe, straight from the checkpoint-layout arrays: no descriptor, no packed records; the gate bounds are float64 too
 * (a float32 card would move gamma by delta * 4e-8).  All 13 bases.  The card is a host struct of sizes and DEVICE pointers:
 * lo / hi [nsplit][max_ranges], delta [nsplit], dim_ranges [n_ranges][nsplit] as for irbfn_net_create.
 * workspace: irbfn_f64_workspace_bytes(card, B, with_vjp) bytes (gamma [B][R], + the gradient slabs for the VJP).
 * irbfn_f64_vjp: cotangent gout [B,O] -> g_centers [R,K,D], g_log_sigs [R,K], g_kernel [K,O], g_bias [O]; O <= 16
 * (IRBFN_ERR_UNSUPPORTED above); deterministic.  Not a throughput path: plain lane-per-query / lane-per-centre kernels. */
typedef struct irbfn_f64_card {
  int D, R, K, O, basis, nsplit, max_ranges, n_ranges;
  const double* lo_dev;
  const double* hi_dev;
  const double* delta_dev;
  const int* dim_ranges_dev;
} irbfn_f64_card;
int64_t irbfn_f64_workspace_bytes(const irbfn_f64_card* card, int64_t B, int with_vjp);
int irbfn_f64_forward(const irbfn_f64_card* card, const double* centers_dev, const double* log_sigs_dev,
                      const double* kernel_dev, const double* bias_dev, const double* x_dev, double* out_dev, int64_t B,
                      void* workspace_dev, int64_t workspace_bytes, void* stream);
int irbfn_f64_vjp(const irbfn_f64_card* card, const double* centers_dev, const double* log_sigs_dev, const double* kernel_dev,
                  const double* x_dev, const double* gout_dev, double* g_centers_dev, double* g_log_sigs_dev,
                  double* g_kernel_dev, double* g_bias_dev, int64_t B, void* workspace_dev, int64_t workspace_bytes,
                  void* stream);

/* Diagnostics */
int irbfn_abi_version(void);
int irbfn_device_count(void);
int irbfn_last_hip_error(void);
const char* irbfn_strerror(int status);
/* Name + launch geometry of the kernel the last forward call on this net used (for bench/profiles). */
int irbfn_net_last_launch(const irbfn_net* net, char* name_buf, int name_len, int* grid, int* block);

#ifdef __cplusplus
}
#endif
#endif /* IRBFN_HIP_H_ */
