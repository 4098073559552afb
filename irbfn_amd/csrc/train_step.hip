// Training-step pieces that sit between the forward (K1) and the parameter VJP (K2), and after it:
// the loss compositions of the reference's train steps (they define the VJP seeds) and the optimiser
// chain, all on device so that a step needs no host round trip (the reference pulls the loss with
// jax.device_get every step, scripts/train_nmpc.py:477-479).
//
//   seeds_oneint  : loss_fn of train_step_oneint  (scripts/train_nmpc.py:268-295)
//   seeds_fullint : loss_fn of train_step_fullint (scripts/train_nmpc.py:306-390)
//   seeds_frenet_fullint : loss_fn of the Frenet train_step_fullint (scripts/train_nmpc_frenet.py:394-421)
//   adam_clip     : optax.chain(clip_by_global_norm(max_norm), adam(lr)) + apply_gradients
//                   (scripts/train_nmpc.py:231-233, :299)
// All reductions are two-stage with a fixed order (deterministic).
#include <string.h>

#include "common.h"
#include "rollout_adjoint.h"
#include "rollout_step.h"

namespace irbfn {

__device__ __forceinline__ float clipgrad_t(float v, float lo, float hi, float tie) {
  return (v > lo && v < hi) ? 1.0f : ((v == lo || v == hi) ? tie : 0.0f);
}

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) sm[t] += sm[t + w];
    __syncthreads();
  }
  const float r = sm[0];
  __syncthreads();
  return r;
}

// ---- train_step_oneint ----------------------------------------------------------------------------
// x[B,D>=7] = [v_c, x_g, y_g, t_g, v_g, beta, angv]; y_pred, y [B,O>=2] = (accel, steer-vel) in cols 0,1.
// initial_state = [0,0,0, x[:,0], 0, x[:,6], x[:,5]]                       (train_nmpc.py:260-266)
// loss = mean(0.5 (y_pred - y)^2) + mean(0.5 (s_pred - s_act)[:, [0,1,3,4]]^2)   (:286-292)
// gy = d loss / d y_pred (through dynamic_st_onestep_aux, dynamics.py:103-187).
__global__ __launch_bounds__(256) void seeds_oneint_kernel(const float* __restrict__ x, const float* __restrict__ yp,
                                                           const float* __restrict__ y, float* __restrict__ gy,
                                                           float* __restrict__ loss_part, long B, int D, int O,
                                                           DynParams dp, float tie) {
  __shared__ float sm[256];
  float lsum = 0.0f;
  const float inv_y = 1.0f / ((float)B * (float)O), inv_s = 1.0f / ((float)B * 4.0f);
  const float dt = dp.p[8], sv_max = dp.p[9], a_max = dp.p[10];
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
    const float* xb = x + b * D;
    float sp[7] = {0.0f, 0.0f, 0.0f, xb[0], 0.0f, xb[6], xb[5]};
    float sa[7] = {0.0f, 0.0f, 0.0f, xb[0], 0.0f, xb[6], xb[5]};
    const float ap = yp[b * O + 0], svp = yp[b * O + 1];
    st_step<false>(sp, ap, svp, dp);                       // predicted_integrated_states  (:276)
    st_step<false>(sa, y[b * O + 0], y[b * O + 1], dp);    // actual_integrated_states     (:275)
    float lam[7] = {0, 0, 0, 0, 0, 0, 0};
    const int idx[4] = {0, 1, 3, 4};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = sp[idx[k]] - sa[idx[k]];
      lsum += 0.5f * d * d * inv_s;
      lam[idx[k]] = d * inv_s;
    }
    for (int o = 0; o < O; ++o) {
      const float d = yp[b * O + o] - y[b * O + o];
      lsum += 0.5f * d * d * inv_y;
      gy[b * O + o] = d * inv_y;
    }
    // adjoint of the kinematic step w.r.t. its controls (oracle/hand_vjp.py: vjp_st_ks, T = 1)
    gy[b * O + 0] += clipgrad_t(ap, -a_max, a_max, tie) * dt * lam[3];
    gy[b * O + 1] += clipgrad_t(svp, -sv_max, sv_max, tie) * dt * lam[2];
  }
  const float tot = block_sum_256(lsum, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = tot;
}

// ---- train_step_fullint ----------------------------------------------------------------------------
// loss = mean|y_pred[:, [0,T]] - y[:, [0,T]]| + mean|final_pred - final_actual|      (train_nmpc.py:386-390;
// the middle term |first_pred - first_pred| is identically 0).  O = 2T.
// TS: the horizon as a compile-time constant (the reference's T = 5: every label / prediction access a register with a static index
// instead of a 16-way select chain), 0 = run-time T <= TMAX
template <int TMAX, int TS = 0>
__global__ __launch_bounds__(256) void seeds_fullint_kernel(const float* __restrict__ x, const float* __restrict__ yp,
                                                            const float* __restrict__ y, float* __restrict__ gy,
                                                            float* __restrict__ loss_part, long B, int D, int Targ,
                                                            float tie) {
  __shared__ float sm[256];
  const int T = TS ? TS : Targ;
  const int O = 2 * T;
  const float DT = 0.1f, WB = 0.33f, VMAX = 7.0f, VMIN = 0.0f, SMAX = 0.4189f;
  const float inv_y = 1.0f / ((float)B * 2.0f), inv_s = 1.0f / ((float)B * 5.0f);
  float lsum = 0.0f;
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
    const float v0 = clipf(x[b * D + 0], VMIN, VMAX);       // :319
    // the row's label and prediction in registers up front (short horizons: the reference's T = 5): the roll-outs below
    // then run without a load in their dependency chain (one load per step: 30 serial round trips per row before)
    constexpr bool REGS = TMAX <= 8;
    float yr[REGS ? 2 * TMAX : 1], pr[REGS ? 2 * TMAX : 1], gr[REGS ? 2 * TMAX : 1];
    if constexpr (REGS) {
#pragma unroll
      for (int o = 0; o < 2 * TMAX; ++o) {
        yr[o] = o < O ? y[b * O + o] : 0.0f;
        pr[o] = o < O ? yp[b * O + o] : 0.0f;
        gr[o] = 0.0f;
      }
    }
    auto lab = [&](int t, int half) -> float {             // control `t` of the label: a_t (half = 0) / sv_t (half = 1)
      if constexpr (REGS && TS != 0) {
        return yr[half * TS + t];                          // t and half are loop constants after unrolling
      } else if constexpr (REGS) {
        float v = 0.0f;
#pragma unroll
        for (int o = 0; o < 2 * TMAX; ++o) v = (o == half * T + t) ? yr[o] : v;
        return v;
      } else {
        return y[b * O + half * T + t];
      }
    };
    auto prd = [&](int t, int half) -> float {
      if constexpr (REGS && TS != 0) {
        return pr[half * TS + t];
      } else if constexpr (REGS) {
        float v = 0.0f;
#pragma unroll
        for (int o = 0; o < 2 * TMAX; ++o) v = (o == half * T + t) ? pr[o] : v;
        return v;
      } else {
        return yp[b * O + half * T + t];
      }
    };
    float sa[5] = {0.0f, 0.0f, 0.0f, v0, 0.0f};
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < T) fullint_step(sa, lab(t, 0), lab(t, 1));     // :329-347
    float sp[5] = {0.0f, 0.0f, 0.0f, v0, 0.0f};
    float pd[TMAX], pv[TMAX], pp[TMAX];                       // delta, v, yaw before step t
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
        pd[t] = sp[2]; pv[t] = sp[3]; pp[t] = sp[4];
        fullint_step(sp, prd(t, 0), prd(t, 1));               // :356-374
      }
    }
    float lam[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const float d = sp[i] - sa[i];
      lsum += fabsf(d) * inv_s;
      lam[i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_s;    // d|.| = sign
    }
    if constexpr (!REGS)
      for (int o = 0; o < O; ++o) gy[b * O + o] = 0.0f;
    auto put = [&](int o_idx, float v) {
      if constexpr (REGS) {
#pragma unroll
        for (int o = 0; o < 2 * TMAX; ++o) gr[o] = (o == o_idx) ? v : gr[o];
      } else {
        gy[b * O + o_idx] = v;
      }
    };
#pragma unroll
    for (int t = TMAX - 1; t >= 0; --t) {                    // reverse sweep (oracle/hand_vjp.py: vjp_fullint)
      if (t < T) {
        const float a = prd(t, 0), dv = prd(t, 1);
        const float dpre = pd[t] + dv * DT, vpre = pv[t] + a * DT;
        const float d1 = clipf(dpre, -SMAX, SMAX), v1 = clipf(vpre, VMIN, VMAX);
        const float md = clipgrad_t(dpre, -SMAX, SMAX, tie), mv = clipgrad_t(vpre, VMIN, VMAX, tie);
        float sn, cs;
        sincos_fast(pp[t], sn, cs);
        const float td = tan_fast(d1);
        const float Ld = lam[2] + lam[4] * (v1 / WB) * (1.0f + td * td) * DT;
        const float Lv = lam[3] + lam[4] * td * DT / WB;
        put(t, mv * Lv * DT);
        put(T + t, md * Ld * DT);
        const float l2 = md * Ld;
        const float l3 = mv * Lv + DT * (lam[0] * cs + lam[1] * sn);
        const float l4 = lam[4] + DT * pv[t] * (-lam[0] * sn + lam[1] * cs);
        lam[2] = l2; lam[3] = l3; lam[4] = l4;
      }
    }
    const int cols[2] = {0, T};                               // y_predictions[:, [0, 5]]  (:387)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float d = prd(0, k) - lab(0, k);                  // column k * T
      lsum += fabsf(d) * inv_y;
      const float sgn = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_y;
      if constexpr (REGS) {
#pragma unroll
        for (int o = 0; o < 2 * TMAX; ++o) gr[o] += (o == cols[k]) ? sgn : 0.0f;
      } else {
        gy[b * O + cols[k]] += sgn;
      }
    }
    if constexpr (REGS) {
#pragma unroll
      for (int o = 0; o < 2 * TMAX; ++o)
        if (o < O) gy[b * O + o] = gr[o];
    }
  }
  const float tot = block_sum_256(lsum, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = tot;
}

// ---- Frenet train_step_fullint (scripts/train_nmpc_frenet.py:394-421) -------------------------------------------
// x[B,8] = [ey, delta, vx_car, vy_car, vx_goal, wz, epsi, curv]; initial_state = x[:, [0,0,1,2,3,5,6,7]] (:398);
// loss = mean|y_pred - y| + mean|integrate_frenet_mult([init, y_pred]) - integrate_frenet_mult([init, y])|
// (:402-412; all T states, all 8 components).  gy = d loss / d y_pred through the T-step Frenet roll-out
// (rollout_adjoint.h: the adjoint of dynamics.py:190-290, low-speed RHS).  O = 2T.
template <int TMAX>
__global__ __launch_bounds__(256) void seeds_frenet_fullint_kernel(const float* __restrict__ x, const float* __restrict__ yp,
                                                                   const float* __restrict__ y, float* __restrict__ gy,
                                                                   float* __restrict__ loss_part, long B, int D, int T,
                                                                   DynParams dp, float tie) {
  __shared__ float sm[256];
  const int O = 2 * T;
  const float inv_y = 1.0f / ((float)B * (float)O), inv_s = 1.0f / ((float)B * (float)T * 8.0f);
  float lsum = 0.0f;
  for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long)gridDim.x * 256) {
    const float* xb = x + b * D;
    float sa[8] = {xb[0], xb[0], xb[1], xb[2], xb[3], xb[5], xb[6], xb[7]};
    float sp[8] = {xb[0], xb[0], xb[1], xb[2], xb[3], xb[5], xb[6], xb[7]};
    const float cur = sp[7];
    float park[TMAX][4], seed[TMAX][8];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
        frenet_step(sa, y[b * O + t], y[b * O + T + t], dp);        // actual_states (:407)
        vjp_park<IRBFN_ROLLOUT_FRENET_LS>(sp, park[t]);
        frenet_step(sp, yp[b * O + t], yp[b * O + T + t], dp);      // pred_states   (:408)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = sp[i] - sa[i];
          lsum += fabsf(d) * inv_s;
          seed[t][i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_s;      // d|.| = sign
        }
      }
    }
    float lam[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = TMAX - 1; t >= 0; --t) {
      if (t < T) {
#pragma unroll
        for (int i = 0; i < 8; ++i) lam[i] += seed[t][i];
        float ga, gsv;
        vjp_back_step<IRBFN_ROLLOUT_FRENET_LS>(park[t], yp[b * O + t], yp[b * O + T + t], lam, cur, tie, dp, ga, gsv);
        gy[b * O + t] = ga;
        gy[b * O + T + t] = gsv;
      }
    }
    for (int o = 0; o < O; ++o) {                             // pred_loss = |y_pred - y|.mean()  (:401)
      const float d = yp[b * O + o] - y[b * O + o];
      lsum += fabsf(d) * inv_y;
      gy[b * O + o] += (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * inv_y;
    }
  }
  const float tot = block_sum_256(lsum, sm);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
  __shared__ float sm[256];
  float v = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) v += part[i];
  const float tot = block_sum_256(v, sm);
  if (threadIdx.x == 0) out[0] = tot;
}

// ---- optimiser: clip_by_global_norm + adam ----------------------------------------------------------
// Also hands the incremented step count to adam_clip_kernel through part[kRedBlocks] (the buffer holds kSeedBlocksMax
// floats): every block of the Adam kernel reads THAT word, its thread (0, 0) stores it back to step[0] -- no block reads
// step[0] there, so no separate "bump" launch is needed.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part,
                                                             const int* __restrict__ step) {
  __shared__ float sm[256];
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<int*>(part)[kRedBlocks] = step[0] + 1;
  float v = 0.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) v += g[i] * g[i];
  const float tot = block_sum_256(v, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// optax.clip_by_global_norm: g <- g if ||g|| < max_norm else g / ||g|| * max_norm.
// optax.adam (scale_by_adam, eps_root = 0): m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ;
//   update = -lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps) ; t is the incremented count.
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        int* __restrict__ step, const float* __restrict__ part,
                                                        float lr, float b1, float b2, float eps, float max_norm) {
  __shared__ float sm[256];
  const float sq = block_sum_256(threadIdx.x < kRedBlocks ? part[threadIdx.x] : 0.0f, sm);   // same in every block
  const float gn = sqrtf(sq);
  const float scale = (max_norm > 0.0f && !(gn < max_norm)) ? max_norm / gn : 1.0f;
  const int t = reinterpret_cast<const int*>(part)[kRedBlocks];   // step[0] + 1, written by sqnorm_partial_kernel
  if (blockIdx.x == 0 && threadIdx.x == 0) step[0] = t;
  const float c1 = 1.0f - powf(b1, (float)t), c2 = 1.0f - powf(b2, (float)t);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr * (mi / c1) / (sqrtf(vi / c2) + eps);
  }
}

}  // namespace irbfn

using namespace irbfn;

extern "C" {

int irbfn_train_loss_partials(void) { return kSeedBlocksMax; }

// blocks of the loss / seed kernels: one row per thread (a second pass over the rows doubled the kernel: 18 -> 9 us at the
// reference's batch of 80000), grid-stride beyond kSeedBlocksMax * 256 rows
static int seed_blocks(int64_t B) {
  const int64_t nb = (B + 255) / 256;
  return nb < 1 ? 1 : (nb > kSeedBlocksMax ? kSeedBlocksMax : (int)nb);
}

int irbfn_train_seeds_oneint(const float* x_dev, const float* y_pred_dev, const float* y_dev,
                             const float* dyn_params_host, float clip_tie, float* gy_dev, float* loss_dev,
                             float* partials_dev, int64_t B, int D, int O, void* stream) {
  if (B < 0 || D < 7 || O < 2 || !dyn_params_host) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!x_dev || !y_pred_dev || !y_dev || !gy_dev)) return IRBFN_ERR_BAD_ARG;
  if (!loss_dev || !partials_dev) return IRBFN_ERR_BAD_ARG;
  DynParams dp;
  memcpy(dp.p, dyn_params_host, sizeof(dp.p));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(seeds_oneint_kernel, dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                     partials_dev, (long)B, D, O, dp, clip_tie);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, partials_dev, seed_blocks(B), loss_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_train_seeds_fullint(const float* x_dev, const float* y_pred_dev, const float* y_dev, float clip_tie,
                              float* gy_dev, float* loss_dev, float* partials_dev, int64_t B, int D, int T,
                              void* stream) {
  if (B < 0 || D < 1 || T < 1) return IRBFN_ERR_BAD_ARG;
  if (T > 64) return IRBFN_ERR_UNSUPPORTED;
  if (B > 0 && (!x_dev || !y_pred_dev || !y_dev || !gy_dev)) return IRBFN_ERR_BAD_ARG;
  if (!loss_dev || !partials_dev) return IRBFN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (T == 5)                                                // the reference's horizon (train_nmpc.py:306-374)
    hipLaunchKernelGGL((seeds_fullint_kernel<8, 5>), dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                       partials_dev, (long)B, D, T, clip_tie);
  else if (T <= 8)
    hipLaunchKernelGGL((seeds_fullint_kernel<8>), dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                       partials_dev, (long)B, D, T, clip_tie);
  else
    hipLaunchKernelGGL((seeds_fullint_kernel<64>), dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                       partials_dev, (long)B, D, T, clip_tie);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, partials_dev, seed_blocks(B), loss_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_train_seeds_frenet_fullint(const float* x_dev, const float* y_pred_dev, const float* y_dev,
                                     const float* dyn_params_host, float clip_tie, float* gy_dev, float* loss_dev,
                                     float* partials_dev, int64_t B, int D, int T, void* stream) {
  if (B < 0 || D < 8 || T < 1 || T > 16 || !dyn_params_host) return IRBFN_ERR_BAD_ARG;
  if (B > 0 && (!x_dev || !y_pred_dev || !y_dev || !gy_dev)) return IRBFN_ERR_BAD_ARG;
  if (!loss_dev || !partials_dev) return IRBFN_ERR_BAD_ARG;
  DynParams dp;
  memcpy(dp.p, dyn_params_host, sizeof(dp.p));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (T <= 5)
    hipLaunchKernelGGL((seeds_frenet_fullint_kernel<5>), dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                       partials_dev, (long)B, D, T, dp, clip_tie);
  else
    hipLaunchKernelGGL((seeds_frenet_fullint_kernel<16>), dim3(seed_blocks(B)), dim3(256), 0, s, x_dev, y_pred_dev, y_dev, gy_dev,
                       partials_dev, (long)B, D, T, dp, clip_tie);
  IRBFN_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, partials_dev, seed_blocks(B), loss_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int irbfn_adam_clip_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, int64_t n,
                         int* step_dev, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                         float* partials_dev, void* stream) {
  if (n < 0 || !step_dev || !partials_dev) return IRBFN_ERR_BAD_ARG;
  if (n > 0 && (!params_dev || !grads_dev || !m_dev || !v_dev)) return IRBFN_ERR_BAD_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(kRedBlocks), dim3(256), 0, s, grads_dev, (long)n, partials_dev, step_dev);
  IRBFN_HIP_CHECK(hipGetLastError());
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, s, params_dev, grads_dev, m_dev, v_dev,
                     (long)n, step_dev, partials_dev, lr, beta1, beta2, eps, max_grad_norm);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

}  // extern "C"
