// K4: hand-derived VJPs of the roll-outs (reverse sweeps).  One lane per trajectory: the forward is
// recomputed, the few pre-step quantities the adjoint needs are parked in LDS ([T][NS][64], one
// column per lane -> conflict-free), then the sweep runs backwards accumulating the seeds of every
// step's state (`all_states` is the differentiated output).
//
// Replaces JAX's transpose of the scans under value_and_grad: scripts/train_nmpc.py:275-276
// (dynamic_st_onestep_aux), :356-374 (inline bicycle), scripts/train_nmpc_frenet.py:408-409
// (integrate_frenet_mult), deprecated/train_newlut.py:194-199 (integrate_path_mult).
// clip() gradient: 1 strictly inside, 0 strictly outside, `tie` on a bound (jnp.clip is
// minimum(maximum(.)), whose tie rule is 1/2).
#include "common.h"
#include "rollout_step.h"

namespace irbfn {

struct RollVjpArgs {
  const float* __restrict__ x0u;      // [B][L]
  const float* __restrict__ gstates;  // [B][T][S]
  float* __restrict__ gx0u;           // [B][L]
  long B;
  int T, L;
  float tie;
  DynParams dp;
};

__device__ __forceinline__ float clipgrad(float v, float lo, float hi, float tie) {
  return (v > lo && v < hi) ? 1.0f : ((v == lo || v == hi) ? tie : 0.0f);
}

// ---- single-track kinematic (dynamics.py:103-187 applied T times) --------------------------------
__global__ __launch_bounds__(64) void rollout_vjp_st_ks(const RollVjpArgs a) {
  extern __shared__ float lds[];                 // [T][3][64]: delta, V(raw), psi before step t
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  if (b >= a.B) return;
  const int T = a.T, L = a.L;
  const float* row = a.x0u + b * L;
  float* grow = a.gx0u + b * L;
  const float* gs = a.gstates + b * (long)T * 7;
  const float lf = a.dp.p[3], lr = a.dp.p[4], dt = a.dp.p[8], sv_max = a.dp.p[9], a_max = a.dp.p[10],
              s_max = a.dp.p[11], v_max = a.dp.p[12];
  const float Lw = lr + lf;
  float s[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) s[i] = row[i];
  for (int t = 0; t < T; ++t) {
    lds[(t * 3 + 0) * kWave + lane] = s[2];
    lds[(t * 3 + 1) * kWave + lane] = s[3];
    lds[(t * 3 + 2) * kWave + lane] = s[4];
    st_step<false>(s, row[7 + t], row[7 + T + t], a.dp);
  }
  float lam[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < 7; ++i) lam[i] += gs[t * 7 + i];
    const float d_raw = lds[(t * 3 + 0) * kWave + lane];
    const float v_raw = lds[(t * 3 + 1) * kWave + lane];
    const float psi = lds[(t * 3 + 2) * kWave + lane];
    const float DELTA = clipf(d_raw, -s_max, s_max), V = clipf(v_raw, -v_max, v_max);
    const float md = clipgrad(d_raw, -s_max, s_max, a.tie), mv = clipgrad(v_raw, -v_max, v_max, a.tie);
    const float ma = clipgrad(row[7 + t], -a_max, a_max, a.tie);
    const float ms = clipgrad(row[7 + T + t], -sv_max, sv_max, a.tie);
    const float cp = cosf(psi), sp = sinf(psi), td = tanf(DELTA);
    grow[7 + t] = ma * dt * lam[3];
    grow[7 + T + t] = ms * dt * lam[2];
    const float l2 = lam[2] + md * lam[4] * (V / Lw) * (1.0f + td * td) * dt;
    const float l3 = lam[3] + mv * dt * (lam[0] * cp + lam[1] * sp + lam[4] * td / Lw);
    const float l4 = lam[4] + dt * V * (-lam[0] * sp + lam[1] * cp);
    lam[2] = l2; lam[3] = l3; lam[4] = l4;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) grow[i] = lam[i];
}

// ---- inline kinematic bicycle (scripts/train_nmpc.py:329-374) ------------------------------------
__global__ __launch_bounds__(64) void rollout_vjp_fullint(const RollVjpArgs a) {
  extern __shared__ float lds[];                 // [T][3][64]: delta, v, yaw before step t
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  if (b >= a.B) return;
  const int T = a.T, L = a.L;
  const float DT = 0.1f, WB = 0.33f, VMAX = 7.0f, VMIN = 0.0f, SMAX = 0.4189f;
  const float* row = a.x0u + b * L;
  float* grow = a.gx0u + b * L;
  const float* gs = a.gstates + b * (long)T * 5;
  float s[5] = {0.0f, 0.0f, 0.0f, clipf(row[0], VMIN, VMAX), 0.0f};
  for (int t = 0; t < T; ++t) {
    lds[(t * 3 + 0) * kWave + lane] = s[2];
    lds[(t * 3 + 1) * kWave + lane] = s[3];
    lds[(t * 3 + 2) * kWave + lane] = s[4];
    fullint_step(s, row[1 + t], row[1 + T + t]);
  }
  float lam[5] = {0, 0, 0, 0, 0};
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < 5; ++i) lam[i] += gs[t * 5 + i];
    const float d0 = lds[(t * 3 + 0) * kWave + lane];
    const float v0 = lds[(t * 3 + 1) * kWave + lane];
    const float psi = lds[(t * 3 + 2) * kWave + lane];
    const float dpre = d0 + row[1 + T + t] * DT, vpre = v0 + row[1 + t] * DT;
    const float d1 = clipf(dpre, -SMAX, SMAX), v1 = clipf(vpre, VMIN, VMAX);
    const float md = clipgrad(dpre, -SMAX, SMAX, a.tie), mv = clipgrad(vpre, VMIN, VMAX, a.tie);
    const float td = tanf(d1), cp = cosf(psi), sp = sinf(psi);
    const float Ld = lam[2] + lam[4] * (v1 / WB) * (1.0f + td * td) * DT;   // cotangent on delta'
    const float Lv = lam[3] + lam[4] * td * DT / WB;                        // cotangent on v'
    grow[1 + t] = mv * Lv * DT;
    grow[1 + T + t] = md * Ld * DT;
    const float l2 = md * Ld;
    const float l3 = mv * Lv + DT * (lam[0] * cp + lam[1] * sp);
    const float l4 = lam[4] + DT * v0 * (-lam[0] * sp + lam[1] * cp);
    lam[2] = l2; lam[3] = l3; lam[4] = l4;
  }
  grow[0] = clipgrad(row[0], VMIN, VMAX, a.tie) * lam[3];
}

// ---- Frenet low-speed model (dynamics.py:190-290) --------------------------------------------------
__global__ __launch_bounds__(64) void rollout_vjp_frenet(const RollVjpArgs a) {
  extern __shared__ float lds[];                 // [T][4][64]: ey, delta, vx, epsi before step t
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  if (b >= a.B) return;
  const int T = a.T, L = a.L;
  const float* row = a.x0u + b * L;
  float* grow = a.gx0u + b * L;
  const float* gs = a.gstates + b * (long)T * 8;
  const float LF = a.dp.p[3], LR = a.dp.p[4], dt = a.dp.p[8], sv_max = a.dp.p[9], a_max = a.dp.p[10],
              s_max = a.dp.p[11];
  const float Lw = LR + LF;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = row[i];
  const float cur = s[7];
  for (int t = 0; t < T; ++t) {
    lds[(t * 4 + 0) * kWave + lane] = s[1];
    lds[(t * 4 + 1) * kWave + lane] = s[2];
    lds[(t * 4 + 2) * kWave + lane] = s[3];
    lds[(t * 4 + 3) * kWave + lane] = s[6];
    frenet_step(s, row[8 + t], row[8 + T + t], a.dp);
  }
  float lam[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) lam[i] += gs[t * 8 + i];
    const float ey = lds[(t * 4 + 0) * kWave + lane];
    const float d_raw = lds[(t * 4 + 1) * kWave + lane];
    const float vx = lds[(t * 4 + 2) * kWave + lane];
    const float epsi = lds[(t * 4 + 3) * kWave + lane];
    const float dc = clipf(d_raw, -s_max, s_max);
    const float md = clipgrad(d_raw, -s_max, s_max, a.tie);
    const float ma = clipgrad(row[8 + t], -a_max, a_max, a.tie);
    const float ms = clipgrad(row[8 + T + t], -sv_max, sv_max, a.tie);
    const float ce = cosf(epsi), se = sinf(epsi), td = tanf(dc);
    const float den = 1.0f - ey * cur;
    const float d0 = vx * ce / den;
    const float A = lam[0] * dt - lam[6] * dt * cur;      // total cotangent on d0
    grow[8 + t] = ma * dt * lam[3];
    grow[8 + T + t] = ms * dt * lam[2];
    const float l1 = lam[1] + A * (vx * ce * cur / (den * den));
    const float l2 = lam[2] + md * lam[6] * dt * vx * (1.0f + td * td) / Lw;
    const float l3 = lam[3] + A * ce / den + lam[1] * dt * se + lam[6] * dt * td / Lw;
    const float l6 = lam[6] + A * (-vx * se / den) + lam[1] * dt * vx * ce;
    const float l7 = lam[7] + A * (vx * ce * ey / (den * den)) - lam[6] * dt * d0;
    lam[1] = l1; lam[2] = l2; lam[3] = l3; lam[6] = l6; lam[7] = l7;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) grow[i] = lam[i];
}

// ---- cubic spiral (planner_utils.py:20-77); T = number of samples N --------------------------------
__global__ __launch_bounds__(64) void rollout_vjp_spiral(const RollVjpArgs a) {
  extern __shared__ float lds[];                 // [N][3][64]: theta_i, dx_i, dy_i
  const int lane = threadIdx.x;
  const long b = (long)blockIdx.x * kWave + lane;
  if (b >= a.B) return;
  const int N = a.T;
  const float* row = a.x0u + b * 5;
  float* grow = a.gx0u + b * 5;
  const float* gs = a.gstates + b * (long)N * 6;
  float q[5], c[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = row[i];
  spiral_coefs(q, c);
  const float slen = q[4];
  float st[6] = {0.0f, 0.0f, 0.0f, c[0], 0.0f, 0.0f};
  for (int i = 0; i < N; ++i) {
    spiral_step(st, c, slen, i, N);
    lds[(i * 3 + 0) * kWave + lane] = st[2];
    lds[(i * 3 + 1) * kWave + lane] = st[4];
    lds[(i * 3 + 2) * kWave + lane] = st[5];
  }
  float gc[4] = {0, 0, 0, 0};
  float g_s = 0.0f, ldx = 0.0f, ldy = 0.0f, lth = 0.0f;
  for (int i = N - 1; i >= 0; --i) {
    const float tau = (i < N - 1) ? ((float)i / (float)(N - 1)) : 1.0f;
    const float sk = (i < N - 1) ? slen * tau : slen;
    const float k = (float)(i + 1);
    const float th = lds[(i * 3 + 0) * kWave + lane];
    const float dx = lds[(i * 3 + 1) * kWave + lane];
    const float dy = lds[(i * 3 + 2) * kWave + lane];
    const float thp = i > 0 ? lds[((i - 1) * 3 + 0) * kWave + lane] : 0.0f;
    const float gx = gs[i * 6 + 0], gy = gs[i * 6 + 1], gth = gs[i * 6 + 2], gka = gs[i * 6 + 3],
                gdx = gs[i * 6 + 4], gdy = gs[i * 6 + 5];
    const float Gdx = gdx + ldx + sk * gx;
    const float Gdy = gdy + ldy + sk * gy;
    float gsk = gx * dx + gy * dy;
    const float Gth = gth + lth + (Gdx * (-sinf(th)) + Gdy * cosf(th)) / (2.0f * k);
    lth = (Gdx * (-sinf(thp)) + Gdy * cosf(thp)) / (2.0f * k);
    ldx = Gdx * (1.0f - 1.0f / k);
    ldy = Gdy * (1.0f - 1.0f / k);
    // theta = sum_j c_j sk^(j+1)/(j+1); kappa = sum_j c_j sk^j
    float pw = 1.0f, kap = 0.0f, dkap = 0.0f, pwm1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gc[j] += Gth * (pw * sk) / (float)(j + 1) + gka * pw;
      kap += c[j] * pw;
      dkap += (float)j * c[j] * pwm1;
      pwm1 = pw;
      pw = pw * sk;
    }
    gsk += Gth * kap + gka * dkap;
    g_s += gsk * tau;
  }
  // coefs -> (k0..k3, s): c_r = (PM_r . q) / s^r  (planner_utils.py:20-29)
  const float PM[4][4] = {{1.0f, 0.0f, 0.0f, 0.0f},
                          {-11.0f / 2, 9.0f, -9.0f / 2, 1.0f},
                          {9.0f, -45.0f / 2, 18.0f, -9.0f / 2},
                          {-9.0f / 2, 27.0f / 2, -27.0f / 2, 9.0f / 2}};
  float inv = 1.0f;
  float gq[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int m = 0; m < 4; ++m) gq[m] += gc[r] * PM[r][m] * inv;
    g_s += -(float)r * c[r] / slen * gc[r];
    inv = inv / slen;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) grow[m] = gq[m];
  grow[4] = g_s;
}

int launch_rollout_vjp(int mode, const float* x0u, const DynParams& dp, const float* gstates,
                       float* g_x0u, int64_t B, int T, float clip_tie, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  RollVjpArgs a;
  a.x0u = x0u; a.gstates = gstates; a.gx0u = g_x0u; a.B = (long)B; a.T = T;
  a.L = rollout_input_dim(mode, T); a.tie = clip_tie; a.dp = dp;
  int ns;
  switch (mode) {
    case IRBFN_ROLLOUT_ST_KS:
    case IRBFN_ROLLOUT_FULLINT:
    case IRBFN_ROLLOUT_SPIRAL: ns = 3; break;
    case IRBFN_ROLLOUT_FRENET_LS: ns = 4; break;
    default: return IRBFN_ERR_UNSUPPORTED;   // ST_SELECT: the reference never differentiates it (SURVEY B-5)
  }
  size_t lds = (size_t)(T > 0 ? T : 1) * ns * kWave * sizeof(float);
  if (lds > 150 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((B + kWave - 1) / kWave)), block(kWave);
#define IRBFN_RV(KERN)                                                                                         \
  do {                                                                                                         \
    if (lds > 48 * 1024)                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                     \
    hipLaunchKernelGGL(KERN, grid, block, lds, s, a);                                                          \
  } while (0)
  switch (mode) {
    case IRBFN_ROLLOUT_ST_KS: IRBFN_RV(rollout_vjp_st_ks); break;
    case IRBFN_ROLLOUT_FULLINT: IRBFN_RV(rollout_vjp_fullint); break;
    case IRBFN_ROLLOUT_FRENET_LS: IRBFN_RV(rollout_vjp_frenet); break;
    case IRBFN_ROLLOUT_SPIRAL: IRBFN_RV(rollout_vjp_spiral); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_RV
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

}  // namespace irbfn
