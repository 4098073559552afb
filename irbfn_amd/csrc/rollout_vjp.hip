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
#include "rollout_adjoint.h"
#include "rollout_pair.h"

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

// ---- cubic spiral, staged (N <= 256) -----------------------------------------------------------------------------------
// The kernel above reads every seed as a per-lane dword at a stride of N * 24 bytes and parks theta / dx / dy of all N
// samples in LDS (measured 0.75 TB/s at B = 262144, N = 100).  Here one wave owns 64 consecutive paths -- their seeds are ONE
// contiguous block of HBM -- and walks the samples in chunks of G = 8, last chunk first:
//  * the chunk's seeds (64 rows x 48 floats) are read by the whole wave as consecutive dwords (256 contiguous bytes per
//    instruction, runs of 192 bytes per row) into an LDS tile of odd pitch and read back per row without bank conflicts;
//  * no park of all samples: the forward pass keeps a checkpoint (theta, dx, dy before the chunk: 3 floats) per chunk in LDS,
//    the reverse pass re-runs one chunk from its checkpoint into registers -- the forward's own values, same step function --
//    and sweeps it backwards with the adjoint of the kernel above.
constexpr int kSpiralG = 8;
constexpr int kSpiralPitch = kSpiralG * 6 + 1;     // odd
__global__ __launch_bounds__(64) void rollout_vjp_spiral_staged(const RollVjpArgs a) {
  extern __shared__ float lds[];
  constexpr int G = kSpiralG, PITCH = kSpiralPitch;
  const int lane = threadIdx.x;
  const long b0 = (long)blockIdx.x * kWave;
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const int N = a.T;
  const int nch = (N + G - 1) / G;
  float* tile = lds;                             // [64][PITCH]
  float* ck = lds + kWave * PITCH;               // [nch][3][64]
  const long b = b0 + (lane < nvalid ? lane : nvalid - 1);
  const float* row = a.x0u + b * 5;
  float q[5], c[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) q[i] = row[i];
  spiral_coefs(q, c);
  const float slen = q[4];
  {                                              // forward pass: checkpoints only
    float st[6] = {0.0f, 0.0f, 0.0f, c[0], 0.0f, 0.0f};
    float sc[2] = {0.0f, 1.0f};
    for (int i = 0; i < N; ++i) {
      if (i % G == 0) {
        const int g = i / G;
        ck[(g * 3 + 0) * kWave + lane] = st[2];
        ck[(g * 3 + 1) * kWave + lane] = st[4];
        ck[(g * 3 + 2) * kWave + lane] = st[5];
      }
      spiral_step(st, c, slen, i, N, sc);
    }
  }
  const float* gs_tile = a.gstates + b0 * (long)N * 6;
  const long rs = (long)N * 6;
  float gc[4] = {0, 0, 0, 0};
  float g_s = 0.0f, ldx = 0.0f, ldy = 0.0f, lth = 0.0f;
#pragma unroll 1
  for (int g = nch - 1; g >= 0; --g) {
    const int i0 = g * G;
    const int n = (N - i0) < G ? (N - i0) : G;
    const int nf = n * 6;
    // the chunk's seeds: consecutive lanes = consecutive floats of a row's segment
    const float rnf = 1.0f / (float)nf;           // idx < 64 * 48: (idx + 0.5) / nf in float32 floors to the exact quotient
    for (int idx = lane; idx < nvalid * nf; idx += kWave) {
      const int r = (int)(((float)idx + 0.5f) * rnf), j = idx - r * nf;
      tile[r * PITCH + j] = gs_tile[r * rs + (long)i0 * 6 + j];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // re-run the chunk from its checkpoint
    float st[6] = {0.0f, 0.0f, ck[(g * 3 + 0) * kWave + lane], 0.0f, ck[(g * 3 + 1) * kWave + lane], ck[(g * 3 + 2) * kWave + lane]};
    // parked per sample: sin / cos of its heading and of the previous one (the step's own evaluations: the adjoint below
    // needs no trigonometry of its own -- four ocml sinf / cosf calls per sample made this kernel compute-bound), dx, dy
    float psn[G], pcs[G], psp[G], pcp[G], pdx[G], pdy[G];
    float sc[2];
    sincos_fast(st[2], sc[0], sc[1]);
#pragma unroll
    for (int tt = 0; tt < G; ++tt) {
      psp[tt] = sc[0]; pcp[tt] = sc[1];
      if (tt < n) spiral_step(st, c, slen, i0 + tt, N, sc);
      psn[tt] = sc[0]; pcs[tt] = sc[1]; pdx[tt] = st[4]; pdy[tt] = st[5];
    }
    const float* mine = tile + (lane < nvalid ? lane : nvalid - 1) * PITCH;
#pragma unroll
    for (int tt = G - 1; tt >= 0; --tt) {
      if (tt < n) {
        const int i = i0 + tt;
        const float tau = (i < N - 1) ? fdiv_fast((float)i, (float)(N - 1)) : 1.0f;    // as spiral_step
        const float sk = (i < N - 1) ? slen * tau : slen;
        const float k = (float)(i + 1);
        const float rk = fdiv_fast(1.0f, k);
        const float dx = pdx[tt], dy = pdy[tt];
        const float gx = mine[tt * 6 + 0], gy = mine[tt * 6 + 1], gth = mine[tt * 6 + 2], gka = mine[tt * 6 + 3],
                    gdx = mine[tt * 6 + 4], gdy = mine[tt * 6 + 5];
        const float Gdx = gdx + ldx + sk * gx;
        const float Gdy = gdy + ldy + sk * gy;
        float gsk = gx * dx + gy * dy;
        const float Gth = gth + lth + (Gdx * (-psn[tt]) + Gdy * pcs[tt]) * (0.5f * rk);
        lth = (Gdx * (-psp[tt]) + Gdy * pcp[tt]) * (0.5f * rk);     // sample 0: previous heading 0 -> (sin, cos) = (0, 1)
        ldx = Gdx * (1.0f - rk);
        ldy = Gdy * (1.0f - rk);
        const float rj[4] = {1.0f, 0.5f, 1.0f / 3.0f, 0.25f};
        float pw = 1.0f, kap = 0.0f, dkap = 0.0f, pwm1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gc[j] += Gth * (pw * sk) * rj[j] + gka * pw;
          kap += c[j] * pw;
          dkap += (float)j * c[j] * pwm1;
          pwm1 = pw;
          pw = pw * sk;
        }
        gsk += Gth * kap + gka * dkap;
        g_s += gsk * tau;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();               // the tile is overwritten by the next chunk
  }
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
  if (lane < nvalid) {
    float* grow = a.gx0u + (b0 + lane) * 5;
#pragma unroll
    for (int m = 0; m < 4; ++m) grow[m] = gq[m];
    grow[4] = g_s;
  }
}

// ---- K4: the same reverse sweeps with the memory machinery of the forward roll-out (T <= 50) ---------------------
// The kernels above touch HBM through per-lane strided dwords (measured 0.5-0.9 TB/s at T = 50).  Here:
//  * input rows: whole-tile LDS-DMA, the 2T controls of a trajectory live in registers (as in rollout.hip);
//  * no [T][3][64] park: the forward pass keeps a CHECKPOINT of the few state components the adjoint needs every
//    G = 10 steps (15 registers); the reverse pass re-runs one 10-step segment at a time into registers
//    (exactly the forward's values: same step function) and sweeps it backwards;
//  * the segment's seeds gstates[b, t0 .. t0+G, :] (G*S floats per row) are fetched as aligned 16-byte pieces by
//    the whole wave into an LDS tile and read back per row;
//  * the control gradients overwrite the controls IN PLACE in the registers -- the arrays rotate circularly by G
//    per segment so that every access has a static register index and the gradients end in natural order -- and
//    leave with the state cotangent as ONE contiguous tile through LDS (whole 128-byte lines, non-temporal).
typedef const __attribute__((address_space(1))) void* vgptr_t;
typedef __attribute__((address_space(3))) void* vlptr_t;
typedef float vf4 __attribute__((ext_vector_type(4)));

struct RollVjp2Args {
  const float* __restrict__ x0u;      // [B][L]
  const float* __restrict__ gstates;  // [B][T][S]
  float* __restrict__ gx0u;           // [B][L]
  long B;
  int T, L, wlds, dma_ok;
  float tie;
  DynParams dp;
};

constexpr int kVjpWaves = 2, kVjpRPP = 16;
constexpr int vjp_group(int TCH) { return TCH >= 10 ? 5 : TCH; }
// a row's seed chunk (G*S floats + up to 3 in front for the 16-byte alignment) as NPC 16-byte pieces; the tile is filled by
// LDS-DMA, so a row is NPC * 4 floats and NPC is odd (rows of an even number of pieces would fall on 4 or 8 banks)
constexpr int vjp_pieces(int S, int G) { return ((G * S + 6) / 4) | 1; }
constexpr int vjp_pitch(int S, int G) { return 4 * vjp_pieces(S, G); }

template <int MODE>
__device__ __forceinline__ void vjp_fwd_step(float* s, float a_in, float sv_in, const DynParams& dp) {
  if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(*reinterpret_cast<float(*)[7]>(s), a_in, sv_in, dp);
  else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(*reinterpret_cast<float(*)[5]>(s), a_in, sv_in);
  else frenet_step(*reinterpret_cast<float(*)[8]>(s), a_in, sv_in, dp);
}

template <int MODE, int TCH>
__global__ __launch_bounds__(64 * kVjpWaves, 2) void rollout_vjp_regs_kernel(const RollVjp2Args a) {
  extern __shared__ float lds[];
  constexpr int S = VjpTraits<MODE>::S, S0 = VjpTraits<MODE>::S0, NP = VjpTraits<MODE>::NP;
  constexpr int G = vjp_group(TCH), NG = TCH / G;
  constexpr int PITCH = vjp_pitch(S, G);
  constexpr int NPC = vjp_pieces(S, G);          // 16-byte pieces of a row's aligned seed chunk
  constexpr int RPP = kVjpRPP;
  static_assert(TCH % G == 0, "whole groups");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long b0 = ((long)blockIdx.x * kVjpWaves + wave) * kWave;
  if (b0 >= a.B) return;
  const long left = a.B - b0;
  const int nvalid = left < kWave ? (int)left : kWave;
  const int T = a.T, L = a.L;
  float* tile = lds + (size_t)wave * a.wlds;
  float* mine = tile + lane * PITCH;
  auto lds_drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  const bool dma = nvalid == kWave && a.dma_ok;

  // ---- prologue: my row -> registers ---------------------------------------------------------------------------
  float q0[S0], ua[TCH], us[TCH];
#pragma unroll
  for (int t = 0; t < TCH; ++t) { ua[t] = 0.0f; us[t] = 0.0f; }
  if (dma) {
#pragma unroll 1
    for (int p = 0; p < kWave / RPP; ++p) {
      const float* src = a.x0u + (b0 + (long)p * RPP) * L;
      const int nf = RPP * L;
      for (int v = lane * 4; v < nf; v += 256)
        __builtin_amdgcn_global_load_lds((vgptr_t)(src + v), (vlptr_t)(tile + (v - lane * 4)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if ((lane / RPP) == p) {
        const float* rr = tile + (lane % RPP) * L;
#pragma unroll
        for (int i = 0; i < S0; ++i) q0[i] = rr[i];
#pragma unroll
        for (int t = 0; t < TCH; ++t) { ua[t] = rr[S0 + t]; us[t] = rr[S0 + T + t]; }     // slots t >= T are never used
      }
      lds_drain();
    }
  } else {
    const float* row = a.x0u + (b0 + (lane < nvalid ? lane : nvalid - 1)) * L;
#pragma unroll
    for (int i = 0; i < S0; ++i) q0[i] = row[i];
#pragma unroll
    for (int t = 0; t < TCH; ++t)
      if (t < T) { ua[t] = row[S0 + t]; us[t] = row[S0 + T + t]; }
  }
  float s[S];
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(q0[0], 0.0f, 7.0f);
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = q0[i];
  }
  [[maybe_unused]] float cur = 0.0f;
  if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) cur = s[7];

  const float* gs_tile = a.gstates + b0 * (long)T * S;
  const long rs = (long)T * S;
  const int g4 = (int)((reinterpret_cast<uintptr_t>(gs_tile) >> 2) & 3);
  // The seeds of a group are requested ONE GROUP AHEAD by LDS-DMA (global_load_lds_dwordx4, gathered 16-byte pieces: no
  // staging VGPRs) into the other of two row tiles: fetched at the top of the group that needs them they cost half the
  // kernel (157 us without the seeds against 315 us with them, at B = 262144, T = 50) -- the latency of a dependent
  // global -> register -> LDS round trip per group with two waves per SIMD to hide it.
  float* tile2 = tile + kWave * PITCH;
  auto request = [&](int gq, float* dst) {       // seeds of group gq -> dst (asynchronous; retired by vmcnt)
    const int tq = gq * G;
    const int nq = (T - tq) < G ? (T - tq) : G;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int idx = j * kWave + lane;
      const int r = idx / NPC, part = idx - r * NPC;
      const int C = (g4 + (int)((r * rs + (long)tq * S) & 3)) & 3;
      const float* src = gs_tile + r * rs + (long)tq * S - C + 4 * part;        // 16-byte aligned, inside the tile
      if (4 * part < C + nq * S)
        __builtin_amdgcn_global_load_lds((vgptr_t)src, (vlptr_t)(dst + j * 256), 16, 0, 0);
    }
  };
  const int g_last = (T - 1) / G;                // the last group with steps
  int cbuf = 0;
  if (dma) request(g_last, tile);                // in flight during the whole forward pass

  // ---- pass 1: forward, one checkpoint per group; the arrays (and the checkpoint list) rotate DOWN circularly:
  //      static register indices inside the rolled loop, natural order after NG groups
  float ck[NG][NP];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < NP; ++i) ck[g][i] = 0.0f;
#pragma unroll 1
  for (int gI = 0; gI < NG; ++gI) {
    {
      float pk[NP];
      vjp_park<MODE>(s, pk);
#pragma unroll
      for (int g = 0; g + 1 < NG; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) ck[g][i] = ck[g + 1][i];
#pragma unroll
      for (int i = 0; i < NP; ++i) ck[NG - 1][i] = pk[i];     // after the remaining rotations: group g at ck[g]
    }
#pragma unroll
    for (int tt = 0; tt < G; ++tt)
      if (gI * G + tt < T) vjp_fwd_step<MODE>(s, ua[tt], us[tt], a.dp);
    if constexpr (NG > 1) {
      float ta[G], ts[G];
#pragma unroll
      for (int i = 0; i < G; ++i) { ta[i] = ua[i]; ts[i] = us[i]; }
#pragma unroll
      for (int i = 0; i + G < TCH; ++i) { ua[i] = ua[i + G]; us[i] = us[i + G]; }
#pragma unroll
      for (int i = 0; i < G; ++i) { ua[TCH - G + i] = ta[i]; us[TCH - G + i] = ts[i]; }
    }
  }

  // ---- pass 2: segments in reverse; the group's controls sit at [TCH - G, TCH) ---------------------------------
  float lam[S];
#pragma unroll
  for (int i = 0; i < S; ++i) lam[i] = 0.0f;
#pragma unroll 1
  for (int gI = NG - 1; gI >= 0; --gI) {
    const int t0 = gI * G;
    if (t0 < T) {                                // wave-uniform
      const int n = (T - t0) < G ? (T - t0) : G;
      float* cur_tile = cbuf ? tile2 : tile;
      const float* mine = cur_tile + lane * PITCH;
      const int myC = (g4 + (int)((lane * rs + (long)t0 * S) & 3)) & 3;
      if (dma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this group's seeds have landed
        __builtin_amdgcn_wave_barrier();
        if (gI > 0) request(gI - 1, cbuf ? tile : tile2);    // the other tile was last read two groups ago
      } else {
        const float* src = gs_tile + (lane < nvalid ? lane : nvalid - 1) * rs + (long)t0 * S;
        float* d = cur_tile + lane * PITCH + myC;
        for (int i = 0; i < n * S; ++i) d[i] = src[i];
      }
      cbuf ^= 1;
      lds_drain();
      // re-run the segment from its checkpoint: pre-step quantities of every step into registers
      float ss[S], park[G][NP], pk[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) pk[i] = ck[NG - 1][i];
#pragma unroll
      for (int i = 0; i < S; ++i) ss[i] = 0.0f;
      if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) { ss[1] = pk[0]; ss[2] = pk[1]; ss[3] = pk[2]; ss[6] = pk[3]; ss[7] = cur; }
      else { ss[2] = pk[0]; ss[3] = pk[1]; ss[4] = pk[2]; }
#pragma unroll
      for (int tt = 0; tt < G; ++tt) {
        vjp_park<MODE>(ss, park[tt]);
        if (tt < n) vjp_fwd_step<MODE>(ss, ua[TCH - G + tt], us[TCH - G + tt], a.dp);
      }
#pragma unroll
      for (int tt = G - 1; tt >= 0; --tt) {
        if (tt < n) {
#pragma unroll
          for (int i = 0; i < S; ++i) lam[i] += mine[myC + tt * S + i];
          float ga, gsv;
          vjp_back_step<MODE>(park[tt], ua[TCH - G + tt], us[TCH - G + tt], lam, cur, a.tie, a.dp, ga, gsv);
          ua[TCH - G + tt] = ga;                 // the control is dead from here on: its slot takes the gradient
          us[TCH - G + tt] = gsv;
        }
      }
      lds_drain();
    }
    if constexpr (NG > 1) {                      // rotate UP circularly: the next (earlier) group moves to [TCH - G, TCH)
#pragma unroll
      for (int g = NG - 1; g > 0; --g)
#pragma unroll
        for (int i = 0; i < NP; ++i) ck[g][i] = ck[g - 1][i];
      float ta[G], ts[G];
#pragma unroll
      for (int i = 0; i < G; ++i) { ta[i] = ua[TCH - G + i]; ts[i] = us[TCH - G + i]; }
#pragma unroll
      for (int i = TCH - 1; i >= G; --i) { ua[i] = ua[i - G]; us[i] = us[i - G]; }
#pragma unroll
      for (int i = 0; i < G; ++i) { ua[i] = ta[i]; us[i] = ts[i]; }
    }
  }
  // cotangent of the initial state
  float g0[S0];
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) g0[0] = clipgrad(q0[0], 0.0f, 7.0f, a.tie) * lam[3];
  else {
#pragma unroll
    for (int i = 0; i < S0; ++i) g0[i] = lam[i];
  }

  // ---- epilogue: gradient rows -> HBM, RPP rows per pass as one contiguous block of whole lines -----------------
  float* gout = a.gx0u + b0 * L;
  if (dma) {
#pragma unroll 1
    for (int p = 0; p < kWave / RPP; ++p) {
      if ((lane / RPP) == p) {
        float* rr = tile + (lane % RPP) * L;
#pragma unroll
        for (int i = 0; i < S0; ++i) rr[i] = g0[i];
#pragma unroll
        for (int t = 0; t < TCH; ++t)
          if (t < T) { rr[S0 + t] = ua[t]; rr[S0 + T + t] = us[t]; }
      }
      lds_drain();
      float* dst = gout + (long)p * RPP * L;
      const int nf = RPP * L;                    // multiple of 4 floats; dst is 16-byte aligned
      for (int v = lane * 4; v < nf; v += 256) {
        const vf4 val = *reinterpret_cast<const vf4*>(tile + v);
        __builtin_nontemporal_store(val, reinterpret_cast<vf4*>(dst + v));
      }
      lds_drain();
    }
  } else if (lane < nvalid) {
    float* grow = gout + (long)lane * L;
#pragma unroll
    for (int i = 0; i < S0; ++i) grow[i] = g0[i];
#pragma unroll
    for (int t = 0; t < TCH; ++t)
      if (t < T) { grow[S0 + t] = ua[t]; grow[S0 + T + t] = us[t]; }
  }
}

// ---- K4p: the same kernel with TWO LANES PER TRAJECTORY (as K3p of the forward, rollout_pair.h) ----------------------
// MEASURED AND NOT TAKEN (round 3, profiles/r03_rollout_vjp_pair_lanes.txt; build with -DIRBFN_VJP_PAIR=1 to A/B, results
// identical to K4 in every test): B = 262144, T = 50: ST kinematic 205.6 vs 212.6 us, inline bicycle 176.5 vs 138.0 us, Frenet
// 268.2 vs 259.7 us.  Three waves per SIMD instead of two, but twice the waves and the same dependent chain per wave: K4 is
// bound by the serial latency of a wave's three 50-step passes times the rounds of waves per SIMD, which pairing lanes does not
// shorten.
// K4 keeps the 2T controls of a trajectory in one lane's registers (100 at T = 50): 243 VGPRs, two waves per SIMD, and what
// remains is one wave's serial latency.  Here a trajectory sits on a lane pair: the even lane holds the acceleration knots, the
// odd lane the steering-rate knots (50 registers each; a step reads both by DPP), the step's trigonometry runs once per pair
// (TrigPair: even lane sin / cos of the heading, odd lane tan of the steering angle, swapped by DPP -- the same values bit for
// bit), everything else is evaluated by both lanes from the same operands; the control gradients replace the controls in
// place, each lane its own stream.  ~145 VGPRs -> three waves per SIMD, 32 trajectories per wave, half the LDS per wave.
constexpr int kVjpPairRows = 32;
template <int MODE, int TCH>
__global__ __launch_bounds__(64 * kVjpWaves, 3) void rollout_vjp_pair_kernel(const RollVjp2Args a) {
  extern __shared__ float lds[];
  constexpr int S = VjpTraits<MODE>::S, S0 = VjpTraits<MODE>::S0, NP = VjpTraits<MODE>::NP;
  constexpr int G = vjp_group(TCH), NG = TCH / G;
  constexpr int PITCH = vjp_pitch(S, G);
  constexpr int NPC = vjp_pieces(S, G);          // 16-byte pieces of a row's aligned seed chunk
  constexpr int RPP = kVjpRPP;
  constexpr int R = kVjpPairRows;
  constexpr int NRQ = (R * NPC + kWave - 1) / kWave;     // DMA instructions per seed request
  static_assert(TCH % G == 0, "whole groups");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int odd = lane & 1, prow = lane >> 1;
  const long b0 = ((long)blockIdx.x * kVjpWaves + wave) * R;
  if (b0 >= a.B) return;
  const long left = a.B - b0;
  const int nvalid = left < R ? (int)left : R;
  const int T = a.T, L = a.L;
  float* tile = lds + (size_t)wave * a.wlds;
  auto lds_drain = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  const bool dma = nvalid == R && a.dma_ok;
  const TrigPair trig{odd};

  // ---- prologue: my row -> registers (each lane ONE control stream) -------------------------------------------------
  float q0[S0], ctl[TCH];
#pragma unroll
  for (int t = 0; t < TCH; ++t) ctl[t] = 0.0f;
  if (dma) {
#pragma unroll 1
    for (int p = 0; p < R / RPP; ++p) {
      const float* src = a.x0u + (b0 + (long)p * RPP) * L;
      const int nf = RPP * L;
      for (int v = lane * 4; v < nf; v += 256)
        __builtin_amdgcn_global_load_lds((vgptr_t)(src + v), (vlptr_t)(tile + (v - lane * 4)), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if ((prow / RPP) == p) {
        const float* rr = tile + (prow % RPP) * L;
#pragma unroll
        for (int i = 0; i < S0; ++i) q0[i] = rr[i];
        const float* cr = rr + S0 + (odd ? T : 0);
#pragma unroll
        for (int t = 0; t < TCH; ++t) ctl[t] = cr[t];          // slots t >= T are never used (they stay inside the tile)
      }
      lds_drain();
    }
  } else {
    const float* row = a.x0u + (b0 + (prow < nvalid ? prow : nvalid - 1)) * L;
#pragma unroll
    for (int i = 0; i < S0; ++i) q0[i] = row[i];
#pragma unroll
    for (int t = 0; t < TCH; ++t)
      if (t < T) ctl[t] = row[S0 + (odd ? T : 0) + t];
  }
  float s[S];
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) {
    s[0] = 0.0f; s[1] = 0.0f; s[2] = 0.0f; s[4] = 0.0f;
    s[3] = clipf(q0[0], 0.0f, 7.0f);
  } else {
#pragma unroll
    for (int i = 0; i < S; ++i) s[i] = q0[i];
  }
  [[maybe_unused]] float cur = 0.0f;
  if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) cur = s[7];
  auto knots = [&](float c, float& ua, float& us) {          // the pair's two knots of one step: even lane's, odd lane's
    const int cv = __builtin_bit_cast(int, c);
    ua = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(cv, 0xA0, 0xF, 0xF, true));
    us = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(cv, 0xF5, 0xF, 0xF, true));
  };
  auto fwd = [&](float* st, float ua, float us) {
    if constexpr (MODE == IRBFN_ROLLOUT_ST_KS) st_step<false>(*reinterpret_cast<float(*)[7]>(st), ua, us, a.dp, trig);
    else if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) fullint_step(*reinterpret_cast<float(*)[5]>(st), ua, us, trig);
    else frenet_step(*reinterpret_cast<float(*)[8]>(st), ua, us, a.dp, trig);
  };

  const float* gs_tile = a.gstates + b0 * (long)T * S;
  const long rs = (long)T * S;
  const int g4 = (int)((reinterpret_cast<uintptr_t>(gs_tile) >> 2) & 3);
  float* tile2 = tile + R * PITCH;
  auto request = [&](int gq, float* dst) {       // seeds of group gq -> dst (asynchronous; retired by vmcnt)
    const int tq = gq * G;
    const int nq = (T - tq) < G ? (T - tq) : G;
#pragma unroll
    for (int j = 0; j < NRQ; ++j) {
      const int idx = j * kWave + lane;
      const int r = idx / NPC, part = idx - r * NPC;
      const int C = (g4 + (int)((r * rs + (long)tq * S) & 3)) & 3;
      const float* src = gs_tile + r * rs + (long)tq * S - C + 4 * part;        // 16-byte aligned, inside the tile
      if (idx < R * NPC && 4 * part < C + nq * S)
        __builtin_amdgcn_global_load_lds((vgptr_t)src, (vlptr_t)(dst + j * 256), 16, 0, 0);
    }
  };
  const int g_last = (T - 1) / G;                // the last group with steps
  int cbuf = 0;
  if (dma) request(g_last, tile);                // in flight during the whole forward pass

  // ---- pass 1: forward, one checkpoint per group; arrays rotate DOWN circularly (static register indices)
  float ck[NG][NP];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < NP; ++i) ck[g][i] = 0.0f;
#pragma unroll 1
  for (int gI = 0; gI < NG; ++gI) {
    {
      float pk[NP];
      vjp_park<MODE>(s, pk);
#pragma unroll
      for (int g = 0; g + 1 < NG; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) ck[g][i] = ck[g + 1][i];
#pragma unroll
      for (int i = 0; i < NP; ++i) ck[NG - 1][i] = pk[i];
    }
#pragma unroll
    for (int tt = 0; tt < G; ++tt) {
      if (gI * G + tt < T) {
        float ua, us;
        knots(ctl[tt], ua, us);
        fwd(s, ua, us);
      }
    }
    if constexpr (NG > 1) {
      float tc[G];
#pragma unroll
      for (int i = 0; i < G; ++i) tc[i] = ctl[i];
#pragma unroll
      for (int i = 0; i + G < TCH; ++i) ctl[i] = ctl[i + G];
#pragma unroll
      for (int i = 0; i < G; ++i) ctl[TCH - G + i] = tc[i];
    }
  }

  // ---- pass 2: segments in reverse; the group's controls sit at [TCH - G, TCH) ---------------------------------
  float lam[S];
#pragma unroll
  for (int i = 0; i < S; ++i) lam[i] = 0.0f;
#pragma unroll 1
  for (int gI = NG - 1; gI >= 0; --gI) {
    const int t0 = gI * G;
    if (t0 < T) {                                // wave-uniform
      const int n = (T - t0) < G ? (T - t0) : G;
      float* cur_tile = cbuf ? tile2 : tile;
      const float* mine = cur_tile + prow * PITCH;
      const int myC = (g4 + (int)((prow * rs + (long)t0 * S) & 3)) & 3;
      if (dma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this group's seeds have landed
        __builtin_amdgcn_wave_barrier();
        if (gI > 0) request(gI - 1, cbuf ? tile : tile2);    // the other tile was last read two groups ago
      } else {
        const float* src = gs_tile + (prow < nvalid ? prow : nvalid - 1) * rs + (long)t0 * S;
        float* d = cur_tile + prow * PITCH + myC;
        for (int i = odd; i < n * S; i += 2) d[i] = src[i];
      }
      cbuf ^= 1;
      lds_drain();
      float ss[S], park[G][NP], pk[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) pk[i] = ck[NG - 1][i];
#pragma unroll
      for (int i = 0; i < S; ++i) ss[i] = 0.0f;
      if constexpr (MODE == IRBFN_ROLLOUT_FRENET_LS) { ss[1] = pk[0]; ss[2] = pk[1]; ss[3] = pk[2]; ss[6] = pk[3]; ss[7] = cur; }
      else { ss[2] = pk[0]; ss[3] = pk[1]; ss[4] = pk[2]; }
#pragma unroll
      for (int tt = 0; tt < G; ++tt) {
        vjp_park<MODE>(ss, park[tt]);
        if (tt < n) {
          float ua, us;
          knots(ctl[TCH - G + tt], ua, us);
          fwd(ss, ua, us);
        }
      }
#pragma unroll
      for (int tt = G - 1; tt >= 0; --tt) {
        if (tt < n) {
#pragma unroll
          for (int i = 0; i < S; ++i) lam[i] += mine[myC + tt * S + i];
          float ga, gsv, ua, us;
          knots(ctl[TCH - G + tt], ua, us);      // still the knots: the sweep has only overwritten the slots behind tt
          vjp_back_step<MODE>(park[tt], ua, us, lam, cur, a.tie, a.dp, ga, gsv, trig);
          ctl[TCH - G + tt] = odd ? gsv : ga;    // the knot is dead from here on: its slot takes this lane's gradient
        }
      }
      lds_drain();
    }
    if constexpr (NG > 1) {                      // rotate UP circularly: the next (earlier) group moves to [TCH - G, TCH)
#pragma unroll
      for (int g = NG - 1; g > 0; --g)
#pragma unroll
        for (int i = 0; i < NP; ++i) ck[g][i] = ck[g - 1][i];
      float tc[G];
#pragma unroll
      for (int i = 0; i < G; ++i) tc[i] = ctl[TCH - G + i];
#pragma unroll
      for (int i = TCH - 1; i >= G; --i) ctl[i] = ctl[i - G];
#pragma unroll
      for (int i = 0; i < G; ++i) ctl[i] = tc[i];
    }
  }
  float g0[S0];
  if constexpr (MODE == IRBFN_ROLLOUT_FULLINT) g0[0] = clipgrad(q0[0], 0.0f, 7.0f, a.tie) * lam[3];
  else {
#pragma unroll
    for (int i = 0; i < S0; ++i) g0[i] = lam[i];
  }

  // ---- epilogue: gradient rows -> HBM, RPP rows per pass as one contiguous block of whole lines -----------------
  float* gout = a.gx0u + b0 * L;
  if (dma) {
#pragma unroll 1
    for (int p = 0; p < R / RPP; ++p) {
      if ((prow / RPP) == p) {
        float* rr = tile + (prow % RPP) * L;
        if (!odd) {
#pragma unroll
          for (int i = 0; i < S0; ++i) rr[i] = g0[i];
        }
        float* cr = rr + S0 + (odd ? T : 0);
#pragma unroll
        for (int t = 0; t < TCH; ++t)
          if (t < T) cr[t] = ctl[t];
      }
      lds_drain();
      float* dst = gout + (long)p * RPP * L;
      const int nf = RPP * L;                    // multiple of 4 floats; dst is 16-byte aligned
      for (int v = lane * 4; v < nf; v += 256) {
        const vf4 val = *reinterpret_cast<const vf4*>(tile + v);
        __builtin_nontemporal_store(val, reinterpret_cast<vf4*>(dst + v));
      }
      lds_drain();
    }
  } else if (prow < nvalid) {
    float* grow = gout + (long)prow * L;
    if (!odd) {
#pragma unroll
      for (int i = 0; i < S0; ++i) grow[i] = g0[i];
    }
#pragma unroll
    for (int t = 0; t < TCH; ++t)
      if (t < T) grow[S0 + (odd ? T : 0) + t] = ctl[t];
  }
}

template <int MODE>
static int launch_vjp_pair(const float* x0u, const DynParams& dp, const float* gstates, float* g_x0u, int64_t B, int T,
                           float tie, hipStream_t s) {
  constexpr int S = VjpTraits<MODE>::S;
  constexpr int TCH = 50;
  RollVjp2Args a;
  a.x0u = x0u; a.gstates = gstates; a.gx0u = g_x0u; a.B = (long)B; a.T = T; a.L = rollout_input_dim(MODE, T);
  a.tie = tie; a.dp = dp;
  const int G = vjp_group(TCH);
  long w = 2L * kVjpPairRows * vjp_pitch(S, G);
  // the prologue reads TCH knots per lane from a staged row whatever T is: keep those reads inside the wave's tile
  const long stage = (long)kVjpRPP * a.L + TCH;
  if (stage > w) w = stage;
  a.wlds = (int)((w + 3) & ~3L);
  a.dma_ok = ((reinterpret_cast<uintptr_t>(x0u) | reinterpret_cast<uintptr_t>(gstates) | reinterpret_cast<uintptr_t>(g_x0u)) & 15) == 0 &&
             ((kVjpRPP * a.L) % 4) == 0;
  const size_t lds = (size_t)kVjpWaves * a.wlds * sizeof(float);
  if (lds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long waves = (B + kVjpPairRows - 1) / kVjpPairRows;
  const dim3 grid((unsigned)((waves + kVjpWaves - 1) / kVjpWaves)), block(kWave * kVjpWaves);
  hipLaunchKernelGGL((rollout_vjp_pair_kernel<MODE, TCH>), grid, block, lds, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int MODE>
static int launch_vjp_regs(const float* x0u, const DynParams& dp, const float* gstates, float* g_x0u, int64_t B, int T,
                           float tie, hipStream_t s) {
  constexpr int S = VjpTraits<MODE>::S;
  RollVjp2Args a;
  a.x0u = x0u; a.gstates = gstates; a.gx0u = g_x0u; a.B = (long)B; a.T = T; a.L = rollout_input_dim(MODE, T);
  a.tie = tie; a.dp = dp;
  const int TCH = T <= 8 ? 8 : 50;
  const int G = vjp_group(TCH);
  long w = (TCH / G > 1 ? 2L : 1L) * kWave * vjp_pitch(S, G);     // two seed tiles where there is a next group to prefetch
  if ((long)kVjpRPP * a.L > w) w = (long)kVjpRPP * a.L;
  a.wlds = (int)((w + 3) & ~3L);
  a.dma_ok = ((reinterpret_cast<uintptr_t>(x0u) | reinterpret_cast<uintptr_t>(gstates) | reinterpret_cast<uintptr_t>(g_x0u)) & 15) == 0;
  const size_t lds = (size_t)kVjpWaves * a.wlds * sizeof(float);
  if (lds > 64 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long waves = (B + kWave - 1) / kWave;
  const dim3 grid((unsigned)((waves + kVjpWaves - 1) / kVjpWaves)), block(kWave * kVjpWaves);
  if (TCH == 8) hipLaunchKernelGGL((rollout_vjp_regs_kernel<MODE, 8>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((rollout_vjp_regs_kernel<MODE, 50>), grid, block, lds, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

int launch_rollout_vjp(int mode, const float* x0u, const DynParams& dp, const float* gstates,
                       float* g_x0u, int64_t B, int T, float clip_tie, hipStream_t s) {
  if (B == 0) return IRBFN_OK;
  RollVjpArgs a;
  a.x0u = x0u; a.gstates = gstates; a.gx0u = g_x0u; a.B = (long)B; a.T = T;
  a.L = rollout_input_dim(mode, T); a.tie = clip_tie; a.dp = dp;
#ifndef IRBFN_VJP_PAIR
#define IRBFN_VJP_PAIR 0
#endif
  if (IRBFN_VJP_PAIR && T > 8 && T <= 50) {      // K4p: two lanes per trajectory (A/B builds only, see the note at the kernel)
    int rc = IRBFN_ERR_UNSUPPORTED;
    if (mode == IRBFN_ROLLOUT_ST_KS) rc = launch_vjp_pair<IRBFN_ROLLOUT_ST_KS>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    else if (mode == IRBFN_ROLLOUT_FULLINT) rc = launch_vjp_pair<IRBFN_ROLLOUT_FULLINT>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    else if (mode == IRBFN_ROLLOUT_FRENET_LS) rc = launch_vjp_pair<IRBFN_ROLLOUT_FRENET_LS>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  }
  if (T >= 1 && T <= 50) {                       // K4 with staged memory traffic (the kernels below: longer horizons)
    int rc = IRBFN_ERR_UNSUPPORTED;
    if (mode == IRBFN_ROLLOUT_ST_KS) rc = launch_vjp_regs<IRBFN_ROLLOUT_ST_KS>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    else if (mode == IRBFN_ROLLOUT_FULLINT) rc = launch_vjp_regs<IRBFN_ROLLOUT_FULLINT>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    else if (mode == IRBFN_ROLLOUT_FRENET_LS) rc = launch_vjp_regs<IRBFN_ROLLOUT_FRENET_LS>(x0u, dp, gstates, g_x0u, B, T, clip_tie, s);
    if (rc != IRBFN_ERR_UNSUPPORTED) return rc;
  }
  if (mode == IRBFN_ROLLOUT_SPIRAL && T >= 1 && T <= 256) {          // 37 KB of LDS at N = 256
    const size_t ldss = ((size_t)kWave * kSpiralPitch + (size_t)((T + kSpiralG - 1) / kSpiralG) * 3 * kWave) * sizeof(float);
    hipLaunchKernelGGL(rollout_vjp_spiral_staged, dim3((unsigned)((B + kWave - 1) / kWave)), dim3(kWave), ldss, s, a);
    IRBFN_HIP_CHECK(hipGetLastError());
    return IRBFN_OK;
  }
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
      IRBFN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN),                                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
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
