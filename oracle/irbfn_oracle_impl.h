/* CPU oracle (C restatement) of the IRBFN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Included twice by irbfn_oracle.c with REAL = float / double and SUF = f32 / f64.
 * Follows the same reference lines as oracle/irbfn_oracle.py (see that header for the
 * pinning status: roll-out B pinned by KAT-1/2/3; everything else "parity unpinned").
 * Citations are path:line relative to the reference root.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#if REAL_IS_FLOAT
#define R_SIN sinf
#define R_COS cosf
#define R_TAN tanf
#define R_TANH tanhf
#define R_EXP expf
#define R_SQRT sqrtf
#define R_LOG logf
#else
#define R_SIN sin
#define R_COS cos
#define R_TAN tan
#define R_TANH tanh
#define R_EXP exp
#define R_SQRT sqrt
#define R_LOG log
#endif

static inline REAL FN(clipr)(REAL v, REAL lo, REAL hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* a-2: deprecated/f1tenth_gym/examples/flax_rbf/flax_rbf/flax_rbf.py:34-111 */
static inline REAL FN(basis_eval)(int basis, REAL a) {
  const REAL one = (REAL)1;
  switch (basis) {
    case 0: return R_EXP(-one * (a * a));                         /* gaussian :35-37 */
    case 1: return R_EXP((REAL)-0.1 * (a * a));                   /* gaussian_wide :40-42 */
    case 2: return R_EXP((REAL)-0.01 * (a * a));                  /* gaussian_wider :45-47 */
    case 3: return one / (one + a * a);                           /* inverse_quadratic :50-52 */
    case 4: return a;                                             /* linear :55-57 */
    case 5: return a * a;                                         /* quadratic :61-63 */
    case 6: return R_SQRT(one + a * a);                           /* multiquadric :67-69 */
    case 7: return one / R_SQRT(one + a * a);                     /* inverse_multiquadric :73-75 */
    case 8: return a * a * R_LOG(a + one);                        /* spline :79-81 */
    case 9: return (a - one) * R_EXP(-a);                         /* poisson_one :85-87 */
    case 10: return ((a - 2 * one) / 2 * one) * a * R_EXP(-a);    /* poisson_two :91-97 */
    case 11: return (one + R_SQRT((REAL)3) * a) * R_EXP(-R_SQRT((REAL)3) * a);   /* matern32 */
    case 12: return (one + R_SQRT((REAL)5) * a + ((REAL)5 / (REAL)3) * a * a) *
                    R_EXP(-R_SQRT((REAL)5) * a);                  /* matern52 :107-111 */
    default: return (REAL)0;
  }
}

/* a-3 + a-1 + a-4: src/irbfn_mpc/model.py:42-95, flax_rbf.py:258-285, model.py:169-198.
 * x[B,D]; centers[R,K,D]; log_sigs[R,K]; W[K,O]; bias[O];
 * gate: lo_tab/hi_tab[nsplit*max_ranges] (row d holds the per-split bounds of dim d),
 *       delta[nsplit], dim_ranges[n_ranges*nsplit] (regions >= n_ranges have gamma 0).
 * out[B,O].  OpenMP over queries. */
void FN(oracle_wcrbf_forward)(const REAL* x, const REAL* centers, const REAL* log_sigs,
                              const REAL* W, const REAL* bias, const REAL* lo_tab,
                              const REAL* hi_tab, const REAL* delta, const int* dim_ranges,
                              int n_ranges, int max_ranges, int nsplit, int basis, long B, int D,
                              int R, int K, int O, REAL* out) {
#pragma omp parallel
  {
    REAL* h = (REAL*)malloc(sizeof(REAL) * (size_t)K);
    REAL* gd = (REAL*)malloc(sizeof(REAL) * (size_t)(nsplit > 0 ? nsplit : 1) * (size_t)max_ranges);
#pragma omp for schedule(static)
    for (long b = 0; b < B; ++b) {
      const REAL* xb = x + b * D;
      for (int d = 0; d < nsplit; ++d)               /* model.py:74-86 */
        for (int j = 0; j < max_ranges; ++j) {
          REAL ld = xb[d] - lo_tab[d * max_ranges + j];
          REAL ud = hi_tab[d * max_ranges + j] - xb[d];
          gd[d * max_ranges + j] = ((R_TANH(delta[d] * ld) + 1) / 2) * ((R_TANH(delta[d] * ud) + 1) / 2);
        }
      for (int k = 0; k < K; ++k) h[k] = (REAL)0;
      for (int r = 0; r < R; ++r) {
        REAL gamma = (REAL)0;                        /* model.py:70 */
        if (r < n_ranges) {                          /* model.py:88-93 */
          gamma = gd[0 * max_ranges + dim_ranges[r * nsplit + 0]];
          for (int j = 1; j < nsplit; ++j) gamma = gamma * gd[j * max_ranges + dim_ranges[r * nsplit + j]];
        }
        for (int k = 0; k < K; ++k) {
          const REAL* c = centers + ((size_t)r * K + k) * D;
          REAL acc = (REAL)0;
          for (int j = 0; j < D; ++j) { REAL df = xb[j] - c[j]; acc += df * df; }   /* flax_rbf.py:280 */
          REAL dd = R_SQRT(acc) / R_EXP(log_sigs[(size_t)r * K + k]);
          h[k] += gamma * FN(basis_eval)(basis, dd);                                /* model.py:193 */
        }
      }
      for (int o = 0; o < O; ++o) {                  /* model.py:196 */
        REAL acc = (REAL)0;
        for (int k = 0; k < K; ++k) acc += h[k] * W[(size_t)k * O + o];
        out[b * O + o] = acc + bias[o];
      }
    }
    free(h);
    free(gd);
  }
}

/* a-5: parameter VJP of a-1..a-4 (what jax.value_and_grad returns at scripts/train_nmpc.py:297-298), the C twin
 * of oracle/irbfn_oracle.py::wcrbfnet_vjp for full-size batches: cotangent gout[B,O] ->
 * g_centers[R,K,D], g_log_sigs[R,K], g_W[K,O], g_bias[O].  d^2-only bases (gaussian family, inverse_quadratic,
 * inverse_multiquadric, multiquadric, quadratic); returns -1 otherwise.  OpenMP over centres (a thread owns its
 * k range: no reduction across threads), queries streamed in chunks that stay in cache. */
int FN(oracle_wcrbf_vjp)(const REAL* x, const REAL* gout, const REAL* centers, const REAL* log_sigs, const REAL* W,
                         const REAL* lo_tab, const REAL* hi_tab, const REAL* delta, const int* dim_ranges,
                         int n_ranges, int max_ranges, int nsplit, int basis, long B, int D, int R, int K, int O,
                         REAL* g_centers, REAL* g_log_sigs, REAL* g_W, REAL* g_bias) {
  if (!(basis == 0 || basis == 1 || basis == 2 || basis == 3 || basis == 5 || basis == 6 || basis == 7)) return -1;
  const REAL one = (REAL)1;
  REAL* gamma = (REAL*)malloc(sizeof(REAL) * (size_t)B * (size_t)R);
#pragma omp parallel
  {
    REAL* gd = (REAL*)malloc(sizeof(REAL) * (size_t)(nsplit > 0 ? nsplit : 1) * (size_t)max_ranges);
#pragma omp for schedule(static)
    for (long b = 0; b < B; ++b) {                   /* model.py:42-95 */
      const REAL* xb = x + b * D;
      for (int d = 0; d < nsplit; ++d)
        for (int j = 0; j < max_ranges; ++j) {
          REAL ld = xb[d] - lo_tab[d * max_ranges + j];
          REAL ud = hi_tab[d * max_ranges + j] - xb[d];
          gd[d * max_ranges + j] = ((R_TANH(delta[d] * ld) + 1) / 2) * ((R_TANH(delta[d] * ud) + 1) / 2);
        }
      for (int r = 0; r < R; ++r) {
        REAL gm = (REAL)0;
        if (r < n_ranges) {
          gm = gd[0 * max_ranges + dim_ranges[r * nsplit + 0]];
          for (int j = 1; j < nsplit; ++j) gm = gm * gd[j * max_ranges + dim_ranges[r * nsplit + j]];
        }
        gamma[b * R + r] = gm;
      }
    }
    free(gd);
#pragma omp for schedule(static)
    for (int o = 0; o < O; ++o) {                    /* d bias = sum_b g */
      REAL acc = (REAL)0;
      for (long b = 0; b < B; ++b) acc += gout[b * O + o];
      g_bias[o] = acc;
    }
#pragma omp for schedule(dynamic, 8)
    for (int k = 0; k < K; ++k) {
      for (int o = 0; o < O; ++o) g_W[(size_t)k * O + o] = (REAL)0;
      for (int r = 0; r < R; ++r) {
        g_log_sigs[(size_t)r * K + k] = (REAL)0;
        for (int j = 0; j < D; ++j) g_centers[((size_t)r * K + k) * D + j] = (REAL)0;
      }
      for (long b = 0; b < B; ++b) {
        const REAL* xb = x + b * D;
        const REAL* gb = gout + b * O;
        REAL hbar = (REAL)0;                         /* hbar = g W^T */
        for (int o = 0; o < O; ++o) hbar += gb[o] * W[(size_t)k * O + o];
        REAL hk = (REAL)0;
        for (int r = 0; r < R; ++r) {
          const REAL gm = gamma[b * R + r];
          const REAL* c = centers + ((size_t)r * K + k) * D;
          const REAL s2 = R_EXP((REAL)-2 * log_sigs[(size_t)r * K + k]);
          REAL r2 = (REAL)0;
          for (int j = 0; j < D; ++j) { REAL df = xb[j] - c[j]; r2 += df * df; }
          const REAL d2 = r2 * s2;
          REAL phi, dphi;                            /* phi(d^2) and d phi / d(d^2) */
          switch (basis) {
            case 0: phi = R_EXP(-d2); dphi = -phi; break;
            case 1: phi = R_EXP((REAL)-0.1 * d2); dphi = (REAL)-0.1 * phi; break;
            case 2: phi = R_EXP((REAL)-0.01 * d2); dphi = (REAL)-0.01 * phi; break;
            case 3: phi = one / (one + d2); dphi = -(phi * phi); break;
            case 5: phi = d2; dphi = one; break;
            case 6: phi = R_SQRT(one + d2); dphi = (REAL)0.5 / phi; break;
            default: phi = one / R_SQRT(one + d2); dphi = (REAL)-0.5 * phi * phi * phi; break;
          }
          hk += gm * phi;
          const REAL t = hbar * gm * dphi;
          g_log_sigs[(size_t)r * K + k] += t * ((REAL)-2 * d2);
          const REAL cf = t * s2 * (REAL)-2;
          for (int j = 0; j < D; ++j) g_centers[((size_t)r * K + k) * D + j] += cf * (xb[j] - c[j]);
        }
        for (int o = 0; o < O; ++o) g_W[(size_t)k * O + o] += hk * gb[o];   /* d W = h^T g */
      }
    }
  }
  free(gamma);
  return 0;
}

/* a-7: src/irbfn_mpc/dynamics.py:9-91 ; kinematic_only != 0 gives a-8 (:103-187) per step. */
static inline void FN(st_step)(REAL* s, REAL accl_in, REAL sv_in, const REAL* p, int kinematic_only) {
  const REAL g = (REAL)9.81;
  REAL mu = p[0], m = p[1], I = p[2], lf = p[3], lr = p[4], C_Sf = p[5], C_Sr = p[6], h = p[7],
       dt = p[8], sv_max = p[9], a_max = p[10], s_max = p[11], v_max = p[12];
  REAL DELTA = FN(clipr)(s[2], -s_max, s_max);       /* :40 */
  REAL V = FN(clipr)(s[3], -v_max, v_max);           /* :41 */
  REAL PSI = s[4], PSI_DOT = s[5], BETA = s[6];
  REAL ACCL = FN(clipr)(accl_in, -a_max, a_max);     /* :46 */
  REAL SV = FN(clipr)(sv_in, -sv_max, sv_max);       /* :47 */
  REAL f[7];
  if (!kinematic_only && V > (REAL)3.0) {            /* :90 */
    f[0] = V * R_COS(PSI + BETA);
    f[1] = V * R_SIN(PSI + BETA);
    f[2] = SV;
    f[3] = ACCL;
    f[4] = PSI_DOT;
    f[5] = ((mu * m) / (I * (lf + lr))) *
           (lf * C_Sf * (g * lr - ACCL * h) * DELTA +
            (lr * C_Sr * (g * lf + ACCL * h) - lf * C_Sf * (g * lr - ACCL * h)) * BETA -
            (lf * lf * C_Sf * (g * lr - ACCL * h) + lr * lr * C_Sr * (g * lf + ACCL * h)) *
                (PSI_DOT / V));
    f[6] = (mu / (V * (lr + lf))) *
               (C_Sf * (g * lr - ACCL * h) * DELTA -
                (C_Sr * (g * lf + ACCL * h) + C_Sf * (g * lr - ACCL * h)) * BETA +
                (C_Sr * (g * lf + ACCL * h) * lr - C_Sf * (g * lr - ACCL * h) * lf) * (PSI_DOT / V)) -
           PSI_DOT;
  } else {
    f[0] = V * R_COS(PSI);                           /* :80 */
    f[1] = V * R_SIN(PSI);
    f[2] = SV;
    f[3] = ACCL;
    f[4] = (V / (lr + lf)) * R_TAN(DELTA);           /* :84 */
    f[5] = (REAL)0;
    f[6] = (REAL)0;
  }
  for (int i = 0; i < 7; ++i) s[i] = s[i] + f[i] * dt;
}

/* x_and_pred_u[B,7+2T] -> states[B,T,7]; u = [a_0..a_{T-1}, sv_0..sv_{T-1}] (dynamics.py:98) */
void FN(oracle_integrate_st_mult)(const REAL* xu, const REAL* p, long B, int T, int kinematic_only,
                                  REAL* states) {
#pragma omp parallel for schedule(static)
  for (long b = 0; b < B; ++b) {
    const REAL* row = xu + b * (7 + 2 * T);
    REAL s[7];
    for (int i = 0; i < 7; ++i) s[i] = row[i];
    for (int t = 0; t < T; ++t) {
      FN(st_step)(s, row[7 + t], row[7 + T + t], p, kinematic_only);
      for (int i = 0; i < 7; ++i) states[(b * T + t) * 7 + i] = s[i];
    }
  }
}

/* a-9: src/irbfn_mpc/dynamics.py:190-290 (low-speed RHS only, :267-280) */
void FN(oracle_integrate_frenet_mult)(const REAL* xu, const REAL* p, long B, int T, REAL* states) {
  REAL LF = p[3], LR = p[4], dt = p[8], sv_max = p[9], a_max = p[10], s_max = p[11];
#pragma omp parallel for schedule(static)
  for (long b = 0; b < B; ++b) {
    const REAL* row = xu + b * (8 + 2 * T);
    REAL s[8];
    for (int i = 0; i < 8; ++i) s[i] = row[i];
    for (int t = 0; t < T; ++t) {
      REAL ey = s[1], delta = FN(clipr)(s[2], -s_max, s_max), vx = s[3], epsi = s[6], cur = s[7];
      REAL a = FN(clipr)(row[8 + t], -a_max, a_max);
      REAL dv = FN(clipr)(row[8 + T + t], -sv_max, sv_max);
      REAL d[8];
      d[0] = (vx * R_COS(epsi)) / (1 - ey * cur);
      d[1] = vx * R_SIN(epsi);
      d[2] = dv;
      d[3] = a;
      d[4] = 0;
      d[5] = 0;
      d[6] = (vx * R_TAN(delta)) / (LR + LF) - cur * ((vx * R_COS(epsi)) / (1 - cur * ey));
      d[7] = 0;
      for (int i = 0; i < 8; ++i) s[i] = s[i] + d[i] * dt;
      for (int i = 0; i < 8; ++i) states[(b * T + t) * 8 + i] = s[i];
    }
  }
}

/* a-6: scripts/train_nmpc.py:306-374 */
void FN(oracle_rollout_fullint)(const REAL* v0, const REAL* u, long B, int T, REAL* states) {
  const REAL DT = (REAL)0.1, WB = (REAL)0.33, VMAX = (REAL)7.0, VMIN = (REAL)0.0, SMAX = (REAL)0.4189;
#pragma omp parallel for schedule(static)
  for (long b = 0; b < B; ++b) {
    REAL x = 0, y = 0, delta = 0, v = FN(clipr)(v0[b], VMIN, VMAX), yaw = 0;
    for (int i = 0; i < T; ++i) {
      REAL a = u[b * 2 * T + i], dv = u[b * 2 * T + T + i];
      x = x + v * R_COS(yaw) * DT;
      y = y + v * R_SIN(yaw) * DT;
      delta = FN(clipr)(delta + dv * DT, -SMAX, SMAX);
      v = FN(clipr)(v + a * DT, VMIN, VMAX);
      yaw = yaw + (v / WB) * R_TAN(delta) * DT;
      REAL* o = states + (b * T + i) * 5;
      o[0] = x; o[1] = y; o[2] = delta; o[3] = v; o[4] = yaw;
    }
  }
}

/* a-10: src/irbfn_mpc/planner_utils.py:8-77 */
void FN(oracle_integrate_path_mult)(const REAL* params, long B, int N, REAL* states) {
  static const double PM[4][4] = {{1.0, 0.0, 0.0, 0.0},
                                  {-11.0 / 2, 9.0, -9.0 / 2, 1.0},
                                  {9.0, -45.0 / 2, 18.0, -9.0 / 2},
                                  {-9.0 / 2, 27.0 / 2, -27.0 / 2, 9.0 / 2}};
#pragma omp parallel for schedule(static)
  for (long b = 0; b < B; ++b) {
    const REAL* q = params + b * 5;
    REAL s = q[4], coefs[4];
    for (int r = 0; r < 4; ++r) {
      REAL acc = (REAL)PM[r][0] * q[0];
      for (int j = 1; j < 4; ++j) acc = acc + (REAL)PM[r][j] * q[j];
      coefs[r] = acc;
    }
    coefs[1] = coefs[1] / s;
    coefs[2] = coefs[2] / (s * s);
    coefs[3] = coefs[3] / (s * s * s);
    REAL st[6] = {0, 0, 0, coefs[0], 0, 0};
    for (int i = 0; i < N; ++i) {
      REAL sk = (i < N - 1) ? s * (REAL)((double)i / (double)(N - 1)) : s;
      REAL k = (REAL)(i + 1);
      REAL kap = 0, th = 0, pw = 1;
      for (int j = 0; j < 4; ++j) {
        REAL temp = coefs[j] * pw;
        kap = kap + temp;
        th = th + temp * sk / (REAL)(j + 1);
        pw = pw * sk;
      }
      REAL dx = st[4] * (1 - 1 / k) + (R_COS(th) + R_COS(st[2])) / 2 / k;
      REAL dy = st[5] * (1 - 1 / k) + (R_SIN(th) + R_SIN(st[2])) / 2 / k;
      st[0] = sk * dx; st[1] = sk * dy; st[2] = th; st[3] = kap; st[4] = dx; st[5] = dy;
      for (int j = 0; j < 6; ++j) states[(b * N + i) * 6 + j] = st[j];
    }
  }
}

#undef R_SIN
#undef R_COS
#undef R_TAN
#undef R_TANH
#undef R_EXP
#undef R_SQRT
#undef R_LOG
#undef FN
#undef CAT
#undef CAT_
