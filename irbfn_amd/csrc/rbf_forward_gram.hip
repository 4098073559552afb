// K1g: the narrow forward (one region, O <= 16, fast bases) with BOTH of its GEMM-shaped pieces on the f16 matrix cores:
// the squared distances as a Gram expansion and Phi x W (the hi/lo pairs of K1h).  Same mathematics as K1 / K1h
// (src/irbfn_mpc/model.py:169-198; RBF stage flax_rbf.py:258-285).
//
// Why: K1h is VALU-issue bound, and 14 of its ~21.5 VALU instructions per (query, centre) pair are the distance
// (x_i - c_i, fma) x 7.  With u = alpha_k |x - c_k|^2 + beta (the argument of the transcendental) written as
//     u = alpha_k Q  +  sum_i (-2 alpha_k c'_ki) x'_i  +  (alpha_k |c'_k|^2 + beta),      x' = x - r, c' = c - r, Q = |x'|^2
// it is a [centres x slots] x [slots x queries] product whose D layout (v_mfma: lane = query column, registers = 4 centre
// rows) is already the A layout of the Phi x W product: no shuffle between the two.  What is left on the VALU per pair is the
// transcendental and the hi/lo split of its result.
//
// Accuracy -- the expansion cancels (terms of size M = alpha (|x'|^2 + |c'|^2) sum to u <~ 30), and the matrix core adds its
// products in a tree of truncating ~24-bit adders (tools/probe_mfma_accum.hip, profiles/r03_mfma_accumulation_probe.txt), so a
// naive expansion carries an error of 2^-24 M.  Here the cancellation is EXACT:
//  * every factor v in {x'_i, Q, C_ki = -2 alpha_k c'_ki, alpha_k, c2_k} with |v| < 2^E is cut into a FIXED-POINT head
//    n0 = rint(v 2^(10-E)) 2^-10 (11 bits: exact in f16) and two float tails n1, n2 (f16 roundings of the scaled residuals:
//    v = 2^E (n0 + 2^-11 n1 + 2^-22 n2) to 2^(E-34)); c2 has two fixed-point heads;
//  * the head x head products of one (query, centre) element are integer multiples of one grid 2^(EX+EC-20) and their partial
//    sums stay below 2^24 grid units (checked at pack time): whatever the order of the adder tree, nothing is truncated --
//    the first MFMA (16x16x16, 10 of its 16 k-slots) returns the cancelled head sum S1 exactly;
//  * the 42 tail products (each < 2^-10 of a head product) are added to S1 by two 16x16x32 MFMAs: truncation there is
//    2^-24 of max(|u|, 2^-10 M).
// So u carries the error of a float32 evaluation of the direct form (a few 2^-24 |u|), not 2^-24 M.  Exponents EX .. E2 are
// chosen per net from the centres (pack_all.hip: statistics role, header per block); a query outside the representable box (|x'_i| >= 2^EX, non-finite)
// makes its WAVE take the VALU distances of K1h for its 32 queries (same records, same f16_arg): results there are K1h's.
// A net whose exponents do not fit (ok = 0 in the header, read back once by irbfn_net_set_params) is not dispatched here.
//
// Layout: v_mfma_f32_16x16x16_f16 / 16x16x32_f16 with rows = 16 centres, k = slots, columns = 16 queries.  A lane owns
// query (l & 15) of its wave's two query tiles; the D registers of centre tile ct hold centres 16 ct + 4 g + r (g = l >> 4),
// which become k-slots 8 g + 4 ct + r of the Phi x W product (the W rows of the image are permuted to match).
// Chunk image (32 centres, 7 KiB): head operands [ct][lane] 8 B, tail operands [ct][half][lane] 16 B, W hi, W lo (1 KiB each);
// the QG waves of a centre slice share an LDS ring of five images filled by LDS-DMA, one barrier per two chunks.
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rbf_forward_gram.h"

namespace irbfn {

// ---- kernel ----------------------------------------------------------------------------------------------
template <int DC, int BC, bool ROLL>
__device__ __forceinline__ void gram_body(const GramArgs& ga, const F16Roll& rl, int mode, unsigned char* lds) {
  static_assert(DC <= kGramDims, "eight coordinate slots");
  const F16Args& a = ga.f;
  constexpr int CBL = f16_chunk_bytes(DC);                   // K1h's chunk image (the VALU path reads its records)
  constexpr int CB = kGramChunkBytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = a.S, QG = a.QG;
  const int slice = wave / QG, qg = wave % QG;               // the QG waves of a slice are adjacent and share its ring
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * 32;
  const GramHdr* hp = ga.hdr;
  long qrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    qrow[t] = q < 0 ? 0 : q;
  }
  // ---- query-side operands: B[k = slot][column = query]
  h4_t bhd[2];
  h8_t btl[2][2];
  const bool bad = gram_query_operands<DC>(a, hp, qrow, g, bhd, btl);
  const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;     // wave-uniform: the VALU distances for these 32 queries

  const int c0 = (int)((long)a.nchunks * slice / S), c1 = (int)((long)a.nchunks * (slice + 1) / S);
  const int na = c1 - c0;
  int nsteps = 0;                                            // every wave of the block walks the longest slice (barriers)
  for (int s2 = 0; s2 < S; ++s2) {
    const int m = (int)((long)a.nchunks * (s2 + 1) / S) - (int)((long)a.nchunks * s2 / S);
    nsteps = m > nsteps ? m : nsteps;
  }
  // Ring of kGramRing chunk images per slice: during step i the waves read the distance operands of chunk i + 1 and the W
  // operands of chunk i while later chunks land (end_of_step below).
  unsigned char* ring = lds + (size_t)slice * kGramRing * CB;
  constexpr int NVI = CB / 1024;                             // 7 wave-instructions per chunk image
  auto stage = [&](int k, int buf) {                         // chunk c0 + k of the slice -> ring slot buf; the QG waves share the copy
    if (k >= na) return;
    const unsigned char* gp = ga.gimg + (size_t)(c0 + k) * CB + lane * 16;
    unsigned char* dst = ring + buf * CB;
    for (int v = qg; v < NVI; v += QG)
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto next3 = [&](int b3) { return b3 == kGramRing - 1 ? 0 : b3 + 1; };
  // A step touches chunks i (W) and i + 1 (distance operands).  One barrier per kGramPer = (kGramRing - 1) / 2 steps: behind it chunks
  // i + 1 .. i + kGramPer + 1 are resident and the next kGramPer are requested into the slots of the chunks everybody has left
  // (kGramRing = 3: a barrier per chunk; 5: one per two chunks).
  constexpr int kGramPer = (kGramRing - 1) / 2;
  static_assert(kGramRing == 2 * kGramPer + 1 && kGramPer >= 1, "ring = 2 x (steps per barrier) + 1");
  auto end_of_step = [&](int i, int b0) {
    if ((i % kGramPer) != kGramPer - 1) return;
    step_barrier();
#pragma unroll
    for (int j = 0; j < kGramPer; ++j) {
      int slot = b0 - (kGramPer - 1) + j;                    // slot of chunk i - (kGramPer - 1) + j
      slot = slot < 0 ? slot + kGramRing : slot;
      stage(i + kGramPer + 2 + j, slot);
    }
  };

  f4_t acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A1: ph * wh
  f4_t acl[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};                // A2: pls * wh + ph * wls
  auto distances = [&](const unsigned char* buf, f4_t (&u)[2][2]) { gram_distances(buf, lane, bhd, btl, u); };
  // transcendental, hi / lo split and Phi x W of the 16 pairs in t16 with the W operands of chunk `buf`
  auto products = [&](float (&t16)[16], const unsigned char* buf, auto pre) {
    const h8_t bh = *reinterpret_cast<const h8_t*>(buf + kGramOpBytes + lane * 16);
    const h8_t bl = *reinterpret_cast<const h8_t*>(buf + kGramOpBytes + kF16WBytes + lane * 16);
    pre(t16);                                                // P = 2^kPhiExp * phi for the step's 16 pairs
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned wh[4], wl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) split_pair_mix(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
      const h8_t ah = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
      const h8_t al = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
      acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acl[t], 0, 0, 0);
      acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acl[t], 0, 0, 0);
    }
  };

#pragma unroll
  for (int k = 0; k <= kGramPer; ++k) stage(k, k);
  step_barrier();                                            // the first kGramPer + 1 chunks are there
#pragma unroll
  for (int k = kGramPer + 1; k < kGramRing; ++k) stage(k, k);
  if (!wave_bad) {
    // two steps per trip: the distances of chunk i + 1 are issued in front of the VALU work on chunk i
    f4_t ua[2][2], ub[2][2];
    if (na > 0) distances(ring, ua);
    // trans16 is inline asm: the hazard recogniser does not see it read MFMA results.  In the loop the 12 distance MFMAs of the
    // next chunk lie in front of it; a slice of ONE chunk reads them right away: explicit wait states, once
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    int b0 = 0;                                              // ring slot of chunk i
    [[maybe_unused]] unsigned long long tph[5] = {0, 0, 0, 0, 0};
    auto one_step = [&](int i, f4_t (&ucur)[2][2], f4_t (&unxt)[2][2]) {
      const int b1 = next3(b0);
      [[maybe_unused]] const unsigned long long t0 = IRBFN_GRAM_T();
      if (i + 1 < na) distances(ring + b1 * CB, unxt);       // issued in front of the VALU work on chunk i
      if (i < na) {
        float t16[16];
        products(t16, ring + b0 * CB, [&](float (&o)[16]) { trans16<BC>(ucur, o); });
      }
      [[maybe_unused]] const unsigned long long t2 = IRBFN_GRAM_T();
      end_of_step(i, b0);
      [[maybe_unused]] const unsigned long long t4 = IRBFN_GRAM_T();
#ifdef IRBFN_GRAM_STAMPS
      tph[1] += t2 - t0; tph[2] += t4 - t2; tph[4] += 1;
#endif
      b0 = b1;
    };
    for (int i = 0; i < nsteps; i += 2) {
      one_step(i, ua, ub);
      if (i + 1 < nsteps) one_step(i + 1, ub, ua);
    }
#ifdef IRBFN_GRAM_STAMPS
    if (blockIdx.x < 2 && tid == 0)
      for (int k = 0; k < 5; ++k) g_gram_stamps[blockIdx.x * 8 + k] = tph[k];
#endif
  } else {
    // a query of this wave lies outside the representable box (or is not finite): K1h's distances for its 32 queries,
    // same barriers and the same share of the copies
    int b0 = 0;
    for (int i = 0; i < nsteps; ++i) {
      if (i < na) {
        float t16[16];
        gram_valu_args<DC, BC>(a, qrow, g, reinterpret_cast<const float*>(a.img + (size_t)(c0 + i) * CBL), t16);
        products(t16, ring + b0 * CB, [&](float (&o)[16]) { trans_block<BC, 16>(o); });
      }
      end_of_step(i, b0);
      b0 = next3(b0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], kLoScale, acc[t][r]);   // A1 + 2^-11 A2

  // ---- smooth region gate of the single region (model.py:42-95), one value per query
  const GateTables gt = a.gate;
  float gam[2] = {0.0f, 0.0f};
  if (slice == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float gv = gt.n_ranges > 0 ? 1.0f : 0.0f;              // model.py:70
#pragma unroll
      for (int d = 0; d < DC; ++d)
        if (d < gt.nsplit && gt.n_ranges > 0) {
          const int e = d * gt.max_ranges + gt.dim_ranges[d];
          gv *= gate_factor(a.x[qrow[t] * a.Dreal + d], gt.lo[e], gt.hi[e], gt.delta[d]);
        }
      gam[t] = gv;
    }
  }
  narrow_epilogue<ROLL>(a, rl, mode, lds, acc, gam, S, slice, qg, q0, 1.0f / (gram_phi_scale<BC>() * kWScale));
}

template <int DC, int BC>
__global__ __launch_bounds__(1024, IRBFN_GRAM_WAVES) void rbf_fwd_f16gram(const GramArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  gram_body<DC, BC, false>(ga, F16Roll{}, -1, lds);
}

// the planning tick of a narrow net in one launch (forward + sign flip + roll-out; irbfn_planner.py:203-212)
template <int DC, int BC>
__global__ __launch_bounds__(1024, IRBFN_GRAM_WAVES) void rbf_tick_f16gram(const GramArgs ga, const F16Roll rl, const int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  gram_body<DC, BC, true>(ga, rl, mode, lds);
}

// ---- host side -------------------------------------------------------------------------------------------
static int gram_nt(const irbfn_net* net) { return (net->O + 15) / 16; }

bool gram_eligible(const irbfn_net* net) {
  return f16_eligible(net) && net->DC <= kGramDims;
}

size_t gram_image_bytes(const irbfn_net* net) {
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  return (size_t)nchunks * gram_chunk_bytes(gram_nt(net));
}

size_t gram_header_bytes() { return sizeof(GramHdr); }

#ifdef IRBFN_GRAM_STAMPS
extern "C" int irbfn_debug_gram_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_gram_stamps), sizeof(unsigned long long) * 32);
}
#endif

template <int DC>
static int launch_gram_bc(const GramArgs& a, int bc, int grid, int block, size_t lds, hipStream_t s) {
#define IRBFN_GCASE(BCV)                                                                                      \
  case BCV: {                                                                                                 \
    auto k = rbf_fwd_f16gram<DC, BCV>;                                                                        \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a);                                                \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_GCASE(BC_GAUSS)
    IRBFN_GCASE(BC_IQ)
    IRBFN_GCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_GCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// S centre slices x QG query groups of 32 per block (S * QG <= 8 waves)
int launch_forward_gram(irbfn_net* net, const float* x, float* out, int64_t B, int S, int QG, hipStream_t s) {
  if (!net->gram_img || !net->f16_img || !gram_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  if (net->O > 16) return launch_forward_gram_wide(net, x, out, B, S, QG, s);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  if (S < 1 || QG < 1 || S * QG > 16 || S > nchunks) return IRBFN_ERR_BAD_ARG;
  GramArgs a;
  a.f.x = x; a.f.img = net->f16_img; a.f.oscale = net->f16_oscale; a.f.bias = net->bias; a.f.out = out; a.f.gate = net->gate();
  a.f.B = (long)B; a.f.Dreal = net->D; a.f.O = net->O; a.f.nchunks = nchunks; a.f.S = S; a.f.QG = QG;
  a.gimg = net->gram_img;
  a.hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
  const int waves = S * QG;
  const size_t ring = (size_t)S * kGramRing * kGramChunkBytes;
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32) * sizeof(float);
  const size_t lds = ring > red ? ring : red;
  if (lds > 160 * 1024) return IRBFN_ERR_UNSUPPORTED;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  int rc;
  switch (net->DC) {
    case 3: rc = launch_gram_bc<3>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 4: rc = launch_gram_bc<4>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 7: rc = launch_gram_bc<7>(a, net->bclass, grid, waves * 64, lds, s); break;
    case 8: rc = launch_gram_bc<8>(a, net->bclass, grid, waves * 64, lds, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_f16gram<D=%d,BC=%d,S=%d,QG=%d>", net->DC, net->bclass, S, QG);
    net->last_grid = grid;
    net->last_block = waves * 64;
  }
  return rc;
}

template <int DC>
static int launch_tick_gram_bc(const GramArgs& a, const F16Roll& rl, int mode, int bc, int grid, int block, size_t lds, hipStream_t s) {
#define IRBFN_GCASE(BCV)                                                                                      \
  case BCV: {                                                                                                 \
    auto k = rbf_tick_f16gram<DC, BCV>;                                                                       \
    if (lds > 48 * 1024) {                                                                                    \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
      if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }                               \
    }                                                                                                         \
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, s, a, rl, mode);                                      \
    break;                                                                                                    \
  }
  switch (bc) {
    IRBFN_GCASE(BC_GAUSS)
    IRBFN_GCASE(BC_IQ)
    IRBFN_GCASE(BC_IMQ)
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_GCASE
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

// The one-launch planning tick of a narrow net on K1g (rbf_tick_f16gram): same conditions as K1h's (launch_tick_f16_narrow),
// and the parameters fit the expansion.  IRBFN_ERR_UNSUPPORTED: no instance -> the caller takes another path
int launch_tick_gram_narrow(irbfn_net* net, int mode, const float* x, const int* mirror, const float* state0, const DynParams& dp,
                            float* controls, float* states, int64_t B, int T, hipStream_t s) {
  if (net->opt[IRBFN_OPT_FWD_KERNEL] != IRBFN_FWD_AUTO && net->opt[IRBFN_OPT_FWD_KERNEL] != IRBFN_FWD_K1G) return IRBFN_ERR_UNSUPPORTED;
  if (!gram_preferred(net, B)) return IRBFN_ERR_UNSUPPORTED;
  if (net->opt[IRBFN_OPT_TICK_FUSED] == 0 || net->O != 2 * T || T > kTickNarrowT) return IRBFN_ERR_UNSUPPORTED;
  const bool st = mode == IRBFN_ROLLOUT_ST_SELECT || mode == IRBFN_ROLLOUT_ST_KS || mode == IRBFN_ROLLOUT_FULLINT;
  if (!((st && net->DC == 7) || (mode == IRBFN_ROLLOUT_FRENET_LS && net->DC == 8))) return IRBFN_ERR_UNSUPPORTED;
  int S, QG;
  gram_geometry(net, B, &S, &QG);
  const int nchunks = (net->N + kF16Chunk - 1) / kF16Chunk;
  const int waves = S * QG;
  const size_t ring = (size_t)S * kGramRing * kGramChunkBytes;
  const size_t red = ((size_t)waves * 2 * 4 * 64 + (size_t)QG * 32 + (size_t)QG * 32 * (kTickNarrowCP + kTickNarrowSP)) * sizeof(float);
  const size_t lds = ring > red ? ring : red;
  if (lds > 160 * 1024 || S > nchunks) return IRBFN_ERR_UNSUPPORTED;
  GramArgs a;
  a.f.x = x; a.f.img = net->f16_img; a.f.oscale = net->f16_oscale; a.f.bias = net->bias; a.f.out = controls; a.f.gate = net->gate();
  a.f.B = (long)B; a.f.Dreal = net->D; a.f.O = net->O; a.f.nchunks = nchunks; a.f.S = S; a.f.QG = QG;
  a.gimg = net->gram_img;
  a.hdr = reinterpret_cast<const GramHdr*>(net->gram_hdr);
  F16Roll rl;
  rl.state0 = state0; rl.states = states; rl.mirror = mirror; rl.T = T; rl.wlds = 0; rl.dp = dp;
  const long groups = (B + 31) / 32;
  const int grid = (int)((groups + QG - 1) / QG);
  const int rc = net->DC == 7 ? launch_tick_gram_bc<7>(a, rl, mode, net->bclass, grid, waves * 64, lds, s)
                              : launch_tick_gram_bc<8>(a, rl, mode, net->bclass, grid, waves * 64, lds, s);
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_tick_f16gram<D=%d,BC=%d,MODE=%d,S=%d,QG=%d>", net->DC, net->bclass, mode, S, QG);
    net->last_grid = grid;
    net->last_block = waves * 64;
  }
  return rc;
}

}  // namespace irbfn
