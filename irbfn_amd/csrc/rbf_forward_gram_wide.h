// K1g for wide outputs (16 < O <= 128): the body of rbf_fwd_f16gram_wide (rbf_forward_gram_wide.hip) and of the one-launch
// planning tick rbf_tick_f16gram_wide (plan_tick_wide.hip).  The squared distances of a 32-centre chunk come from the matrix
// cores as the exactly-cancelling Gram expansion of rbf_forward_gram.hip (12 MFMAs for the wave's 32 queries instead of
// 16 x 15 VALU instructions), the transcendental and the (hi, lo) split stay on the VALU, and the 6 NT MFMAs of Phi x W follow as
// in K1h's wide kernel (rbf_forward_f16_wide.h) -- whose epilogue (slice sums per column tile, gate, scale, bias, optional
// roll-out of the block's trajectories) is shared.
//
// Chunk image: 5 KiB of distance operands (rbf_forward_gram.h) + NT x (W hi, W lo) with the W rows in the k order the distance
// MFMAs leave the basis values in (centre 16 ct + 4 g + r <-> k = 8 g + 4 ct + r).  The QG waves of a centre slice share an
// LDS ring of three images filled by LDS-DMA; during step i a wave turns the arguments of chunk i into (hi, lo) basis operands,
// issues the distance MFMAs of chunk i + 1 and, behind them, the Phi x W products of chunk i; one barrier per step.
#pragma once

#include "rbf_forward_gram.h"

namespace irbfn {

template <int DC, int BC, int NT, int MODE>
__device__ __forceinline__ void wide_gram_body(const GramArgs& ga, const F16Roll& rl, unsigned char* lds) {
  const F16Args& a = ga.f;
  constexpr int CBL = f16_chunk_bytes(DC, NT);               // K1h's chunk image (the VALU path reads its records)
  constexpr int CB = gram_chunk_bytes(NT);
  constexpr int NVI = CB / 1024;                             // wave-instructions per chunk image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int SW = a.S, QG = a.QG;
  const int slice = wave / QG, qg = wave % QG;
  const int g = lane >> 4, n = lane & 15;
  const long q0 = ((long)blockIdx.x * QG + qg) * 32;
  long qrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    long q = q0 + t * 16 + n;
    q = q < a.B ? q : a.B - 1;
    qrow[t] = q < 0 ? 0 : q;
  }
  h4_t bhd[2];
  h8_t btl[2][2];
  const bool bad = gram_query_operands<DC>(a, ga.hdr, qrow, g, bhd, btl);
  const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;     // wave-uniform: the VALU distances for these 32 queries

  const int nsteps = (a.nchunks + SW - 1) / SW;              // chunks per slice (the last slice may have fewer)
  const int c0 = slice * nsteps;
  const int c1 = (c0 + nsteps) < a.nchunks ? (c0 + nsteps) : a.nchunks;
  const int na = c1 > c0 ? c1 - c0 : 0;
  unsigned char* ring = lds + (size_t)slice * 3 * CB;
  auto stage = [&](int k, int buf) {                         // chunk c0 + k of the slice -> ring slot buf; the QG waves share the copy
    if (k >= na) return;
    const unsigned char* gp = ga.gimg + (size_t)(c0 + k) * CB + lane * 16;
    unsigned char* dst = ring + buf * CB;
    for (int v = qg; v < NVI; v += QG)
      __builtin_amdgcn_global_load_lds((gptr_t)(gp + v * 1024), (lptr_t)(dst + v * 1024), 16, 0, 0);
  };
  auto step_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto next3 = [&](int b3) { return b3 == 2 ? 0 : b3 + 1; };

  f4_t acc[2][NT], acl[2][NT];                               // A1, A2 (f16_split.h)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) { acc[t][ct] = f4_t{0, 0, 0, 0}; acl[t][ct] = f4_t{0, 0, 0, 0}; }
  // (hi, lo) A operands of the Phi x W products from the 16 basis values in t16
  auto split16 = [&](const float (&t16)[16], h8_t (&ah)[2], h8_t (&al)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned wh[4], wl[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) split_pair_mix(t16[t * 8 + 2 * jj], t16[t * 8 + 2 * jj + 1], wh[jj], wl[jj]);
      ah[t] = __builtin_bit_cast(h8_t, u4_t{wh[0], wh[1], wh[2], wh[3]});
      al[t] = __builtin_bit_cast(h8_t, u4_t{wl[0], wl[1], wl[2], wl[3]});
    }
  };
  // the 6 NT MFMAs of Phi x W with the W operands of the chunk image at `buf`; the W operands of column tile ct + 1 are read
  // while the MFMAs of tile ct run
  auto products = [&](const h8_t (&ah)[2], const h8_t (&al)[2], const unsigned char* buf) {
    const unsigned char* wp = buf + kGramOpBytes + lane * 16;
    h8_t bh = *reinterpret_cast<const h8_t*>(wp);
    h8_t bl = *reinterpret_cast<const h8_t*>(wp + kF16WBytes);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
      h8_t nbh = bh, nbl = bl;
      if (ct + 1 < NT) {
        nbh = *reinterpret_cast<const h8_t*>(wp + (ct + 1) * 2 * kF16WBytes);
        nbl = *reinterpret_cast<const h8_t*>(wp + (ct + 1) * 2 * kF16WBytes + kF16WBytes);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bh, acc[t][ct], 0, 0, 0);
        acl[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], bh, acl[t][ct], 0, 0, 0);
        acl[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], bl, acl[t][ct], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      bh = nbh; bl = nbl;
    }
  };

  stage(0, 0);
  stage(1, 1);
  step_barrier();                                            // chunks 0 and 1 are there
  stage(2, 2);
  if (!wave_bad) {
    // per step: basis values and operand split of chunk i (VALU), then the distance MFMAs of chunk i + 1 -- issued once the
    // arguments of chunk i are consumed: one set of 16 argument registers -- and the Phi x W MFMAs of chunk i behind them
    f4_t u[2][2];
    if (na > 0) gram_distances(ring, lane, bhd, btl, u);
    // trans16 is inline asm: the hazard recogniser does not see it read MFMA results.  In the loop 6 NT MFMAs lie between the
    // distance MFMAs and the transcendentals that read them; here, once, explicit wait states do
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    int b0 = 0;                                              // ring slot of chunk i
    for (int i = 0; i < nsteps; ++i) {
      const int b1 = next3(b0);
      if (i < na) {
        h8_t ah[2], al[2];
        {
          float t16[16];
          trans16<BC>(u, t16);
          split16(t16, ah, al);
        }
        if (i + 1 < na) gram_distances(ring + b1 * CB, lane, bhd, btl, u);
        products(ah, al, ring + b0 * CB);
      }
      step_barrier();                                        // chunk i + 2 is there; everybody has left chunk i
      stage(i + 3, b0);
      b0 = b1;
    }
  } else {
    int b0 = 0;
    for (int i = 0; i < nsteps; ++i) {
      if (i < na) {
        h8_t ah[2], al[2];
        {
          float t16[16];
          gram_valu_args<DC, BC>(a, qrow, g, reinterpret_cast<const float*>(a.img + (size_t)(c0 + i) * CBL), t16);
          trans_block<BC, 16>(t16);
          split16(t16, ah, al);
        }
        products(ah, al, ring + b0 * CB);
      }
      step_barrier();
      stage(i + 3, b0);
      b0 = next3(b0);
    }
  }
  wide_epilogue<DC, NT, MODE>(a, rl, lds, acc, acl, slice, qg, q0, 1.0f / (gram_phi_scale<BC>() * kWScale));
}

}  // namespace irbfn
