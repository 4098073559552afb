// K1r / K2r: REGION-SPARSE evaluation of multi-region nets (R > 1) for gfx950.
//
// The reference evaluates WCRBFNet densely: every query meets all R regions and the products with a zero region
// weight are computed and thrown away (src/irbfn_mpc/model.py:187-193: gamma_rep * all_x over [B,R,K]).  The smooth
// indicator gamma (model.py:42-95) is a product of tanh sigmoids that is EXACTLY 0 in float32 outside a margin of
// 9.01 / delta around a region's box (rbf_forward.h, half_tanh_plus_one), so on the reference's own planners a query
// has gamma != 0 in 4-6 of 128 (or 12) regions.  Skipping gamma == 0 changes no result (0 * phi, phi finite).
//
// K1's skip is per WAVE (a region is skipped when all 64 queries of a wave have gamma == 0), which never fires on a
// batch in arrival order.  Here a query's live regions are found per LANE and the (query, region) pairs are the unit of work:
//   1. the net sits in LDS as ONE image (sp_img_layout: region masks and index words, factor entries, Dense weight rows,
//      centre table {c[D], folded width} -- 43 KB for 128 regions x 10 centres), copied by LDS-DMA; everything a lane later
//      reads by a wave-uniform address is a broadcast LDS read, never a scalar load the wave would stall on;
//   2. a lane evaluates the per-(dimension, range) factors of its query once (E <= 32 of them, LDS column) and keeps
//      "factor != 0" as a bit mask M; region r is live iff (M & req[r]) == req[r] (req[r] = the factor bits region r
//      multiplies): 32 regions at a time become one bit word without a branch;
//   3. the (query, live region) pairs of a workgroup are numbered query-major (prefix sum of the popcounts), written out
//      as flat arrays (region, owner) and dealt to the lanes round by round, one pair per lane and round: the list lengths
//      are very uneven (mean 3.6, max 32 on the 128-region planner -- a corner of the gate grid) and a lane that walked its
//      own list would hold its whole wave for the longest one (measured: 62 us at any batch size, the time of ONE
//      32-region query);
//   4. a lane forms gamma of its pair (product of the region's factors in the order of the reference's loop,
//      model.py:88-93 -- bit-identical to the dense kernels' gamma) and runs the K centres of that region: per-lane LDS
//      reads of the centre, broadcast reads of the weight rows; the pair's partial output goes to an LDS tile and the
//      query's own lane adds its pairs in region order.
// Result per query = sum over its live regions in ascending region order of the per-region sums: deterministic,
// independent of the batch it is part of and of which lane served which pair, and equal to the dense kernels up to
// the order of the float32 sum (K1 splits the centres over waves).
//
// The list capacity is a HARD bound computed on the host from the card (max number of simultaneously non-zero
// ranges per dimension, intervals widened by 1e-4), nets whose bound or tables do not fit are served by the dense K1.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "rbf_forward.h"
#include "rbf_vjp_f16.h"
#include "rollout_step.h"

namespace irbfn {

#ifndef IRBFN_SP_NT
#define IRBFN_SP_NT 256
#endif
constexpr int kSpNT = IRBFN_SP_NT;      // lanes (= queries) per workgroup
constexpr int kSpMaxE = 32;             // factor entries (bits of the activity mask)
constexpr int kSpMaxCap = 96;           // list capacity per query
constexpr int kSpMaxRegions = 1024;     // the scan is linear in the number of regions
constexpr size_t kSpMaxLds = 150 * 1024;

int sparse_ops(int O) { return O <= 2 ? 2 : (O <= 5 ? 5 : (O <= 10 ? 10 : (O <= 16 ? 16 : -1))); }

static inline int sp_entry_width(int DC) { return (DC + 1 + 3) & ~3; }

// The net as ONE device image, laid out as it sits in LDS (dwords): [ centre table | region index words | region masks |
// factor entries | Dense weight rows ], padded to whole KiB so that it is copied by LDS-DMA pieces alone.
struct SpImg {
  int idx, req, ent, wtab, small, ctab, total;
};
constexpr int kSpWP = 16;               // floats per weight row (OP <= 16; 16-byte reads)
constexpr int kSpPartPitch = 20;        // floats per pair of the partial-output tile: 16-byte rows, conflict-free for b128 reads
__host__ __device__ inline SpImg sp_img_layout(int nr, int RS, int E, int K) {
  SpImg m;
  int o = 0;
  m.idx = o;  o += nr * 2;
  o = (o + 3) & ~3;
  m.req = o;  o += (nr + 31) & ~31;                              // whole 32-region words (padding: masks nobody satisfies)
  m.ent = o;  o += E * 4;
  m.wtab = o; o += K * kSpWP;
  m.small = (o + 255) & ~255;                                    // the small tables: whole KiB pieces, requested first
  m.ctab = m.small;
  m.total = m.small + ((nr * RS + 255) & ~255);
  return m;
}
// LDS layout (dwords), shared by host and kernel: the image, then the per-lane columns
struct SpLds {
  int ftab, xs, part, hw, flat_r, flat_q, wsum, total;
};
__host__ __device__ inline SpLds sp_lds_layout(int img_in_lds, int nr, int E, int cap, int wide) {
  SpLds l;
  int o = img_in_lds;                            // whole image, or the small tables only (centre table read from global)
  l.ftab = o; o += E * kSpNT;
  l.xs = o;   o += 9 * kSpNT;                                    // x tile, pitch D|1 <= 9
  o = (o + 3) & ~3;
  l.part = o; o += kSpPartPitch * kSpNT;                         // partial outputs of a round
  l.hw = o;   o += ((nr + 31) >> 5) * kSpNT;                     // hit words: bit r of a query's column = gamma_r != 0
  l.flat_r = o; o += (cap * kSpNT * (wide ? 2 : 1) + 3) / 4;     // region of pair i (query-major numbering)
  l.flat_q = o; o += (cap * kSpNT + 3) / 4;                      // owner (query) of pair i
  l.wsum = o; o += kSpNT / 64 + 1;
  l.total = o;
  return l;
}

// ---------------------------------------------------------------------------------------------------------------
// host: tables of the card (once per descriptor)
// ---------------------------------------------------------------------------------------------------------------
int sparse_setup(irbfn_net* net, const float* lo, const float* hi, const float* delta, const int* dr) {
  net->sp_ok = 0;
  const int ns = net->nsplit, nr = net->n_ranges, mr = net->max_ranges;
  if (net->R < 2 || nr < 2 || ns < 1 || ns > kMaxSplit || nr > kSpMaxRegions) return IRBFN_OK;
  const int OPS = sparse_ops(net->O);
  if (OPS < 0) return IRBFN_OK;
  // compact factor entries: one per (dimension, range index that some region uses)
  std::vector<int> entry_of((size_t)ns * mr, -1);
  std::vector<float> ent;                 // [E][4] = lo, hi, delta, (float)d
  int E = 0;
  double cap = 1.0, mean = 1.0;
  for (int d = 0; d < ns; ++d) {
    if (!(delta[d] > 0.0f) || !std::isfinite(delta[d])) return IRBFN_OK;     // the margin below needs delta > 0
    std::vector<int> used;
    for (int r = 0; r < nr; ++r) used.push_back(dr[r * ns + d]);
    std::sort(used.begin(), used.end());
    used.erase(std::unique(used.begin(), used.end()), used.end());
    // non-zero interval of a range: (lo - Z/delta, hi + Z/delta), widened by 1e-4 (float32 rounding of delta*(x-lo))
    const double marg = (double)kGateSatZ / delta[d] * (1.0 + 1e-4);
    std::vector<std::pair<double, int>> ev;
    double L = 1e300, U = -1e300, covered = 0.0;
    for (int j : used) {
      const double a = lo[d * mr + j], b = hi[d * mr + j];
      if (!std::isfinite(a) || !std::isfinite(b)) return IRBFN_OK;
      entry_of[(size_t)d * mr + j] = E++;
      ent.insert(ent.end(), {(float)a, (float)b, delta[d], (float)d});
      const double tiny = 1e-6 * (fabs(a) + fabs(b) + 1.0);
      ev.push_back({a - marg - tiny, +1});
      ev.push_back({b + marg + tiny, -1});
      L = std::min(L, a); U = std::max(U, b);
    }
    std::sort(ev.begin(), ev.end(), [](const std::pair<double, int>& p, const std::pair<double, int>& q) {
      return p.first < q.first || (p.first == q.first && p.second > q.second);      // openings first: closed intervals
    });
    int cur = 0, best = 0;
    for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
    for (int j : used) {                    // expected number of non-zero ranges for x uniform over the dimension's span
      const double a = std::max(L, lo[d * mr + j] - marg), b = std::min(U, hi[d * mr + j] + marg);
      covered += std::max(0.0, b - a);
    }
    cap *= best;
    mean *= (U > L) ? std::max(1.0, covered / (U - L)) : (double)used.size();
  }
  if (E > kSpMaxE) return IRBFN_OK;
  cap = std::min(cap, (double)nr);
  mean = std::min(mean, (double)nr);
  if (cap > kSpMaxCap) return IRBFN_OK;
  const int EW = sp_entry_width(net->DC);
  int RS = net->K * EW;
  if (((RS / 4) & 1) == 0) RS += 4;          // odd number of 16-byte slots per region: lanes on different regions spread over the banks
  // LDS of the forward: centre table + region index words + per-lane columns (factors, list, x tile)
  const SpImg im = sp_img_layout(nr, RS, E, net->K);
  if ((size_t)sp_lds_layout(im.total, nr, E, (int)cap, nr > 256 ? 1 : 0).total * 4 > kSpMaxLds) return IRBFN_OK;

  std::vector<unsigned> req(nr), idx((size_t)nr * 2, 0xFFFFFFFFu);
  for (int r = 0; r < nr; ++r) {
    unsigned m = 0;
    for (int d = 0; d < ns; ++d) {
      const int e = entry_of[(size_t)d * mr + dr[r * ns + d]];
      m |= 1u << e;
      unsigned& w = idx[(size_t)r * 2 + (d >> 2)];
      w = (w & ~(0xFFu << (8 * (d & 3)))) | ((unsigned)e << (8 * (d & 3)));
    }
    req[r] = m;
  }
  std::vector<unsigned> img((size_t)im.total, 0u);
  memcpy(img.data() + im.idx, idx.data(), idx.size() * sizeof(unsigned));
  for (int i = 0; i < ((nr + 31) & ~31); ++i) img[im.req + i] = i < nr ? req[i] : 0xFFFFFFFFu;
  memcpy(img.data() + im.ent, ent.data(), ent.size() * sizeof(float));
  hipError_t e = hipMalloc((void**)&net->sp_img, (size_t)im.total * 4);
  if (e == hipSuccess) e = hipMemcpy(net->sp_img, img.data(), (size_t)im.total * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { g_last_hip_error = (int)e; return IRBFN_ERR_HIP; }
  net->sp_E = E; net->sp_cap = (int)cap; net->sp_RS = RS; net->sp_EW = EW; net->sp_OPS = OPS;
  net->sp_mean_active = (float)mean;
  net->sp_ok = 1;
  return IRBFN_OK;
}

void sparse_free(irbfn_net* net) {
  if (net->sp_img) (void)hipFree(net->sp_img);
  net->sp_img = nullptr;
  net->sp_ok = 0;
}

// would the automatic dispatch take the region-sparse kernels?  (expected share of non-zero regions, uniform queries)
bool sparse_preferred(const irbfn_net* net, int64_t B) {
  // measured (profiles/r03_sparse_*.txt): 128 regions x 10 centres, 3.6 non-zero regions per query: 24 us against 74 dense;
  // 12 regions x 100 centres, 5.7 of 12 non-zero: 60 us against 48 dense (per-lane centre reads cost more than half the
  // pairs save) -> only clearly sparse gates
  return net->sp_ok && B > 64 && net->sp_mean_active <= 0.25f * (float)net->n_ranges;
}

// ---------------------------------------------------------------------------------------------------------------
// pack (once per parameter upload): ctab[r][k] = { c[0..DC), folded width scale, 0.. }, wtab[k] = W[k, 0..kSpWP) -- a role of the
// first pack launch (pack_all.hip, role P); here only where the tables live
// ---------------------------------------------------------------------------------------------------------------
void sparse_pack_tables(const irbfn_net* net, float** ctab, float** wtab, int* wp) {
  const SpImg im = sp_img_layout(net->n_ranges, net->sp_RS, net->sp_E, net->K);
  *ctab = net->sp_img + im.ctab;
  *wtab = net->sp_img + im.wtab;
  *wp = kSpWP;
}

// ---------------------------------------------------------------------------------------------------------------
// K1r
// ---------------------------------------------------------------------------------------------------------------
struct SpArgs {
  const float* __restrict__ x;
  float* __restrict__ out;
  const float* __restrict__ bias;
  const float* __restrict__ img;       // sp_img_layout
  long B;
  int Dreal, O, K, nr, E, ns, cap, RS, basis, wide_list, gc;
  const int* __restrict__ mirror;
  int sv0;
  // fused roll-out (ROLL)
  const float* __restrict__ state0;
  float* __restrict__ states;
  int T, mode;
  DynParams dp;
};

// Diagnosis build only (tools/build_variant.py ... -DIRBFN_SP_STAMPS): wave 0 of block 0 records s_memtime at the phase
// boundaries into a device array of its own; no output depends on it and the regular build contains none of it.
#ifdef IRBFN_SP_STAMPS
__device__ unsigned long long g_sp_stamps[32];
#define IRBFN_SP_STAMP(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sp_stamps[n] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define IRBFN_SP_STAMP(n) do { } while (0)
#endif

// GC: the centre table stays in global memory (L1 / L2 resident: 43 KB on the 128-region planner) and the lanes gather their
// centres from there -- 47 KB of LDS per workgroup instead of 93, i.e. THREE workgroups per CU instead of one.  The kernel is
// a chain of dependent phases with one wave per SIMD; co-resident workgroups are what hides it.
template <int D, int OP, int BC, bool ROLL, bool GC>
__global__ __launch_bounds__(kSpNT) void rbf_fwd_sparse(const SpArgs a) {
  extern __shared__ float lds[];
  constexpr int NT = kSpNT;
  constexpr int EW = (D + 1 + 3) & ~3;
  constexpr int XP = D | 1;                      // odd pitch of the x tile
  const SpImg IM = sp_img_layout(a.nr, a.RS, a.E, a.K);
  const SpLds L = sp_lds_layout(GC ? IM.small : IM.total, a.nr, a.E, a.cap, a.wide_list);
  const float* ctab_s = GC ? nullptr : lds + IM.ctab;
  const unsigned* idx_s = reinterpret_cast<const unsigned*>(lds + IM.idx);
  const unsigned* req_s = reinterpret_cast<const unsigned*>(lds + IM.req);
  const float* ent_s = lds + IM.ent;
  const float* wtab_s = lds + IM.wtab;
  float* ftab = lds + L.ftab;
  float* xs = lds + L.xs;
  float* part = lds + L.part;
  unsigned* hw_s = reinterpret_cast<unsigned*>(lds + L.hw);
  unsigned char* flat_r8 = reinterpret_cast<unsigned char*>(lds + L.flat_r);
  unsigned short* flat_r16 = reinterpret_cast<unsigned short*>(lds + L.flat_r);
  unsigned char* flat_q = reinterpret_cast<unsigned char*>(lds + L.flat_q);
  int* wsum = reinterpret_cast<int*>(lds + L.wsum);
  static_assert(NT <= 256, "pair owners are stored as bytes");

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  IRBFN_SP_STAMP(0);
  const long row0 = (long)blockIdx.x * NT;
  const long left = a.B - row0;
  const int nvalid = left < NT ? (int)left : NT;
  const int Dr = a.Dreal;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const unsigned char* gimg = reinterpret_cast<const unsigned char*>(a.img) + lane * 16;
  unsigned char* limg = reinterpret_cast<unsigned char*>(lds);

  // ---- phase A: the small tables of the net (region masks and index words, factor entries, weight rows: everything a lane
  // later reads by a wave-uniform address is a broadcast LDS read, not a scalar load the wave would stall on) by LDS-DMA,
  // and the block's query tile through registers
  {
    float xv[D];
    const float* xsrc = a.x + row0 * Dr;
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int i = tid + u * NT;
      xv[u] = i < nvalid * Dr ? xsrc[i] : 0.0f;
    }
    for (int v = wave; v < IM.small / 256; v += NT / kWave)
      __builtin_amdgcn_global_load_lds((gptr_t)(gimg + v * 1024), (lptr_t)(limg + v * 1024), 16, 0, 0);
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int i = tid + u * NT;
      if (i < nvalid * Dr) {
        const int r = i / Dr, j = i - r * Dr;
        xs[r * XP + j] = xv[u];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // ---- the centre table (43 KB on the 128-region planner: ~2 us at the LDS-DMA rate of one CU) is requested now and lands
  // while the lanes evaluate their gates; nobody reads it before the barrier in front of the rounds
  if constexpr (!GC) {
    for (int v = IM.small / 256 + wave; v < IM.total / 256; v += NT / kWave)
      __builtin_amdgcn_global_load_lds((gptr_t)(gimg + v * 1024), (lptr_t)(limg + v * 1024), 16, 0, 0);
  }

  // ---- this lane's query: factors of every (dimension, range) entry, activity mask.  Groups of four: the four entry
  // reads are in flight together (a store to the factor column may alias them for the compiler: one at a time, every
  // entry cost a full LDS round trip)
  IRBFN_SP_STAMP(1);
  const int q0 = tid < nvalid ? tid : nvalid - 1;
  unsigned M = 0;
  for (int e0 = 0; e0 < a.E; e0 += 4) {
    float4 en[4];
    float xv[4], f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u < a.E ? e0 + u : a.E - 1;
      en[u] = *reinterpret_cast<const float4*>(ent_s + e * 4);    // {lo, hi, delta, dimension}: broadcast read
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = xs[q0 * XP + (int)en[u].w];
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = gate_factor(xv[u], en[u].x, en[u].y, en[u].z);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u < a.E) {
        ftab[(e0 + u) * NT + tid] = f[u];
        M |= (f[u] != 0.0f ? 1u : 0u) << (e0 + u);               // NaN counts as non-zero: it must reach the output
      }
    }
  }
  // ---- scan: regions whose factors are all non-zero (model.py:88-93: gamma = product of the region's factors).  32 regions
  // at a time become a bit word without a branch; the words stay in the lane's LDS column
  IRBFN_SP_STAMP(2);
  int cnt = 0;
  const int nwords = (a.nr + 31) >> 5;
  for (int w = 0; w < nwords; ++w) {
    unsigned h = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const uint4 rq = *reinterpret_cast<const uint4*>(req_s + w * 32 + v * 4);
      h |= ((M & rq.x) == rq.x ? 1u : 0u) << (4 * v);
      h |= ((M & rq.y) == rq.y ? 1u : 0u) << (4 * v + 1);
      h |= ((M & rq.z) == rq.z ? 1u : 0u) << (4 * v + 2);
      h |= ((M & rq.w) == rq.w ? 1u : 0u) << (4 * v + 3);
    }
    const int rem = a.nr - w * 32;               // the padding masks are all ones: only a query with E = 32 live factors meets them
    if (rem < 32) h &= (1u << rem) - 1u;
    if (tid >= nvalid) h = 0;
    // the list capacity is a hard bound for finite queries; a query with NaN coordinates can exceed it -- its output is NaN
    // after the first region already, the surplus bits are dropped
    while (cnt + __builtin_popcount(h) > a.cap) h &= ~(0x80000000u >> __builtin_clz(h));
    cnt += __builtin_popcount(h);
    hw_s[w * NT + tid] = h;
  }
  // ---- number the block's (query, region) pairs query-major: exclusive prefix sum of the list lengths
  IRBFN_SP_STAMP(3);
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  if (lane == kWave - 1) wsum[wave] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NT / kWave; ++w) {
    const int v = wsum[w];
    wbase += w < wave ? v : 0;
    total += v;
  }
  const int my_base = wbase + incl - cnt;
  // ---- the pairs as flat arrays: region and owner of pair i.  One pass over the set bits of the lane's hit words.
  {
    int p = my_base;
    for (int w = 0; w < nwords; ++w) {
      unsigned h = hw_s[w * NT + tid];
      while (h != 0) {
        const int r = w * 32 + __builtin_ctz(h);
        h &= h - 1;
        if (a.wide_list) flat_r16[p] = (unsigned short)r;
        else flat_r8[p] = (unsigned char)r;
        flat_q[p] = (unsigned char)tid;
        ++p;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces of the centre table have landed
  __syncthreads();

  // ---- rounds: one pair per lane; the query's own lane then adds its pairs of the round in region order
  float acc[OP];                                 // this lane's OWN query (tid): running sum over its pairs
#pragma unroll
  for (int o = 0; o < OP; ++o) acc[o] = 0.0f;
  constexpr int WV = (OP + 3) / 4;               // 16-byte reads per weight row / partial-output row
  IRBFN_SP_STAMP(4);
  for (int i0 = 0; i0 < total; i0 += NT) {
    if (i0 == 0) IRBFN_SP_STAMP(5);
    const int i = i0 + tid;
    const bool act = i < total;
    const int ii = act ? i : 0;
    const int qq = flat_q[ii];
    const int r = a.wide_list ? (int)flat_r16[ii] : (int)flat_r8[ii];
    const uint2 iw = *reinterpret_cast<const uint2*>(idx_s + r * 2);
    float xq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) xq[j] = j < Dr ? xs[qq * XP + j] : 0.0f;
    float g = act ? 1.0f : 0.0f;                 // model.py:88-93, factors in dimension order like the dense kernels
#pragma unroll
    for (int d = 0; d < kMaxSplit; ++d) {
      if (d < a.ns) {
        const unsigned e = ((d < 4 ? iw.x : iw.y) >> (8 * (d & 3))) & 0xFFu;
        g *= ftab[e * NT + qq];
      }
    }
    float pacc[OP];
#pragma unroll
    for (int o = 0; o < OP; ++o) pacc[o] = 0.0f;
    const float* cp = (GC ? a.img + IM.ctab : ctab_s) + r * a.RS;
    auto load2 = [&](int kk, float (&cv)[2][EW], float (&wv)[2][4 * WV]) {      // centres kk, kk + 1 and their weight rows
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4* c4 = reinterpret_cast<const float4*>(cp + (kk + u) * EW);
#pragma unroll
        for (int v = 0; v < EW / 4; ++v) {
          const float4 t4 = c4[v];
          cv[u][4 * v] = t4.x; cv[u][4 * v + 1] = t4.y; cv[u][4 * v + 2] = t4.z; cv[u][4 * v + 3] = t4.w;
        }
        const float4* w4 = reinterpret_cast<const float4*>(wtab_s + (kk + u) * kSpWP);
#pragma unroll
        for (int v = 0; v < WV; ++v) {
          const float4 t4 = w4[v];
          wv[u][4 * v] = t4.x; wv[u][4 * v + 1] = t4.y; wv[u][4 * v + 2] = t4.z; wv[u][4 * v + 3] = t4.w;
        }
      }
    };
    auto dist2 = [&](const float* cv) {
      float r2 = 0.0f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float dd = xq[j] - cv[j];
        r2 = __builtin_fmaf(dd, dd, r2);
      }
      return r2;
    };
    if (i0 == 0) IRBFN_SP_STAMP(6);
    int k = 0;
    if constexpr (BC != BC_GENERIC) {
      // two centres per step through two register sets: the next two (and their weight rows) are requested before the
      // current two are used, no copies between the sets
      auto step2 = [&](const float (&cv)[2][EW], const float (&wv)[2][4 * WV]) {
        float tt[2] = {basis_arg<BC>(dist2(cv[0]), cv[0][D]), basis_arg<BC>(dist2(cv[1]), cv[1][D])};
        trans_block<BC, 2>(tt);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float pg = tt[u] * g;
#pragma unroll
          for (int o = 0; o < OP; ++o) pacc[o] = __builtin_fmaf(pg, wv[u][o], pacc[o]);
        }
      };
      if (a.K >= 2) {
        float cA[2][EW], wA[2][4 * WV], cB[2][EW], wB[2][4 * WV];
        load2(0, cA, wA);
        for (; k + 4 <= a.K; k += 4) {
          load2(k + 2, cB, wB);
          step2(cA, wA);
          load2(k + 4 <= a.K - 2 ? k + 4 : a.K - 2, cA, wA);     // the pair after next (at the end: any valid pair)
          step2(cB, wB);
        }
        if (k + 2 <= a.K) {                      // cA holds centres k, k + 1 (k <= K - 2 here)
          step2(cA, wA);
          k += 2;
        }
      }
    }
    for (; k < a.K; ++k) {
      float cv[EW];
      const float4* c4 = reinterpret_cast<const float4*>(cp + k * EW);
#pragma unroll
      for (int v = 0; v < EW / 4; ++v) {
        const float4 t4 = c4[v];
        cv[4 * v] = t4.x; cv[4 * v + 1] = t4.y; cv[4 * v + 2] = t4.z; cv[4 * v + 3] = t4.w;
      }
      const float pg = basis_from_r2<BC>(dist2(cv), cv[D], a.basis) * g;
#pragma unroll
      for (int o = 0; o < OP; ++o) pacc[o] = __builtin_fmaf(pg, wtab_s[k * kSpWP + o], pacc[o]);
    }
    if (i0 == 0) IRBFN_SP_STAMP(7);
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      float4 t4;
      t4.x = pacc[4 * v];
      t4.y = 4 * v + 1 < OP ? pacc[4 * v + 1] : 0.0f;
      t4.z = 4 * v + 2 < OP ? pacc[4 * v + 2] : 0.0f;
      t4.w = 4 * v + 3 < OP ? pacc[4 * v + 3] : 0.0f;
      *reinterpret_cast<float4*>(part + tid * kSpPartPitch + 4 * v) = t4;
    }
    __syncthreads();
    if (i0 == 0) IRBFN_SP_STAMP(8);
    {                                            // owner lanes: pairs [my_base, my_base + cnt) that fell into this round,
      int j0 = my_base > i0 ? my_base : i0;      // added in region order; four rows in flight per step
      int j1 = my_base + cnt;
      j1 = j1 < i0 + NT ? j1 : i0 + NT;
      auto row = [&](int j, float (&pv)[4 * WV]) {
#pragma unroll
        for (int v = 0; v < WV; ++v) {
          const float4 t4 = *reinterpret_cast<const float4*>(part + (j - i0) * kSpPartPitch + 4 * v);
          pv[4 * v] = t4.x; pv[4 * v + 1] = t4.y; pv[4 * v + 2] = t4.z; pv[4 * v + 3] = t4.w;
        }
      };
      int j = j0;
      for (; j + 4 <= j1; j += 4) {
        float p0[4 * WV], p1[4 * WV], p2[4 * WV], p3[4 * WV];
        row(j, p0); row(j + 1, p1); row(j + 2, p2); row(j + 3, p3);
#pragma unroll
        for (int o = 0; o < OP; ++o) acc[o] = (((acc[o] + p0[o]) + p1[o]) + p2[o]) + p3[o];
      }
      for (; j < j1; ++j) {
        float p0[4 * WV];
        row(j, p0);
#pragma unroll
        for (int o = 0; o < OP; ++o) acc[o] += p0[o];
      }
    }
    __syncthreads();
    if (i0 == 0) IRBFN_SP_STAMP(9);
  }
  IRBFN_SP_STAMP(10);
  const int qq = tid;                            // from here on every lane finishes its own query
  const int qrow = q0;

  // ---- Dense bias, mirror flip, output tile (coalesced)
  const long bq = row0 + qrow;
  const bool flip = a.mirror != nullptr && a.mirror[bq] != 0;
#pragma unroll
  for (int o = 0; o < OP; ++o) {
    if (o < a.O) {
      float v = acc[o] + a.bias[o];
      if (flip && o >= a.sv0) v = -v;
      acc[o] = v;
    }
  }
  if (a.out != nullptr && tid < nvalid) {        // 4 * O bytes per lane, consecutive lanes = consecutive rows
    float* dst = a.out + bq * a.O;
#pragma unroll
    for (int o = 0; o < OP; ++o)
      if (o < a.O) dst[o] = acc[o];
  }
  IRBFN_SP_STAMP(11);
  if constexpr (ROLL) {
    // batched IRBFNPlanner.plan (irbfn_planner.py:205-212): the lane that holds a query's controls integrates its
    // trajectory; the T x S states of the block's rows -- one contiguous piece of HBM -- leave through an LDS tile
    constexpr int TM = OP / 2;
    const int T = a.T;
    const int Sdim = (a.mode == IRBFN_ROLLOUT_FULLINT) ? 5 : (a.mode == IRBFN_ROLLOUT_FRENET_LS ? 8 : 7);
    const int rowf = T * Sdim;                   // <= 64
    __syncthreads();
    float* stage = lds;                          // [NT][65] over the centre table and the columns (all dead)
    float* o = stage + qq * 65;
    if (a.mode == IRBFN_ROLLOUT_ST_SELECT || a.mode == IRBFN_ROLLOUT_ST_KS) {
      float s[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) s[i] = a.state0[bq * 7 + i];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (t < T) {
          // controls: u = [a_0..a_{T-1}, sv_0..sv_{T-1}] (dynamics.py:98); T == O/2 <= TM, static register indices
          float ua = 0.0f, us = 0.0f;
#pragma unroll
          for (int o2 = 0; o2 < OP; ++o2) { ua = (o2 == t) ? acc[o2] : ua; us = (o2 == T + t) ? acc[o2] : us; }
          if (a.mode == IRBFN_ROLLOUT_ST_SELECT) st_step<true>(s, ua, us, a.dp);
          else st_step<false>(s, ua, us, a.dp);
#pragma unroll
          for (int i = 0; i < 7; ++i) o[t * 7 + i] = s[i];
        }
      }
    } else if (a.mode == IRBFN_ROLLOUT_FRENET_LS) {
      float s[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] = a.state0[bq * 8 + i];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (t < T) {
          float ua = 0.0f, us = 0.0f;
#pragma unroll
          for (int o2 = 0; o2 < OP; ++o2) { ua = (o2 == t) ? acc[o2] : ua; us = (o2 == T + t) ? acc[o2] : us; }
          frenet_step(s, ua, us, a.dp);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[t * 8 + i] = s[i];
        }
      }
    } else {
      float s[5] = {0.0f, 0.0f, 0.0f, clipf(a.state0[bq], 0.0f, 7.0f), 0.0f};       // train_nmpc.py:319
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (t < T) {
          float ua = 0.0f, us = 0.0f;
#pragma unroll
          for (int o2 = 0; o2 < OP; ++o2) { ua = (o2 == t) ? acc[o2] : ua; us = (o2 == T + t) ? acc[o2] : us; }
          fullint_step(s, ua, us);
#pragma unroll
          for (int i = 0; i < 5; ++i) o[t * 5 + i] = s[i];
        }
      }
    }
    __syncthreads();
    float* gout = a.states + row0 * (long)rowf;
    for (int i = tid; i < nvalid * rowf; i += NT) {
      const int r = i / rowf, c = i - r * rowf;
      gout[i] = stage[r * 65 + c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------------------------
template <int D, int OP, int BC, bool ROLL, bool GC>
static int sp_launch_gc(const SpArgs& a, size_t lds, hipStream_t s, int* grid_out) {
  auto k = rbf_fwd_sparse<D, OP, BC, ROLL, GC>;
  // the attribute call is a slow host call (it made back-to-back launches host-bound at ~30 us each): once per
  // instance and device is enough -- the largest size any card may ask for
  static thread_local int attr_dev = -1;
  int dev = 0;
  IRBFN_HIP_CHECK(hipGetDevice(&dev));
  if (attr_dev != dev) {
    IRBFN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kSpMaxLds));
    attr_dev = dev;
  }
  const long grid = (a.B + kSpNT - 1) / kSpNT;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kSpNT), lds, s, a);
  IRBFN_HIP_CHECK(hipGetLastError());
  *grid_out = (int)grid;
  return IRBFN_OK;
}

template <int D, int OP, int BC, bool ROLL>
static int sp_launch_one(const SpArgs& a, size_t lds, hipStream_t s, int* grid_out) {
  return a.gc ? sp_launch_gc<D, OP, BC, ROLL, true>(a, lds, s, grid_out) : sp_launch_gc<D, OP, BC, ROLL, false>(a, lds, s, grid_out);
}

template <int D, int OP, bool ROLL>
static int sp_launch_bc(const SpArgs& a, int bc, size_t lds, hipStream_t s, int* g) {
  switch (bc) {
    case BC_GAUSS: return sp_launch_one<D, OP, BC_GAUSS, ROLL>(a, lds, s, g);
    case BC_IQ: return sp_launch_one<D, OP, BC_IQ, ROLL>(a, lds, s, g);
    case BC_IMQ: return sp_launch_one<D, OP, BC_IMQ, ROLL>(a, lds, s, g);
    default:
      if constexpr (ROLL) return IRBFN_ERR_UNSUPPORTED;
      else return sp_launch_one<D, OP, BC_GENERIC, false>(a, lds, s, g);
  }
}

template <int D, bool ROLL>
static int sp_launch_d(const SpArgs& a, int OPS, int bc, size_t lds, hipStream_t s, int* g) {
  switch (OPS) {
    case 2: return sp_launch_bc<D, 2, ROLL>(a, bc, lds, s, g);
    case 5: if constexpr (ROLL) return IRBFN_ERR_UNSUPPORTED; else return sp_launch_bc<D, 5, false>(a, bc, lds, s, g);
    case 10: return sp_launch_bc<D, 10, ROLL>(a, bc, lds, s, g);
    case 16: return sp_launch_bc<D, 16, ROLL>(a, bc, lds, s, g);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

#ifdef IRBFN_SP_STAMPS
extern "C" int irbfn_debug_sparse_stamps(unsigned long long* out32) {
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_sp_stamps), sizeof(unsigned long long) * 32);
}
#endif

// forward (states == nullptr) or the one-launch planning tick (states != nullptr, O == 2T, T*S <= 64)
int launch_forward_sparse(irbfn_net* net, const float* x, float* out, int64_t B, const int* mirror, int sv0,
                          int mode, const float* state0, const DynParams* dp, float* states, int T, hipStream_t s) {
  if (!net->sp_ok) return IRBFN_ERR_UNSUPPORTED;
  const bool roll = states != nullptr;
  if (roll) {
    if (net->bclass == BC_GENERIC || net->O != 2 * T || T * rollout_state_dim(mode) > 64) return IRBFN_ERR_UNSUPPORTED;
    if (net->sp_OPS != 2 && net->sp_OPS != 10 && net->sp_OPS != 16) return IRBFN_ERR_UNSUPPORTED;
    if (net->DC != 7 && net->DC != 8) return IRBFN_ERR_UNSUPPORTED;
  }
  SpArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.out = out; a.bias = net->bias; a.img = net->sp_img;
  a.B = (long)B; a.Dreal = net->D; a.O = net->O; a.K = net->K; a.nr = net->n_ranges; a.E = net->sp_E; a.ns = net->nsplit;
  a.cap = net->sp_cap; a.RS = net->sp_RS; a.basis = net->basis; a.wide_list = net->n_ranges > 256 ? 1 : 0;
  a.mirror = mirror; a.sv0 = sv0;
  if (roll) { a.state0 = state0; a.states = states; a.T = T; a.mode = mode; a.dp = *dp; }
  // Where the centre table lives.  One workgroup per CU is all that fits with the table in LDS (93 KB on the 128-region
  // planner): best while the launch has at most one workgroup per CU (B = 65536: 22.2 vs 23.4 us; the tick 27.5 vs 37.4).
  // Beyond that the workgroups would queue up in rounds -- with the table gathered from global memory (L2-resident) three
  // fit per CU and the whole launch is resident at once (train step at the reference's batch of 80000: 135 vs 149 us).
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    IRBFN_HIP_CHECK(hipGetDevice(&dev));
    IRBFN_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    n_cu = v > 0 ? v : 256;
  }
  a.gc = ((B + kSpNT - 1) / kSpNT > n_cu) ? 1 : 0;
  const SpImg imh = sp_img_layout(a.nr, a.RS, a.E, a.K);
  const SpLds L = sp_lds_layout(a.gc ? imh.small : imh.total, a.nr, a.E, a.cap, a.wide_list);
  size_t lds = (size_t)L.total * 4;
  if (roll) lds = std::max(lds, (size_t)kSpNT * 65 * 4);
  if (lds > kSpMaxLds) return IRBFN_ERR_UNSUPPORTED;
  int grid = 0, rc;
  switch (net->DC) {
    case 3: rc = roll ? IRBFN_ERR_UNSUPPORTED : sp_launch_d<3, false>(a, net->sp_OPS, net->bclass, lds, s, &grid); break;
    case 4: rc = roll ? IRBFN_ERR_UNSUPPORTED : sp_launch_d<4, false>(a, net->sp_OPS, net->bclass, lds, s, &grid); break;
    case 7: rc = roll ? sp_launch_d<7, true>(a, net->sp_OPS, net->bclass, lds, s, &grid)
                      : sp_launch_d<7, false>(a, net->sp_OPS, net->bclass, lds, s, &grid); break;
    case 8: rc = roll ? sp_launch_d<8, true>(a, net->sp_OPS, net->bclass, lds, s, &grid)
                      : sp_launch_d<8, false>(a, net->sp_OPS, net->bclass, lds, s, &grid); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  if (rc == IRBFN_OK) {
    snprintf(net->last_name, sizeof(net->last_name), "rbf_fwd_sparse<D=%d,OP=%d,BC=%d,ROLL=%d,GC=%d>", net->DC, net->sp_OPS,
             net->bclass, (int)roll, a.gc);
    net->last_grid = grid;
    net->last_block = kSpNT;
  }
  return rc;
}


// ===============================================================================================================
// K2r: region-sparse parameter VJP
// ===============================================================================================================
// The dense K2 streams EVERY query past every centre (lane = centre, 206 us for the 128-region planner at the
// reference's batch of 80000, plus 30-60 us of query packing).  Here the (query, region) pairs with gamma != 0 -- 3.6 per
// query -- are listed per REGION, in batch order (no position comes from an atomic: the order of a region's list, hence the
// order of every sum, is fixed -> bitwise reproducible gradients):
//   sparse_pairs_kernel   per block of 256 queries: the block's pairs {query index, gamma}, sorted by region (stable), into
//                         the block's own segment of the pair buffer; per (block, region): count and offset in the segment
//   sparse_scan_kernel    per region: exclusive prefix of the counts over the blocks, region totals
//   rbf_vjp_sparse        one wave per (region, slice of its list): pair i of the region = pair (i - prefix) of the block
//                         found by binary search in the region's prefix row; lanes = pairs (P per lane in registers: x,
//                         cotangent row, gamma), loop over the region's K centres (wave-uniform: scalar loads), per centre the
//                         D + 1 + O sums over the wave's pairs by a fixed butterfly, accumulated per slice in LDS -> slab
//   vjp_reduce_kernel (+ regions)  the dense path's fixed-order slab reduce (rbf_vjp.hip)
// (A first version counted, scanned and then re-ran the whole gate evaluation to scatter the pairs to their final places:
// that second pass cost 28 us of the 94 us VJP.)
struct SpPairArgs {
  const float* __restrict__ x;
  const float* __restrict__ img;
  long B;
  int Dreal, nr, E, ns, RS, K, cap, nblk;
  int* __restrict__ cnt;               // [nblk][nr] pairs of (block, region)
  int* __restrict__ loff;              // [nblk][nr] their offset inside the block's segment
  int2* __restrict__ pairs;            // [nblk][NT * cap] {query index, gamma bits}
};

// bit u of every lane's word -> ballot -> lane LB + u of (mlo, mhi).  v_writelane_b32 takes ONE scalar register (constant-bus
// rule): the lane select is an inline constant, hence the compile-time recursion (this clang has no builtin for the instruction)
template <int LB, int U>
struct SpTranspose {
  static __device__ __forceinline__ void run(unsigned h, unsigned& mlo, unsigned& mhi) {
    const unsigned long long bal = __ballot((h >> U) & 1u);
    // gfx940+: a VALU that reads an SGPR a VALU has just written (the ballot's v_cmp) needs 2 wait states.  hipcc inserts them
    // for its own instructions, never inside inline asm: without the s_nop the lane received a STALE mask now and then
    // (two passes disagreed -> pairs read from unwritten slots -> memory fault on the second call).
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                 : "+v"(mlo), "+v"(mhi)
                 : "s"((unsigned)bal), "s"((unsigned)(bal >> 32)), "n"(LB + U));
    SpTranspose<LB, U + 1>::run(h, mlo, mhi);
  }
};
template <int LB>
struct SpTranspose<LB, 32> {
  static __device__ __forceinline__ void run(unsigned, unsigned&, unsigned&) {}
};

template <int D>
__global__ __launch_bounds__(kSpNT) void sparse_pairs_kernel(const SpPairArgs a) {
  extern __shared__ float lds[];
  constexpr int NT = kSpNT;
  constexpr int NW = NT / kWave;
  constexpr int XP = D | 1;
  const SpImg IM = sp_img_layout(a.nr, a.RS, a.E, a.K);
  const unsigned* idx_s = reinterpret_cast<const unsigned*>(lds + IM.idx);
  const unsigned* req_s = reinterpret_cast<const unsigned*>(lds + IM.req);
  const float* ent_s = lds + IM.ent;
  const int nwords = (a.nr + 31) >> 5;
  const int nr64 = (a.nr + 63) & ~63;
  float* ftab = lds + IM.small;                  // [E][NT]
  float* xs = ftab + a.E * NT;                   // [NT][XP]
  unsigned* hw_s = reinterpret_cast<unsigned*>(xs + 9 * NT);          // [nwords][NT] hit words of the lanes
  uint2* masks_s = reinterpret_cast<uint2*>(hw_s + nwords * NT);      // [NW][nr64] lanes of the wave that hit region r
  int* woff_s = reinterpret_cast<int*>(masks_s + NW * nr64);          // [NW][nr64] pairs of (block, r) in earlier waves
  int* loff_s = woff_s + NW * nr64;              // [nr64] offset of (block, r) in the block's segment
  int* scan_s = loff_s + nr64;                   // [NW + 1]

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid >> 6;
  const long row0 = (long)blockIdx.x * NT;
  const long left = a.B - row0;
  const int nvalid = left < NT ? (int)left : NT;
  const int Dr = a.Dreal;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  {
    float xv[D];
    const float* xsrc = a.x + row0 * Dr;
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int i = tid + u * NT;
      xv[u] = i < nvalid * Dr ? xsrc[i] : 0.0f;
    }
    const unsigned char* gimg = reinterpret_cast<const unsigned char*>(a.img) + lane * 16;
    unsigned char* limg = reinterpret_cast<unsigned char*>(lds);
    for (int v = wave; v < IM.small / 256; v += NW)
      __builtin_amdgcn_global_load_lds((gptr_t)(gimg + v * 1024), (lptr_t)(limg + v * 1024), 16, 0, 0);
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int i = tid + u * NT;
      if (i < nvalid * Dr) {
        const int r = i / Dr, j = i - r * Dr;
        xs[r * XP + j] = xv[u];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // factors + activity mask: the forward's code (rbf_fwd_sparse), same groups of four
  const int q0 = tid < nvalid ? tid : nvalid - 1;
  unsigned M = 0;
  for (int e0 = 0; e0 < a.E; e0 += 4) {
    float4 en[4];
    float xv[4], f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u < a.E ? e0 + u : a.E - 1;
      en[u] = *reinterpret_cast<const float4*>(ent_s + e * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = xs[q0 * XP + (int)en[u].w];
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = gate_factor(xv[u], en[u].x, en[u].y, en[u].z);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u < a.E) {
        ftab[(e0 + u) * NT + tid] = f[u];
        M |= (f[u] != 0.0f ? 1u : 0u) << (e0 + u);
      }
    }
  }
  // hit words (as the forward), and their TRANSPOSE: for every region the 64-bit mask of the wave's lanes that hit it.
  // Bit u of a word -> one ballot -> lane (r mod 64) of a register pair: 4 instructions per region, no branch.  The mask
  // gives the wave's count of a region (popcount) and a lane's rank in it (popcount below the lane): a stable order.
  int cnt = 0;
  unsigned mlo = 0, mhi = 0;
  for (int w = 0; w < nwords; ++w) {
    unsigned h = 0;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const uint4 rq = *reinterpret_cast<const uint4*>(req_s + w * 32 + v * 4);
      h |= ((M & rq.x) == rq.x ? 1u : 0u) << (4 * v);
      h |= ((M & rq.y) == rq.y ? 1u : 0u) << (4 * v + 1);
      h |= ((M & rq.z) == rq.z ? 1u : 0u) << (4 * v + 2);
      h |= ((M & rq.w) == rq.w ? 1u : 0u) << (4 * v + 3);
    }
    const int rem = a.nr - w * 32;
    if (rem < 32) h &= (1u << rem) - 1u;
    if (tid >= nvalid) h = 0;
    while (cnt + __builtin_popcount(h) > a.cap) h &= ~(0x80000000u >> __builtin_clz(h));     // as the forward (NaN queries)
    cnt += __builtin_popcount(h);
    hw_s[w * NT + tid] = h;
    if (w & 1) SpTranspose<32, 0>::run(h, mlo, mhi);
    else SpTranspose<0, 0>::run(h, mlo, mhi);
    if ((w & 1) == 1 || w == nwords - 1) {       // 64 regions done: lane l holds the mask of region (w / 2) * 64 + l
      masks_s[wave * nr64 + (w >> 1) * 64 + lane] = make_uint2(mlo, mhi);
      mlo = 0; mhi = 0;
    }
  }
  __syncthreads();
  // (block, region): pairs in the earlier waves of the block, block count, offset in the segment (exclusive scan over r)
  int run_base = 0;
  for (int r0 = 0; r0 < nr64; r0 += NT) {
    const int r = r0 + tid;
    int tot = 0;
    if (r < nr64) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const uint2 mk = masks_s[w * nr64 + r];
        woff_s[w * nr64 + r] = tot;
        tot += __builtin_popcount(mk.x) + __builtin_popcount(mk.y);
      }
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == kWave - 1) scan_s[wave] = incl;
    __syncthreads();
    int wb = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int v = scan_s[w];
      wb += w < wave ? v : 0;
      all += v;
    }
    if (r < nr64) loff_s[r] = run_base + wb + incl - tot;
    if (r < a.nr) {
      a.cnt[(size_t)blockIdx.x * a.nr + r] = tot;
      a.loff[(size_t)blockIdx.x * a.nr + r] = run_base + wb + incl - tot;
    }
    run_base += all;
    __syncthreads();
  }
  // each lane walks its own hits: rank from the region's mask, gamma from the region's factors, position, store
  int2* seg = a.pairs + (size_t)blockIdx.x * NT * a.cap;
  for (int w = 0; w < nwords; ++w) {
    unsigned h = hw_s[w * NT + tid];
    while (h != 0) {
      const int r = w * 32 + __builtin_ctz(h);
      h &= h - 1;
      const uint2 mk = masks_s[wave * nr64 + r];
      const uint2 iw = *reinterpret_cast<const uint2*>(idx_s + r * 2);
      const int pos0 = loff_s[r] + woff_s[wave * nr64 + r];
      const int rank = __builtin_amdgcn_mbcnt_hi(mk.y, __builtin_amdgcn_mbcnt_lo(mk.x, 0u));
      float g = 1.0f;                            // model.py:88-93, dimension order: the gamma of every other kernel, bit for bit
#pragma unroll
      for (int d = 0; d < kMaxSplit; ++d) {
        if (d < a.ns) {
          const unsigned e = ((d < 4 ? iw.x : iw.y) >> (8 * (d & 3))) & 0xFFu;
          g *= ftab[e * NT + tid];
        }
      }
      seg[pos0 + rank] = make_int2((int)(row0 + tid), __float_as_int(g));
    }
  }
}

// per region: exclusive prefix of the per-block counts (the order of a region's list = batch order) as a contiguous row
// pre[r][0..nblk) for the binary search of K2r, and the region's total
__global__ __launch_bounds__(256) void sparse_scan_kernel(const int* __restrict__ cnt, int* __restrict__ pre,
                                                          int* __restrict__ tot, int nblk, int nr) {
  __shared__ int sm[256];
  const int r = blockIdx.x, t = threadIdx.x;
  const int chunk = (nblk + 255) / 256;
  const int w0 = t * chunk;
  const int w1 = (w0 + chunk) < nblk ? (w0 + chunk) : nblk;
  int local = 0;
  for (int w = w0; w < w1; ++w) local += cnt[(size_t)w * nr + r];
  sm[t] = local;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {      // Hillis-Steele inclusive scan
    const int v = t >= off ? sm[t - off] : 0;
    __syncthreads();
    sm[t] += v;
    __syncthreads();
  }
  int run = sm[t] - local;
  for (int w = w0; w < w1; ++w) {
    const int c = cnt[(size_t)w * nr + r];
    pre[(size_t)r * nblk + w] = run;
    run += c;
  }
  if (t == 255) tot[r] = sm[255];
}

struct SpVjpArgs {
  const float* __restrict__ x;
  const float* __restrict__ g;
  const float* __restrict__ rec;       // [N][S] dense records {c[DC], folded scale, W[OP]}
  const float* __restrict__ sig2;      // [N]
  const int2* __restrict__ pairs;      // [nblk][NT * cap] region-sorted pairs of every query block
  const int* __restrict__ tot;         // [nr] pairs of the region
  const int* __restrict__ pre;         // [nr][nblk] exclusive prefix of the block counts
  const int* __restrict__ loff;        // [nblk][nr] offset of (block, region) in the block's segment
  float* __restrict__ part;            // [SL][V][Npad]
  int Dreal, O, K, nr, S, Npad, SL, nblk, nsteps, segsize;
  float gscale;
};

#ifndef IRBFN_SP_VP
#define IRBFN_SP_VP 8
#endif
constexpr int kSpVP = IRBFN_SP_VP;     // pairs per lane held in registers

// rows of x / the cotangent are 4-byte aligned only (28- and 40-byte rows): multi-dword loads with that alignment
struct __attribute__((packed, aligned(4))) SpF4 { float v[4]; };
struct __attribute__((packed, aligned(4))) SpF2 { float v[2]; };
template <int N>
__device__ __forceinline__ void sp_load_row(const float* __restrict__ p, int nreal, bool valid, float (&dst)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) dst[j] = 0.0f;
  if (!valid) return;
  if (nreal == N) {                              // the compiled width is the real width: whole row in 16- / 8- / 4-byte pieces
    int j = 0;
#pragma unroll
    for (; j + 4 <= N; j += 4) {
      const SpF4 t = *reinterpret_cast<const SpF4*>(p + j);
      dst[j] = t.v[0]; dst[j + 1] = t.v[1]; dst[j + 2] = t.v[2]; dst[j + 3] = t.v[3];
    }
    if constexpr ((N & 3) >= 2) {
      const SpF2 t = *reinterpret_cast<const SpF2*>(p + (N & ~3));
      dst[N & ~3] = t.v[0]; dst[(N & ~3) + 1] = t.v[1];
    }
    if constexpr (N & 1) dst[N - 1] = p[N - 1];
  } else {
#pragma unroll
    for (int j2 = 0; j2 < N; ++j2)
      if (j2 < nreal) dst[j2] = p[j2];
  }
}

// Sums of NV values over the 64 lanes by a TRANSPOSING butterfly: at every step a lane keeps one half of its values and hands
// the other half to its partner (lane ^ 32, 16, 8, 4, 2), so NV/2 + NV/4 + ... (+ 1 for the last pair) values cross lanes
// instead of 6 NV.  Returns the lane's value index (or -1): lanes with an even number and index < NV hold that value's total.
// Fixed order -> deterministic.
template <int NV>
__device__ __forceinline__ int sp_reduce_transpose(float (&vals)[NV], int lane) {
  static_assert(NV >= 1 && NV <= 64, "one value per lane at the end");
  int n = NV, idx = 0, m = 32;
#pragma unroll
  for (int step = 0; step < 6; ++step) {         // halvings: lane ^ 32, 16, ... while more than one value is left
    if (n > 1) {
      const int half = (n + 1) / 2;
      const bool up = (lane & m) != 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (i < half) {
          const float lo_v = vals[i];
          const float hi_v = (i + half < n) ? vals[i + half < NV ? i + half : 0] : 0.0f;
          const float send = up ? lo_v : hi_v;
          const float keep = up ? hi_v : lo_v;
          vals[i] = keep + __shfl_xor(send, m);
        }
      }
      idx += up ? half : 0;
      n = half;
      m >>= 1;
    }
  }
  const int rest = 2 * m - 1;                    // lane bits not used by a halving: plain butterfly over them
#pragma unroll
  for (int step = 0; step < 6; ++step) {
    if (m >= 1) {
      vals[0] += __shfl_xor(vals[0], m);
      m >>= 1;
    }
  }
  return ((lane & rest) == 0 && idx < NV) ? idx : -1;
}

template <int D, int OP, int BC>
__global__ __launch_bounds__(kWave) void rbf_vjp_sparse(const SpVjpArgs a) {
  extern __shared__ float accs[];                // [K][V]
  constexpr int V = D + 1 + OP;
  constexpr int P = kSpVP;
  const int lane = threadIdx.x;
  const int r = blockIdx.x, sl = blockIdx.y;
  for (int i = lane; i < a.K * V; i += kWave) accs[i] = 0.0f;
  // length of the region's list (regions beyond the card's ranges have gamma == 0, model.py:70: empty)
  const int Lr = r < a.nr ? a.tot[r] : 0;
  const int i_beg = (int)((long)Lr * sl / a.SL), i_end = (int)((long)Lr * (sl + 1) / a.SL);
  const int* pre_r = a.pre + (size_t)(r < a.nr ? r : 0) * a.nblk;
  typedef const float __attribute__((address_space(4)))* crec_t;
  for (int c0 = i_beg; c0 < i_end; c0 += kWave * P) {
    float xq[P][D], gq[P][OP], gam[P];
    // pair i of the region = pair (i - pre[blk]) of the last block whose prefix is <= i: P binary searches side by side
    int lo_b[P], hi_b[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { lo_b[p] = 0; hi_b[p] = a.nblk; }
    for (int st = 0; st < a.nsteps; ++st) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i0 = c0 + p * kWave + lane;
        const int i = i0 < i_end ? i0 : i_beg;   // padding lanes search for a valid pair (their result is not used)
        const int mid = (lo_b[p] + hi_b[p]) >> 1;                // invariant: pre[lo] <= i, pre[hi] > i (or hi == nblk)
        const bool le = pre_r[mid] <= i;                         // mid == lo once hi - lo == 1: nothing changes any more
        lo_b[p] = le ? mid : lo_b[p];
        hi_b[p] = (!le && mid > lo_b[p]) ? mid : hi_b[p];
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int i = c0 + p * kWave + lane;
      const bool valid = i < i_end;
      int2 pr = make_int2(0, 0);
      if (valid) {
        const int blk = lo_b[p];
        pr = a.pairs[(size_t)blk * a.segsize + a.loff[(size_t)blk * a.nr + r] + (i - pre_r[blk])];
      }
      gam[p] = __int_as_float(pr.y);             // 0 for the padding lanes: x = 0 keeps phi finite, gamma = 0 removes it
      sp_load_row<D>(a.x + (long)pr.x * a.Dreal, a.Dreal, valid, xq[p]);
      sp_load_row<OP>(a.g + (long)pr.x * a.O, a.O, valid, gq[p]);
    }
    for (int k = 0; k < a.K; ++k) {
      const int n = r * a.K + k;
      const crec_t rp = (crec_t)(uintptr_t)(a.rec + (size_t)n * a.S);
      float vals[V];                             // [0, D): d centre, D: d log_sig, D + 1 + o: per-centre Dense gradient
#pragma unroll
      for (int v = 0; v < V; ++v) vals[v] = 0.0f;
      const float sc = rp[D];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float diff[D], r2 = 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          diff[j] = xq[p][j] - rp[j];
          r2 = __builtin_fmaf(diff[j], diff[j], r2);
        }
        const float phi = basis_from_r2<BC>(r2, sc, 0);
        float hb = 0.0f;
        const float gphi = gam[p] * phi;
#pragma unroll
        for (int o = 0; o < OP; ++o) {
          hb = __builtin_fmaf(gq[p][o], rp[D + 1 + o], hb);
          vals[D + 1 + o] = __builtin_fmaf(gphi, gq[p][o], vals[D + 1 + o]);
        }
        const float t = hb * gam[p] * dphi_dd2_h<BC>(phi, a.gscale);
        vals[D] = __builtin_fmaf(t, r2, vals[D]);
#pragma unroll
        for (int j = 0; j < D; ++j) vals[j] = __builtin_fmaf(t, diff[j], vals[j]);
      }
      // the wave's sums (fixed butterfly), added to the slice's accumulators: chunks in list order
      const int vi = sp_reduce_transpose<V>(vals, lane);
      if (vi >= 0) accs[k * V + vi] += vals[0];
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  // slab row of this slice: the centre's -2 / sigma^2 multiplies the finished sums (as in K2)
  for (int i = lane; i < a.K * V; i += kWave) {
    const int k = i / V, v = i - k * V;
    const int n = r * a.K + k;
    float val = accs[i];
    if (v <= D) val *= -2.0f * a.sig2[n];
    a.part[((size_t)sl * V + v) * a.Npad + n] = val;
  }
}

template <int D, int OP>
static int spv_launch_bc(const SpVjpArgs& a, int bc, dim3 grid, size_t lds, hipStream_t s) {
  switch (bc) {
    case BC_GAUSS: hipLaunchKernelGGL((rbf_vjp_sparse<D, OP, BC_GAUSS>), grid, dim3(kWave), lds, s, a); break;
    case BC_IQ: hipLaunchKernelGGL((rbf_vjp_sparse<D, OP, BC_IQ>), grid, dim3(kWave), lds, s, a); break;
    case BC_IMQ: hipLaunchKernelGGL((rbf_vjp_sparse<D, OP, BC_IMQ>), grid, dim3(kWave), lds, s, a); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
  IRBFN_HIP_CHECK(hipGetLastError());
  return IRBFN_OK;
}

template <int D>
static int spv_launch_d(const SpVjpArgs& a, int OP, int bc, dim3 grid, size_t lds, hipStream_t s) {
  switch (OP) {
    case 2: return spv_launch_bc<D, 2>(a, bc, grid, lds, s);
    case 4: return spv_launch_bc<D, 4>(a, bc, grid, lds, s);
    case 5: return spv_launch_bc<D, 5>(a, bc, grid, lds, s);
    case 8: return spv_launch_bc<D, 8>(a, bc, grid, lds, s);
    case 10: return spv_launch_bc<D, 10>(a, bc, grid, lds, s);
    case 16: return spv_launch_bc<D, 16>(a, bc, grid, lds, s);
    default: return IRBFN_ERR_UNSUPPORTED;
  }
}

bool sparse_vjp_eligible(const irbfn_net* net) {
  return net->sp_ok && net->bclass != BC_GENERIC && net->OP <= 16 && (size_t)net->K * (net->DC + 1 + net->OP) * 4 <= 60 * 1024;
}

int sparse_vjp_slices(const irbfn_net* net, int64_t B) {
  // slices of a region's list: about one register chunk (64 * P pairs) per wave at the expected list length, and enough
  // waves for the chip
  const double per_region = (double)B * net->sp_mean_active / net->n_ranges;
  int sl = (int)(per_region / (kWave * kSpVP) + 0.999);
  while ((long)sl * net->n_ranges < 1024 && sl < 64) ++sl;
  return sl < 1 ? 1 : (sl > 64 ? 64 : sl);
}

// workspace of the pair lists, in bytes: cnt, loff [nblk][nr]; pre [nr][nblk]; tot [nr]; pairs [nblk][NT * cap]
struct SpWs {
  size_t cnt, loff, pre, tot, pairs, total;
  int nblk;
};
static SpWs sp_ws_layout(const irbfn_net* net, int64_t B) {
  SpWs w;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  w.nblk = (int)((B + kSpNT - 1) / kSpNT);
  const size_t tab = al((size_t)w.nblk * net->n_ranges * 4);
  size_t o = 0;
  w.cnt = o; o += tab;
  w.loff = o; o += tab;
  w.pre = o; o += tab;
  w.tot = o; o += al((size_t)net->n_ranges * 4);
  w.pairs = o; o += al((size_t)w.nblk * kSpNT * net->sp_cap * 8);
  w.total = o;
  return w;
}
size_t sparse_vjp_workspace_bytes(const irbfn_net* net, int64_t B) { return sp_ws_layout(net, B).total; }

// pair lists + K2r -> slab part[SL][V][Npad]; the caller runs the slab reduce and the bias sums
int launch_vjp_sparse(irbfn_net* net, const float* x, const float* gout, int64_t B, void* spws, float* part, int SL, int Npad,
                      hipStream_t s) {
  if (!sparse_vjp_eligible(net)) return IRBFN_ERR_UNSUPPORTED;
  const int nr = net->n_ranges;
  const SpWs w = sp_ws_layout(net, B);
  char* base = static_cast<char*>(spws);
  int* cnt = reinterpret_cast<int*>(base + w.cnt);
  int* loff = reinterpret_cast<int*>(base + w.loff);
  int* pre = reinterpret_cast<int*>(base + w.pre);
  int* tot = reinterpret_cast<int*>(base + w.tot);
  int2* pairs = reinterpret_cast<int2*>(base + w.pairs);
  SpPairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.x = x; pa.img = net->sp_img; pa.B = (long)B; pa.Dreal = net->D; pa.nr = nr; pa.E = net->sp_E; pa.ns = net->nsplit;
  pa.RS = net->sp_RS; pa.K = net->K; pa.cap = net->sp_cap; pa.nblk = w.nblk; pa.cnt = cnt; pa.loff = loff; pa.pairs = pairs;
  const SpImg im = sp_img_layout(nr, net->sp_RS, net->sp_E, net->K);
  const size_t nr64 = ((size_t)nr + 63) & ~(size_t)63;
  const size_t NW = kSpNT / kWave;
  const size_t lds = ((size_t)im.small + (size_t)net->sp_E * kSpNT + 9 * kSpNT + (size_t)((nr + 31) / 32) * kSpNT +
                      2 * NW * nr64 + NW * nr64 + nr64 + NW + 8) * 4;
  if (lds > kSpMaxLds) return IRBFN_ERR_UNSUPPORTED;
  const dim3 gridp((unsigned)w.nblk);
#define IRBFN_SPP(DV)                                                                                                \
  do {                                                                                                               \
    auto k = sparse_pairs_kernel<DV>;                                                                                \
    static thread_local int attr_dev = -1;                                                                           \
    int dev = 0;                                                                                                     \
    IRBFN_HIP_CHECK(hipGetDevice(&dev));                                                                             \
    if (attr_dev != dev) {                                                                                           \
      IRBFN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          (int)kSpMaxLds));                                                          \
      attr_dev = dev;                                                                                                \
    }                                                                                                                \
    hipLaunchKernelGGL(k, gridp, dim3(kSpNT), lds, s, pa);                                                           \
    IRBFN_HIP_CHECK(hipGetLastError());                                                                              \
  } while (0)
  switch (net->DC) {
    case 3: IRBFN_SPP(3); break;
    case 4: IRBFN_SPP(4); break;
    case 7: IRBFN_SPP(7); break;
    case 8: IRBFN_SPP(8); break;
    default: return IRBFN_ERR_UNSUPPORTED;
  }
#undef IRBFN_SPP
  hipLaunchKernelGGL(sparse_scan_kernel, dim3(nr), dim3(256), 0, s, cnt, pre, tot, w.nblk, nr);
  IRBFN_HIP_CHECK(hipGetLastError());
  SpVjpArgs va;
  memset(&va, 0, sizeof(va));
  va.x = x; va.g = gout; va.rec = net->rec; va.sig2 = net->sig2; va.pairs = pairs; va.tot = tot; va.pre = pre; va.loff = loff;
  va.part = part;
  va.Dreal = net->D; va.O = net->O; va.K = net->K; va.nr = nr; va.S = net->S; va.Npad = Npad; va.SL = SL;
  va.nblk = w.nblk; va.segsize = kSpNT * net->sp_cap;
  va.nsteps = 0;
  while ((1 << va.nsteps) < w.nblk) ++va.nsteps;  // halvings of [0, nblk) down to one block
  va.gscale = gauss_scale(net->basis);
  const dim3 gridv(net->R, SL);                  // every region gets its slab rows (zeros where no query is live)
  const size_t ldsv = (size_t)net->K * (net->DC + 1 + net->OP) * 4;
  int rc;
  switch (net->DC) {
    case 3: rc = spv_launch_d<3>(va, net->OP, net->bclass, gridv, ldsv, s); break;
    case 4: rc = spv_launch_d<4>(va, net->OP, net->bclass, gridv, ldsv, s); break;
    case 7: rc = spv_launch_d<7>(va, net->OP, net->bclass, gridv, ldsv, s); break;
    case 8: rc = spv_launch_d<8>(va, net->OP, net->bclass, gridv, ldsv, s); break;
    default: rc = IRBFN_ERR_UNSUPPORTED;
  }
  return rc;
}

}  // namespace irbfn
