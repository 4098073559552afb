// Synthetic K1 inner loop body (no memory, no waits): G centres x [7 x (v_subrev sgpr; v_fmac d,d), v_mul sgpr],
// G v_exp adjacent, G x 5 v_pk_fma_f32 with SGPR-pair weights.  Large unrolled body, 8 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 400
#define U4(X) X X X X
// distance of one centre into R (temps D0/D1), x in v104..v110, c in s[%0..%6], scale %7
#define DIST(R, D0, D1) \
  "v_subrev_f32_e32 " D0 ", %0, v104\n v_mul_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_subrev_f32_e32 " D1 ", %1, v105\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_subrev_f32_e32 " D0 ", %2, v106\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_subrev_f32_e32 " D1 ", %3, v107\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_subrev_f32_e32 " D0 ", %4, v108\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_subrev_f32_e32 " D1 ", %5, v109\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_subrev_f32_e32 " D0 ", %6, v110\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_mul_f32_e32 " R ", %7, " R "\n"
#define DISTV(R, D0, D1) \
  "v_sub_f32_e32 " D0 ", v104, v112\n v_mul_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_sub_f32_e32 " D1 ", v105, v113\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_sub_f32_e32 " D0 ", v106, v114\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_sub_f32_e32 " D1 ", v107, v115\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_sub_f32_e32 " D0 ", v108, v116\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_sub_f32_e32 " D1 ", v109, v117\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n" \
  "v_sub_f32_e32 " D0 ", v110, v118\n v_fmac_f32_e32 " R ", " D0 ", " D0 "\n" \
  "v_mul_f32_e32 " R ", v119, " R "\n"
#define WFV(R) \
  "v_fmac_f32_e32 v60, v84, " R "\n v_fmac_f32_e32 v61, v85, " R "\n v_fmac_f32_e32 v62, v86, " R "\n v_fmac_f32_e32 v63, v87, " R "\n v_fmac_f32_e32 v64, v88, " R "\n" \
  "v_fmac_f32_e32 v65, v89, " R "\n v_fmac_f32_e32 v66, v90, " R "\n v_fmac_f32_e32 v67, v91, " R "\n v_fmac_f32_e32 v68, v92, " R "\n v_fmac_f32_e32 v69, v93, " R "\n"
#define WPV(RP) \
  "v_pk_fma_f32 v[60:61], " RP ", v[84:85], v[60:61] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[62:63], " RP ", v[86:87], v[62:63] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[64:65], " RP ", v[88:89], v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], " RP ", v[90:91], v[66:67] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[68:69], " RP ", v[92:93], v[68:69] op_sel_hi:[0,1,1]\n"
#define WFVS(R) \
  "v_fmac_f32_e32 v60, " R ", v84\n v_fmac_f32_e32 v61, " R ", v85\n v_fmac_f32_e32 v62, " R ", v86\n v_fmac_f32_e32 v63, " R ", v87\n v_fmac_f32_e32 v64, " R ", v88\n" \
  "v_fmac_f32_e32 v65, " R ", v89\n v_fmac_f32_e32 v66, " R ", v90\n v_fmac_f32_e32 v67, " R ", v91\n v_fmac_f32_e32 v68, " R ", v92\n v_fmac_f32_e32 v69, " R ", v93\n"
#define EXP(R) "v_exp_f32_e32 " R ", " R "\n"
// weight rows: phi in R (pair register R:R+1 with op_sel_hi 0 = broadcast low), accumulators v[60:69]
#define WPK(RP) \
  "v_pk_fma_f32 v[60:61], " RP ", %8, v[60:61] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[62:63], " RP ", %9, v[62:63] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[64:65], " RP ", %10, v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], " RP ", %8, v[66:67] op_sel_hi:[0,1,1]\n" \
  "v_pk_fma_f32 v[68:69], " RP ", %9, v[68:69] op_sel_hi:[0,1,1]\n"
#define WFM(R) \
  "v_fmac_f32_e32 v60, %0, " R "\n v_fmac_f32_e32 v61, %1, " R "\n v_fmac_f32_e32 v62, %2, " R "\n v_fmac_f32_e32 v63, %3, " R "\n v_fmac_f32_e32 v64, %4, " R "\n" \
  "v_fmac_f32_e32 v65, %5, " R "\n v_fmac_f32_e32 v66, %6, " R "\n v_fmac_f32_e32 v67, %7, " R "\n v_fmac_f32_e32 v68, %0, " R "\n v_fmac_f32_e32 v69, %1, " R "\n"
#define OPS :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6), "s"(s7), "s"(d0), "s"(d1), "s"(d2) \
  : "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v80","v81","v82","v83"

template <int M>
__global__ __launch_bounds__(1024) void k(float* out, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7) {
  const double d0 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s0) | ((unsigned long long)__builtin_bit_cast(unsigned, s1) << 32));
  const double d1 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s2) | ((unsigned long long)__builtin_bit_cast(unsigned, s3) << 32));
  const double d2 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s4) | ((unsigned long long)__builtin_bit_cast(unsigned, s5) << 32));
  asm volatile("v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n v_mov_b32 v108, 0.25\n v_mov_b32 v109, 0.25\n v_mov_b32 v110, 0.25\n"
               "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n"
               ::: "v104","v105","v106","v107","v108","v109","v110","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
  asm volatile("v_mov_b32 v112, 0.5\n v_mov_b32 v113, 0.5\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 0.5\n v_mov_b32 v116, 0.25\n v_mov_b32 v117, 0.25\n v_mov_b32 v118, 0.25\n v_mov_b32 v119, -1.0\n"
               "v_mov_b32 v84, 0.5\n v_mov_b32 v85, 0.5\n v_mov_b32 v86, 0.5\n v_mov_b32 v87, 0.5\n v_mov_b32 v88, 0.25\n v_mov_b32 v89, 0.25\n v_mov_b32 v90, 0.25\n v_mov_b32 v91, 0.5\n v_mov_b32 v92, 0.5\n v_mov_b32 v93, 0.5\n"
               ::: "v112","v113","v114","v115","v116","v117","v118","v119","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93");
  for (int it = 0; it < ITERS; ++it) {
    if (M == 7) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    if (M == 8) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    if (M == 9) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WPV("v[70:71]") WPV("v[72:73]") WPV("v[74:75]") WPV("v[76:77]") OPS);
    if (M == 10) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83") OPS);
    if (M == 11) asm volatile(WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    // P3: exp result as src0 of the consumers
    if (M == 13) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WFVS("v70") WFVS("v72") WFVS("v74") WFVS("v76") OPS);
    // P4: explicit nops between the exp block and its consumers
    if (M == 14) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") "s_nop 7\n" WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    // P6: software pipelined: exps of group g (regs v70..v76 hold t of the PREVIOUS group) -> distances of the next group
    //     into v71,v73,v75,v77 -> W of group g; then roles swap
    if (M == 15) asm volatile(EXP("v70") EXP("v72") EXP("v74") EXP("v76")
                             DISTV("v71","v80","v81") DISTV("v73","v82","v83") DISTV("v75","v80","v81") DISTV("v77","v82","v83")
                             WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    // P7: exp block copies results through v_mov before use (consumer reads a non-trans result)
    if (M == 16) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") "v_mov_b32 v71, v70\n v_mov_b32 v73, v72\n v_mov_b32 v75, v74\n v_mov_b32 v77, v76\n"
                             WFV("v71") WFV("v73") WFV("v75") WFV("v77") OPS);
    // P8: exps interleaved one per centre but consumers delayed by a whole centre (G = 1 pipelined, all VGPR)
    if (M == 17) asm volatile(DISTV("v70","v80","v81") EXP("v70") WFV("v72") DISTV("v72","v82","v83") EXP("v72") WFV("v70")
                             DISTV("v70","v80","v81") EXP("v70") WFV("v72") DISTV("v72","v82","v83") EXP("v72") WFV("v70") OPS);
#define ALLV(NOPS) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83") \
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") NOPS WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS)
#define SGPK(NOPS) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83") \
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") NOPS WPK("v[70:71]") WPK("v[72:73]") WPK("v[74:75]") WPK("v[76:77]") OPS)
    if (M == 20) ALLV("s_nop 0\n");
    if (M == 21) ALLV("s_nop 1\n");
    if (M == 22) ALLV("s_nop 3\n");
    if (M == 23) ALLV("s_nop 5\n");
    if (M == 24) ALLV("s_nop 7\n s_nop 3\n");
    if (M == 25) ALLV("s_nop 7\n s_nop 7\n");
    if (M == 26) SGPK("s_nop 0\n");
    if (M == 27) SGPK("s_nop 3\n");
    if (M == 28) SGPK("s_nop 7\n");
    if (M == 29) SGPK("s_nop 7\n s_nop 7\n");
    // nop BEFORE the exp block as well
    if (M == 30) asm volatile(DISTV("v70","v80","v81") DISTV("v72","v82","v83") DISTV("v74","v80","v81") DISTV("v76","v82","v83") "s_nop 7\n"
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") "s_nop 7\n" WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    // G = 1 with nop after each exp (all VGPR)
    if (M == 31) asm volatile(DISTV("v70","v80","v81") EXP("v70") "s_nop 7\n" WFV("v70") DISTV("v72","v82","v83") EXP("v72") "s_nop 7\n" WFV("v72")
                             DISTV("v70","v80","v81") EXP("v70") "s_nop 7\n" WFV("v70") DISTV("v72","v82","v83") EXP("v72") "s_nop 7\n" WFV("v72") OPS);
    // G = 1 SGPR/pk with nop after each exp
    if (M == 32) asm volatile(U4(DIST("v70","v80","v81") EXP("v70") "s_nop 7\n" WPK("v[70:71]")) OPS);
    // SGPR distances + VGPR weights
    if (M == 12) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WFV("v70") WFV("v72") WFV("v74") WFV("v76") OPS);
    // G = 1: dist, exp, W  (x4 per body)
    if (M == 0) asm volatile(U4(DIST("v70","v80","v81") EXP("v70") WPK("v[70:71]")) OPS);
    // G = 4: 4 dist (4 result regs, interleaving left to hardware), 4 exps adjacent, 4 W
    if (M == 1) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WPK("v[70:71]") WPK("v[72:73]") WPK("v[74:75]") WPK("v[76:77]") OPS);
    // no exp at all
    if (M == 2) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83")
                             WPK("v[70:71]") WPK("v[72:73]") WPK("v[74:75]") WPK("v[76:77]") OPS);
    // distances only
    if (M == 3) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83") OPS);
    // W only
    if (M == 4) asm volatile(WPK("v[70:71]") WPK("v[72:73]") WPK("v[74:75]") WPK("v[76:77]") OPS);
    // G = 4 with plain fmac + SGPR weights
    if (M == 5) asm volatile(DIST("v70","v80","v81") DIST("v72","v82","v83") DIST("v74","v80","v81") DIST("v76","v82","v83")
                             EXP("v70") EXP("v72") EXP("v74") EXP("v76") WFM("v70") WFM("v72") WFM("v74") WFM("v76") OPS);
    // G = 1, exp placed after the NEXT centre's distance (software-pipelined by hand)
    if (M == 6) asm volatile(DIST("v70","v80","v81") EXP("v72") WPK("v[72:73]") DIST("v72","v82","v83") EXP("v70") WPK("v[70:71]")
                             DIST("v70","v80","v81") EXP("v72") WPK("v[72:73]") DIST("v72","v82","v83") EXP("v70") WPK("v[70:71]") OPS);
  }
  float r;
  asm volatile("v_add_f32 %0, v60, v61\n v_add_f32 %0, %0, v62\n v_add_f32 %0, %0, v64\n v_add_f32 %0, %0, v66\n v_add_f32 %0, %0, v68\n v_add_f32 %0, %0, v70" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int M>
void run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<M><<<blocks, 1024>>>(out, .1f, .2f, .3f, .4f, .5f, .6f, .7f, -.8f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); k<M><<<blocks, 1024>>>(out, .1f, .2f, .3f, .4f, .5f, .6f, .7f, -.8f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wps = blocks * 16.0 / 1024.0;
  printf("%-56s wps=%2.0f %7.3f ms  %6.1f cyc@2.4 per centre per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 4 * wps));
}

int main() {
  float* out; (void)hipMalloc(&out, 512 * 1024 * 4);
  for (int blocks : {512, 256}) {
    run<0>("G=1: dist, exp, 5 pk_fma", out, blocks);
    run<6>("G=1, exp one centre late", out, blocks);
    run<1>("G=4: 4 dist, 4 exp, 4x5 pk_fma", out, blocks);
    run<5>("G=4 with 10 v_fmac sgpr instead of pk", out, blocks);
    run<2>("G=4 without exp", out, blocks);
    run<3>("4 dist only", out, blocks);
    run<4>("4x5 pk_fma only", out, blocks);
    run<7>("all-VGPR: 4 dist, 4 exp, 4x10 fmac", out, blocks);
    run<8>("all-VGPR without exp", out, blocks);
    run<9>("all-VGPR with 5 pk_fma (VGPR pairs)", out, blocks);
    run<10>("all-VGPR 4 dist only", out, blocks);
    run<11>("all-VGPR 4x10 fmac only", out, blocks);
    run<12>("SGPR dist + VGPR-weight fmac, 4 exp", out, blocks);
    run<20>("all-VGPR s_nop 0", out, blocks); run<21>("all-VGPR s_nop 1", out, blocks); run<22>("all-VGPR s_nop 3", out, blocks);
    run<23>("all-VGPR s_nop 5", out, blocks); run<24>("all-VGPR s_nop 7+3", out, blocks); run<25>("all-VGPR s_nop 7+7", out, blocks);
    run<26>("SGPR/pk s_nop 0", out, blocks); run<27>("SGPR/pk s_nop 3", out, blocks); run<28>("SGPR/pk s_nop 7", out, blocks); run<29>("SGPR/pk s_nop 7+7", out, blocks);
    run<30>("all-VGPR nop before and after exps", out, blocks); run<31>("all-VGPR G=1 nop after each exp", out, blocks); run<32>("SGPR/pk G=1 nop after each exp", out, blocks);
    run<13>("all-VGPR, exp result as src0", out, blocks);
    run<14>("all-VGPR, s_nop 7 after exps", out, blocks);
    run<15>("all-VGPR, exps | next dists | W (pipelined)", out, blocks);
    run<16>("all-VGPR, exp -> v_mov -> consumers", out, blocks);
    run<17>("all-VGPR, G=1 pipelined", out, blocks);
  }
  return 0;
}
