// Cost of the squared-distance sequence (v_sub -> v_fmac d,d) on gfx950 in a LARGE unrolled body
// (tiny loops mis-measure: tools/ubench_trans.hip), 8 / 4 waves per SIMD, register banks chosen by hand.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 500
#define U8(X) X X X X X X X X
// x: v104..v110, c: v112..v118 (VGPR) ; r accumulators v100 (bank 0), v101 (bank 1); temps named per variant
#define PAIR(D, R, X, C) "v_sub_f32_e32 " D ", " X ", " C "\n v_fmac_f32_e32 " R ", " D ", " D "\n"
#define PAIRS(D, R, X, C) "v_subrev_f32_e32 " D ", " C ", " X "\n v_fmac_f32_e32 " R ", " D ", " D "\n"
#define CLOB "v100","v101","v102","v103","v120","v121","v122","v123","v124","v125","v126","v127"

template <int M>
__global__ __launch_bounds__(1024) void k(float* out, float s0, float s1, float s2, float s3, float s4, float s5, float s6) {
  asm volatile("v_mov_b32 v100, 1.0\n v_mov_b32 v101, 1.0\n v_mov_b32 v102, 1.0\n v_mov_b32 v103, 1.0\n"
               "v_mov_b32 v104, 0.5\n v_mov_b32 v105, 0.5\n v_mov_b32 v106, 0.5\n v_mov_b32 v107, 0.5\n"
               "v_mov_b32 v108, 0.25\n v_mov_b32 v109, 0.25\n v_mov_b32 v110, 0.25\n v_mov_b32 v111, 0.25\n"
               "v_mov_b32 v112, 0.25\n v_mov_b32 v113, 0.25\n v_mov_b32 v114, 0.25\n v_mov_b32 v115, 0.25\n"
               "v_mov_b32 v116, 0.25\n v_mov_b32 v117, 0.25\n v_mov_b32 v118, 0.25\n v_mov_b32 v119, 0.25\n"
               ::: "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
  const double d0 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s0) | ((unsigned long long)__builtin_bit_cast(unsigned, s1) << 32));
  const double d1 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s2) | ((unsigned long long)__builtin_bit_cast(unsigned, s3) << 32));
  const double d2 = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, s4) | ((unsigned long long)__builtin_bit_cast(unsigned, s5) << 32));
  asm volatile("v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n"
               ::: "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
  for (int it = 0; it < ITERS; ++it) {
    // A: one chain, temps alternate v121/v122 (banks 1,2), r = v100 (bank 0): conflict-free
    if (M == 0) asm volatile(U8(PAIR("v121","v100","v104","v113") PAIR("v122","v100","v105","v114") PAIR("v121","v100","v106","v115") PAIR("v122","v100","v107","v112")
                                PAIR("v121","v100","v108","v117") PAIR("v122","v100","v109","v118") PAIR("v121","v100","v110","v119")) ::: CLOB);
    // B: one chain, single temp v120 in the SAME bank as r (v100): d,d,r all bank 0
    if (M == 1) asm volatile(U8(PAIR("v120","v100","v104","v113") PAIR("v120","v100","v105","v114") PAIR("v120","v100","v106","v115") PAIR("v120","v100","v107","v112")
                                PAIR("v120","v100","v108","v117") PAIR("v120","v100","v109","v118") PAIR("v120","v100","v110","v119")) ::: CLOB);
    // C: two interleaved chains (r v100 / v101), temps v122 / v123: conflict-free
    if (M == 2) asm volatile(U8(PAIR("v122","v100","v104","v113") PAIR("v123","v101","v105","v114") PAIR("v122","v100","v106","v115") PAIR("v123","v101","v107","v112")
                                PAIR("v122","v100","v108","v117") PAIR("v123","v101","v109","v118") PAIR("v122","v100","v110","v119")) ::: CLOB);
    // D: sub sources in the same bank (x v104 bank 0, c v112 bank 0), rest conflict-free
    if (M == 3) asm volatile(U8(PAIR("v121","v102","v104","v112") PAIR("v123","v102","v108","v116") PAIR("v121","v102","v104","v112") PAIR("v123","v102","v108","v116")
                                PAIR("v121","v102","v104","v112") PAIR("v123","v102","v108","v116") PAIR("v121","v102","v104","v112")) ::: CLOB);
    // E: SGPR centres (K1 form), conflict-free temps
    if (M == 4) asm volatile(U8(PAIRS("v121","v100","v104","%0") PAIRS("v122","v100","v105","%1") PAIRS("v121","v100","v106","%2") PAIRS("v122","v100","v107","%3")
                                PAIRS("v121","v100","v108","%4") PAIRS("v122","v100","v109","%5") PAIRS("v121","v100","v110","%6"))
                             :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6) : CLOB);
    // F: SGPR centres, temp in r's bank
    if (M == 5) asm volatile(U8(PAIRS("v120","v100","v104","%0") PAIRS("v120","v100","v105","%1") PAIRS("v120","v100","v106","%2") PAIRS("v120","v100","v107","%3")
                                PAIRS("v120","v100","v108","%4") PAIRS("v120","v100","v109","%5") PAIRS("v120","v100","v110","%6"))
                             :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6) : CLOB);
    // G: all 7 subs first (7 temps), then the 7 fmacs (independent of the adjacent instruction)
    if (M == 6) asm volatile(U8("v_sub_f32_e32 v121, v104, v113\n v_sub_f32_e32 v122, v105, v114\n v_sub_f32_e32 v123, v106, v115\n v_sub_f32_e32 v125, v107, v112\n"
                                "v_sub_f32_e32 v126, v108, v117\n v_sub_f32_e32 v127, v109, v118\n v_sub_f32_e32 v124, v110, v119\n"
                                "v_fmac_f32_e32 v100, v121, v121\n v_fmac_f32_e32 v100, v122, v122\n v_fmac_f32_e32 v100, v123, v123\n v_fmac_f32_e32 v100, v125, v125\n"
                                "v_fmac_f32_e32 v100, v126, v126\n v_fmac_f32_e32 v100, v127, v127\n v_fmac_f32_e32 v101, v124, v124\n") ::: CLOB);
    // H: 14 independent conflict-free fmacs (baseline)
    if (M == 7) asm volatile(U8("v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v101, v106, v111\n v_fmac_f32_e32 v102, v107, v108\n v_fmac_f32_e32 v103, v104, v109\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v101, v106, v111\n v_fmac_f32_e32 v102, v107, v108\n v_fmac_f32_e32 v103, v104, v109\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v101, v106, v111\n v_fmac_f32_e32 v102, v107, v108\n v_fmac_f32_e32 v103, v104, v109\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v101, v106, v111\n") ::: CLOB);
    // I: v_mul d,d into fresh temp then add chain? (fma form): v_fma_f32 r, d, d, r (VOP3)
    if (M == 8) asm volatile(U8("v_sub_f32_e32 v121, v104, v113\n v_fma_f32 v100, v121, v121, v100\n v_sub_f32_e32 v122, v105, v114\n v_fma_f32 v100, v122, v122, v100\n"
                                "v_sub_f32_e32 v121, v106, v115\n v_fma_f32 v100, v121, v121, v100\n v_sub_f32_e32 v122, v107, v112\n v_fma_f32 v100, v122, v122, v100\n"
                                "v_sub_f32_e32 v121, v108, v117\n v_fma_f32 v100, v121, v121, v100\n v_sub_f32_e32 v122, v109, v118\n v_fma_f32 v100, v122, v122, v100\n"
                                "v_sub_f32_e32 v121, v110, v119\n v_fma_f32 v100, v121, v121, v100\n") ::: CLOB);
    // J: 14 dependent fmacs on ONE accumulator, conflict-free sources
    if (M == 9) asm volatile(U8("v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v100, v106, v111\n v_fmac_f32_e32 v100, v107, v109\n v_fmac_f32_e32 v100, v105, v110\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v100, v106, v111\n v_fmac_f32_e32 v100, v107, v109\n v_fmac_f32_e32 v100, v105, v110\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v100, v106, v111\n v_fmac_f32_e32 v100, v107, v109\n v_fmac_f32_e32 v100, v105, v110\n"
                                "v_fmac_f32_e32 v100, v105, v110\n v_fmac_f32_e32 v100, v106, v111\n") ::: CLOB);
    // W part: 10 v_fmac with 10 different SGPR weights, one phi VGPR (v105), accumulators v60..v69
    if (M == 10) asm volatile(U8("v_fmac_f32_e32 v60, %0, v105\n v_fmac_f32_e32 v61, %1, v105\n v_fmac_f32_e32 v62, %2, v105\n v_fmac_f32_e32 v63, %3, v105\n"
                                 "v_fmac_f32_e32 v64, %4, v105\n v_fmac_f32_e32 v65, %5, v105\n v_fmac_f32_e32 v66, %6, v105\n v_fmac_f32_e32 v67, %0, v105\n"
                                 "v_fmac_f32_e32 v68, %1, v105\n v_fmac_f32_e32 v69, %2, v105\n v_fmac_f32_e32 v60, %3, v105\n v_fmac_f32_e32 v61, %4, v105\n"
                                 "v_fmac_f32_e32 v62, %5, v105\n v_fmac_f32_e32 v63, %6, v105\n")
                              :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6) : "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
    // same with the phi VGPR in a bank different from every accumulator? (acc v60.. cover all banks) -> phi alternates
    if (M == 11) asm volatile(U8("v_fmac_f32_e32 v60, %0, v105\n v_fmac_f32_e32 v61, %1, v106\n v_fmac_f32_e32 v62, %2, v107\n v_fmac_f32_e32 v63, %3, v104\n"
                                 "v_fmac_f32_e32 v64, %4, v105\n v_fmac_f32_e32 v65, %5, v106\n v_fmac_f32_e32 v66, %6, v107\n v_fmac_f32_e32 v67, %0, v104\n"
                                 "v_fmac_f32_e32 v68, %1, v105\n v_fmac_f32_e32 v69, %2, v106\n v_fmac_f32_e32 v60, %3, v105\n v_fmac_f32_e32 v61, %4, v106\n"
                                 "v_fmac_f32_e32 v62, %5, v107\n v_fmac_f32_e32 v63, %6, v104\n")
                              :: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(s6) : "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
    // 7 v_pk_fma_f32 with SGPR-pair weights (what SLP emits): counted as 14 "instr" -> /2 for per-FMA-pair
    if (M == 12) asm volatile(U8("v_pk_fma_f32 v[60:61], v[104:105], %0, v[60:61] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[62:63], v[104:105], %1, v[62:63] op_sel_hi:[0,1,1]\n"
                                 "v_pk_fma_f32 v[64:65], v[104:105], %2, v[64:65] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[66:67], v[104:105], %0, v[66:67] op_sel_hi:[0,1,1]\n"
                                 "v_pk_fma_f32 v[68:69], v[104:105], %1, v[68:69] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[60:61], v[104:105], %2, v[60:61] op_sel_hi:[0,1,1]\n"
                                 "v_pk_fma_f32 v[62:63], v[104:105], %0, v[62:63] op_sel_hi:[0,1,1]\n")
                              :: "s"(d0), "s"(d1), "s"(d2) : "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
    // VGPR weights (v112..v118), one phi
    if (M == 13) asm volatile(U8("v_fmac_f32_e32 v60, v113, v105\n v_fmac_f32_e32 v61, v114, v105\n v_fmac_f32_e32 v62, v115, v105\n v_fmac_f32_e32 v63, v116, v105\n"
                                 "v_fmac_f32_e32 v64, v117, v105\n v_fmac_f32_e32 v65, v118, v105\n v_fmac_f32_e32 v66, v119, v105\n v_fmac_f32_e32 v67, v112, v105\n"
                                 "v_fmac_f32_e32 v68, v113, v105\n v_fmac_f32_e32 v69, v114, v105\n v_fmac_f32_e32 v60, v115, v105\n v_fmac_f32_e32 v61, v116, v105\n"
                                 "v_fmac_f32_e32 v62, v117, v105\n v_fmac_f32_e32 v63, v118, v105\n")
                              ::: "v60","v61","v62","v63","v64","v65","v66","v67","v68","v69");
  }
  float r;
  asm volatile("v_add_f32 %0, v100, v101\n v_add_f32 %0, %0, v102\n v_add_f32 %0, %0, v103\n" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int M>
void run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<M><<<blocks, 1024>>>(out, .1f, .2f, .3f, .4f, .5f, .6f, .7f); hipDeviceSynchronize();
  hipEventRecord(e0); k<M><<<blocks, 1024>>>(out, .1f, .2f, .3f, .4f, .5f, .6f, .7f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wps = blocks * 16.0 / 1024.0;
  printf("%-64s wps=%2.0f %7.3f ms  %5.2f cyc@2.4 per instr per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)ITERS * 8 * 14 * wps));
}

int main() {
  float* out; hipMalloc(&out, 512 * 1024 * 4);
  for (int blocks : {512, 256}) {
    run<7>("14 independent fmacs (baseline)", out, blocks);
    run<9>("14 dependent fmacs, one accumulator", out, blocks);
    run<0>("sub->fmac d,d one chain, banks free (VGPR c)", out, blocks);
    run<1>("sub->fmac d,d one chain, d in r's bank", out, blocks);
    run<2>("sub->fmac two interleaved chains, banks free", out, blocks);
    run<3>("sub sources same bank", out, blocks);
    run<4>("subrev SGPR c -> fmac, banks free", out, blocks);
    run<5>("subrev SGPR c -> fmac, d in r's bank", out, blocks);
    run<6>("7 subs then 7 fmacs", out, blocks);
    run<8>("sub -> v_fma (VOP3) chain", out, blocks);
    run<10>("W part: 14 fmac, SGPR weights, one phi", out, blocks);
    run<11>("W part: 14 fmac, SGPR weights, phi rotating banks", out, blocks);
    run<12>("W part: 7 pk_fma SGPR-pair (x2 = per pk instr)", out, blocks);
    run<13>("W part: 14 fmac, VGPR weights, one phi", out, blocks);
  }
  return 0;
}
