// Micro-benchmark (dev tool, round 6): VALU issue cost PER INSTRUCTION CLASS on gfx950, alone, finely interleaved with float64
// and PHASE-SEPARATED from it (runs of R 32-bit instructions, then runs of float64 instructions, 2-4 waves per SIMD, staggered
// or not) -- the question VERDICT r5 #1 left open: `roofline.peak` of the on-chip LDPC decoder assumes one wave64 VALU
// instruction per 4 cycles per SIMD whatever the class; profiles/r3_issue_probe.txt measured that only for streams that mix the
// classes inside one wave.
//   Every figure below is the time until ALL waves of a SIMD are done (max over the workgroup's waves of s_memtime cycles,
//   median over the 256 workgroups) divided by the instructions the SIMD issued: cycles per instruction per SIMD.  (The
//   median-over-waves figure of the round-3 probe under-reads when the arbiter serves the oldest wave first.)
//   hipcc --offload-arch=gfx950 -O2 -o issue_probe2 issue_probe2.hip && ./issue_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", \
             "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "s20", "s21", \
             "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "vcc", "scc", "memory"

// BODY runs `iters` times.  `stagger` > 0: wave k of a SIMD (k = wave index / 4) first spends k * stagger v_xor_b32 of its own.
#define DEF(NAME, BODY)                                                                                         \
  __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters, int stagger) {                     \
    unsigned long long t0, t1, q0, q1;                                                                                  \
    unsigned lane = threadIdx.x & 63u;                                                                          \
    asm volatile("v_and_b32 v48, 7, %0\n v_mov_b32 v49, 3\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n"    \
                 "v_cvt_f64_u32 v[20:21], %0\n v_cvt_f64_u32 v[22:23], %0\n v_cvt_f64_u32 v[24:25], %0\n"      \
                 "v_cvt_f64_u32 v[26:27], %0\n v_cvt_f64_u32 v[28:29], %0\n v_cvt_f64_u32 v[30:31], %0\n"      \
                 "v_cvt_f64_u32 v[32:33], %0\n v_cvt_f64_u32 v[34:35], %0\n v_cvt_f64_u32 v[36:37], %0\n"      \
                 "v_cvt_f64_u32 v[38:39], %0\n v_cvt_f64_u32 v[40:41], %0\n v_cvt_f64_u32 v[42:43], %0\n"      \
                 "v_cvt_f64_u32 v[44:45], %0\n v_cvt_f64_u32 v[46:47], %0\n"                                   \
                 "s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333\n s_mov_b64 s[24:25], 0\n s_mov_b64 s[26:27], 1\n" \
                 "s_mov_b32 s28, 0x80000000\n s_mov_b64 vcc, s[20:21]\n" ::"v"(lane) : CLOB);                   \
    __syncthreads();                                                                                            \
    asm volatile("s_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0)::"memory");           \
    for (int i = (int)(threadIdx.x >> 8) * stagger; i > 0; --i) asm volatile("v_xor_b32 v20,v20,v40" ::: CLOB); \
    for (int i = 0; i < iters; ++i) asm volatile(".p2align 3\n" BODY ::: CLOB);                                                \
    asm volatile("s_mov_b64 exec, -1\n s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1)::"memory"); \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = 0.0;                                                          \
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;            \
      if (threadIdx.x == 0) cyc[gridDim.x * 16 + blockIdx.x] = q1 - q0; }                                       \
  }

// ---- eight independent instructions of one class
#define FMA8 "v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[28:29],v[28:29],v[40:41],v[42:43]\n v_fma_f64 v[30:31],v[30:31],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[32:33],v[32:33],v[40:41],v[42:43]\n v_fma_f64 v[34:35],v[34:35],v[40:41],v[42:43]\n"
#define ADD64_8 "v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[22:23],v[22:23],v[40:41]\n v_add_f64 v[24:25],v[24:25],v[40:41]\n v_add_f64 v[26:27],v[26:27],v[40:41]\n" \
                "v_add_f64 v[28:29],v[28:29],v[40:41]\n v_add_f64 v[30:31],v[30:31],v[40:41]\n v_add_f64 v[32:33],v[32:33],v[40:41]\n v_add_f64 v[34:35],v[34:35],v[40:41]\n"
#define MIN64_8 "v_min_f64 v[20:21],v[20:21],v[40:41]\n v_max_f64 v[22:23],v[22:23],v[40:41]\n v_min_f64 v[24:25],v[24:25],v[40:41]\n v_max_f64 v[26:27],v[26:27],v[40:41]\n" \
                "v_min_f64 v[28:29],v[28:29],v[40:41]\n v_max_f64 v[30:31],v[30:31],v[40:41]\n v_min_f64 v[32:33],v[32:33],v[40:41]\n v_max_f64 v[34:35],v[34:35],v[40:41]\n"
#define CMPX64_8 ".rept 8\n v_cmpx_eq_f64_e32 v[40:41], v[40:41]\n .endr\n"
#define MOV64_8 "v_mov_b64 v[20:21],v[40:41]\n v_mov_b64 v[22:23],v[40:41]\n v_mov_b64 v[24:25],v[40:41]\n v_mov_b64 v[26:27],v[40:41]\n" \
                "v_mov_b64 v[28:29],v[40:41]\n v_mov_b64 v[30:31],v[40:41]\n v_mov_b64 v[32:33],v[40:41]\n v_mov_b64 v[34:35],v[40:41]\n"
#define OP2_8(OP) OP " v20,v20,v40\n " OP " v21,v21,v40\n " OP " v22,v22,v40\n " OP " v23,v23,v40\n " OP " v24,v24,v40\n " OP " v25,v25,v40\n " OP " v26,v26,v40\n " OP " v27,v27,v40\n"
#define OP3_8(OP) OP " v20,v36,v41,v42\n " OP " v21,v36,v41,v42\n " OP " v22,v36,v41,v42\n " OP " v23,v36,v41,v42\n " OP " v24,v36,v41,v42\n " OP " v25,v36,v41,v42\n " OP " v26,v36,v41,v42\n " OP " v27,v36,v41,v42\n"
#define XOR8 OP2_8("v_xor_b32_e32")
#define XOR64E_8 OP2_8("v_xor_b32_e64")
#define ADDU8 OP2_8("v_add_u32_e32")
#define ANDOR8 OP3_8("v_and_or_b32")
#define BFI8 OP3_8("v_bfi_b32")
#define LSHLOR8 OP3_8("v_lshl_or_b32")
#define ALIGN8 "v_alignbit_b32 v20,v37,v42,31\n v_alignbit_b32 v21,v37,v42,31\n v_alignbit_b32 v22,v37,v42,31\n v_alignbit_b32 v23,v37,v42,31\n" \
               "v_alignbit_b32 v24,v37,v42,31\n v_alignbit_b32 v25,v37,v42,31\n v_alignbit_b32 v26,v37,v42,31\n v_alignbit_b32 v27,v37,v42,31\n"
#define CND32_8 "v_cndmask_b32_e32 v20,v37,v38,vcc\n v_cndmask_b32_e32 v21,v37,v38,vcc\n v_cndmask_b32_e32 v22,v37,v38,vcc\n v_cndmask_b32_e32 v23,v37,v38,vcc\n" \
                "v_cndmask_b32_e32 v24,v37,v38,vcc\n v_cndmask_b32_e32 v25,v37,v38,vcc\n v_cndmask_b32_e32 v26,v37,v38,vcc\n v_cndmask_b32_e32 v27,v37,v38,vcc\n"
#define CND64_8 "v_cndmask_b32_e64 v20,v37,v38,s[20:21]\n v_cndmask_b32_e64 v21,v37,v38,s[20:21]\n v_cndmask_b32_e64 v22,v37,v38,s[20:21]\n v_cndmask_b32_e64 v23,v37,v38,s[20:21]\n" \
                "v_cndmask_b32_e64 v24,v37,v38,s[20:21]\n v_cndmask_b32_e64 v25,v37,v38,s[20:21]\n v_cndmask_b32_e64 v26,v37,v38,s[20:21]\n v_cndmask_b32_e64 v27,v37,v38,s[20:21]\n"
#define MOV8 "v_mov_b32 v20,v40\n v_mov_b32 v21,v40\n v_mov_b32 v22,v40\n v_mov_b32 v23,v40\n v_mov_b32 v24,v40\n v_mov_b32 v25,v40\n v_mov_b32 v26,v40\n v_mov_b32 v27,v40\n"
#define MOVI8 "v_mov_b32 v20,5\n v_mov_b32 v21,5\n v_mov_b32 v22,5\n v_mov_b32 v23,5\n v_mov_b32 v24,5\n v_mov_b32 v25,5\n v_mov_b32 v26,5\n v_mov_b32 v27,5\n"
#define CMPXU8 ".rept 8\n v_cmpx_eq_u32_e32 v48, v48\n .endr\n"
#define CMPU8 ".rept 8\n v_cmp_eq_u32_e32 vcc, v48, v48\n .endr\n"
#define LSHL64_8 "v_lshlrev_b64 v[20:21],1,v[20:21]\n v_lshlrev_b64 v[22:23],1,v[22:23]\n v_lshlrev_b64 v[24:25],1,v[24:25]\n v_lshlrev_b64 v[26:27],1,v[26:27]\n" \
                 "v_lshlrev_b64 v[28:29],1,v[28:29]\n v_lshlrev_b64 v[30:31],1,v[30:31]\n v_lshlrev_b64 v[32:33],1,v[32:33]\n v_lshlrev_b64 v[34:35],1,v[34:35]\n"

#define PURE(NAME, A8) DEF(NAME, ".rept 256\n" A8 ".endr\n")   // 2048 instructions per pass
PURE(fma64, FMA8) PURE(add64, ADD64_8) PURE(min64, MIN64_8) PURE(cmpx64, CMPX64_8) PURE(mov64, MOV64_8) PURE(lshl64, LSHL64_8)
PURE(xor_e32, XOR8) PURE(xor_e64, XOR64E_8) PURE(addu_e32, ADDU8) PURE(andor, ANDOR8) PURE(bfi, BFI8) PURE(lshlor, LSHLOR8)
PURE(alignbit, ALIGN8) PURE(cnd_e32, CND32_8) PURE(cnd_e64, CND64_8) PURE(mov32, MOV8) PURE(movi32, MOVI8) PURE(cmpxu32, CMPXU8) PURE(cmpu32, CMPU8)

// ---- phase-separated: R instructions of class A, then R of float64 fma; 2048 instructions per pass
#define PH(NAME, A8, R8, REP) DEF(NAME, ".rept " #REP "\n .rept " #R8 "\n" A8 ".endr\n .rept " #R8 "\n" FMA8 ".endr\n .endr\n")
PH(ph_xor_8, XOR8, 1, 128) PH(ph_xor_16, XOR8, 2, 64) PH(ph_xor_32, XOR8, 4, 32) PH(ph_xor_64, XOR8, 8, 16) PH(ph_xor_128, XOR8, 16, 8) PH(ph_xor_256, XOR8, 32, 4)
PH(ph_xor_1024, XOR8, 128, 1)
PH(ph_andor_32, ANDOR8, 4, 32) PH(ph_andor_128, ANDOR8, 16, 8)
PH(ph_cnd32_32, CND32_8, 4, 32) PH(ph_cnd32_128, CND32_8, 16, 8)
PH(ph_cnd64_32, CND64_8, 4, 32) PH(ph_cnd64_128, CND64_8, 16, 8)
PH(ph_mov_32, MOVI8, 4, 32) PH(ph_mov_128, MOVI8, 16, 8)
// finely interleaved 1 : 1
DEF(il_xor_fma, ".rept 256\n v_xor_b32 v36,v36,v40\n v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_xor_b32 v37,v37,v40\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n"
                "v_xor_b32 v38,v38,v40\n v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_xor_b32 v39,v39,v40\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n .endr\n")
// two waves' worth of work in ONE stream cannot overlap: the same phase-separated code with the f64 half replaced by s_nop
DEF(ph_xor_nop_128, ".rept 8\n .rept 16\n" XOR8 ".endr\n .rept 128\n s_nop 0\n .endr\n .endr\n")

// ---- the decoder's edge, as it is (pass 1: wrap select, unit from the sign word, fma, cmpx, fma under exec; min/max + sign
// collection; pass 2: unit, fma, cmpx_f64, fma, mov idx): 16 VALU per edge, 8 edges per pass = 128 instructions x 16
#define E_P1(T, A) "v_cndmask_b32_e64 " A ",v37,v38,s[20:21]\n v_and_or_b32 v51,v32,s28,v50\n v_add_u32_e32 v32,v32,v32\n" \
                   "v_fma_f64 " T ",-v[50:51],v[40:41]," T "\n v_cmpx_eq_u32_e32 v48,v48\n v_fma_f64 " T ",-v[50:51],v[42:43]," T "\n s_mov_b64 exec,-1\n"
#define E_MM(T) "v_min_f64 v[44:45],v[44:45],|" T "|\n v_max_f64 v[46:47],v[46:47],|" T "|\n v_min_f64 v[46:47],v[46:47],v[44:45]\n v_alignbit_b32 v33,v33,v21,31\n"
#define E_P2(T) "v_and_or_b32 v51,v34,s28,v50\n v_fma_f64 " T ",v[50:51],v[40:41]," T "\n v_cmpx_eq_f64_e32 v[40:41],v[40:41]\n" \
                "v_fma_f64 " T ",v[50:51],v[42:43]," T "\n v_mov_b32 v35,5\n s_mov_b64 exec,-1\n"
#define EDGE_NOW(T, A) E_P1(T, A) E_MM(T) E_P2(T)
DEF(edge_now, ".rept 16\n" EDGE_NOW("v[20:21]", "v36") EDGE_NOW("v[22:23]", "v37") EDGE_NOW("v[24:25]", "v38") EDGE_NOW("v[26:27]", "v39")
                           EDGE_NOW("v[28:29]", "v36") EDGE_NOW("v[30:31]", "v37") EDGE_NOW("v[24:25]", "v38") EDGE_NOW("v[26:27]", "v39") ".endr\n")
// ... and regrouped by class over 8 edges: 8 selects, 8 unit builds (+ the shift), then the float64 work, then 8 alignbit, 8 mov
#define G_SEL(A) "v_cndmask_b32_e64 " A ",v37,v38,s[20:21]\n"
#define G_UNIT(U) "v_and_or_b32 " U ",v32,s28,v50\n v_add_u32_e32 v32,v32,v32\n"
#define G_F1(T) "v_fma_f64 " T ",-v[50:51],v[40:41]," T "\n v_cmpx_eq_u32_e32 v48,v48\n v_fma_f64 " T ",-v[50:51],v[42:43]," T "\n s_mov_b64 exec,-1\n" \
                "v_min_f64 v[44:45],v[44:45],|" T "|\n v_max_f64 v[46:47],v[46:47],|" T "|\n v_min_f64 v[46:47],v[46:47],v[44:45]\n"
#define G_F2(T) "v_fma_f64 " T ",v[50:51],v[40:41]," T "\n v_cmpx_eq_f64_e32 v[40:41],v[40:41]\n v_fma_f64 " T ",v[50:51],v[42:43]," T "\n s_mov_b64 exec,-1\n"
DEF(edge_grouped, ".rept 16\n" G_SEL("v36") G_SEL("v37") G_SEL("v38") G_SEL("v39") G_SEL("v36") G_SEL("v37") G_SEL("v38") G_SEL("v39")
                  G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51")
                  G_F1("v[20:21]") G_F1("v[22:23]") G_F1("v[24:25]") G_F1("v[26:27]") G_F1("v[28:29]") G_F1("v[30:31]") G_F1("v[24:25]") G_F1("v[26:27]")
                  ".rept 8\n v_alignbit_b32 v33,v33,v21,31\n .endr\n"
                  G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51") G_UNIT("v51")
                  G_F2("v[20:21]") G_F2("v[22:23]") G_F2("v[24:25]") G_F2("v[26:27]") G_F2("v[28:29]") G_F2("v[30:31]") G_F2("v[24:25]") G_F2("v[26:27]")
                  ".rept 8\n v_mov_b32 v35,5\n .endr\n .endr\n")


// ---- round-6 follow-ups: (a) 4-byte float64 (v_fmac_f64_e32) against the 8-byte v_fma_f64: is it the encoding or the operands?
#define FMAC8 "v_fmac_f64_e32 v[20:21],v[36:37],v[40:41]\n v_fmac_f64_e32 v[22:23],v[36:37],v[42:43]\n v_fmac_f64_e32 v[24:25],v[38:39],v[40:41]\n v_fmac_f64_e32 v[26:27],v[38:39],v[42:43]\n" \
              "v_fmac_f64_e32 v[28:29],v[36:37],v[44:45]\n v_fmac_f64_e32 v[30:31],v[36:37],v[46:47]\n v_fmac_f64_e32 v[32:33],v[38:39],v[44:45]\n v_fmac_f64_e32 v[34:35],v[38:39],v[46:47]\n"
PURE(fmac64_e32, FMAC8)
DEF(fma64_4k, ".rept 512\n" FMA8 ".endr\n")      // 32 KB body (the round-3 probe's sl_fma64)
// (b) an immediate v_mov finely interleaved with float64 (the decoder's `v_mov idx, J` sits between two fma), 1:1 and 1:3
DEF(il_movi_fma, ".rept 256\n v_mov_b32 v36,5\n v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_mov_b32 v37,5\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n"
                 "v_mov_b32 v38,5\n v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_mov_b32 v39,5\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n .endr\n")
DEF(il_movi_fma3, ".rept 256\n v_mov_b32 v36,5\n v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_fma_f64 v[28:29],v[28:29],v[40:41],v[42:43]\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n"
                  "v_mov_b32 v38,5\n v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_fma_f64 v[30:31],v[30:31],v[40:41],v[42:43]\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n .endr\n")
DEF(il_addu_fma3, ".rept 256\n v_add_u32_e32 v36,v36,v36\n v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_fma_f64 v[28:29],v[28:29],v[40:41],v[42:43]\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n"
                  "v_add_u32_e32 v38,v38,v38\n v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_fma_f64 v[30:31],v[30:31],v[40:41],v[42:43]\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n .endr\n")
// (c) runs of 8 / 16 immediate moves between float64 runs of 3 x that length (the decoder's ratio if the D `v_mov idx` of a layer were one run)
DEF(ph_movi_8_24, ".rept 64\n" MOVI8 FMA8 FMA8 FMA8 ".endr\n")
DEF(ph_movi_16_48, ".rept 32\n" MOVI8 MOVI8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 ".endr\n")
DEF(ph_addu_16_48, ".rept 32\n" ADDU8 ADDU8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 ".endr\n")

// (d) INSTRUCTION ALIGNMENT: the same 8-byte instructions at addresses = 4 (mod 8) (one s_nop in front of the aligned body)
DEF(fma64_mis, "s_nop 0\n .rept 256\n" FMA8 ".endr\n s_nop 0\n")
DEF(andor_mis, "s_nop 0\n .rept 256\n" ANDOR8 ".endr\n s_nop 0\n")
// the decoder's edge with every 8-byte instruction on an 8-byte boundary: e64 encodings for the compares, the 4-byte ones in pairs
#define A_P1(T, A) "v_cndmask_b32_e64 " A ",v37,v38,s[20:21]\n v_and_or_b32 v51,v32,s28,v50\n" \
                   "v_fma_f64 " T ",-v[50:51],v[40:41]," T "\n v_cmpx_eq_u32_e64 vcc,v48,v48\n v_fma_f64 " T ",-v[50:51],v[42:43]," T "\n s_mov_b64 exec,-1\n v_add_u32_e32 v32,v32,v32\n"
#define A_P2(T) "v_and_or_b32 v51,v34,s28,v50\n v_fma_f64 " T ",v[50:51],v[40:41]," T "\n v_cmpx_eq_f64_e64 vcc,v[40:41],v[40:41]\n" \
                "v_fma_f64 " T ",v[50:51],v[42:43]," T "\n v_mov_b32 v35,5\n s_mov_b64 exec,-1\n"
#define EDGE_AL(T, A) A_P1(T, A) E_MM(T) A_P2(T)
DEF(edge_aligned, ".rept 16\n" EDGE_AL("v[20:21]", "v36") EDGE_AL("v[22:23]", "v37") EDGE_AL("v[24:25]", "v38") EDGE_AL("v[26:27]", "v39")
                               EDGE_AL("v[28:29]", "v36") EDGE_AL("v[30:31]", "v37") EDGE_AL("v[24:25]", "v38") EDGE_AL("v[26:27]", "v39") ".endr\n")
// ... and with every 8-byte instruction OFF the boundary
DEF(edge_misaligned, "s_nop 0\n .rept 16\n" EDGE_AL("v[20:21]", "v36") EDGE_AL("v[22:23]", "v37") EDGE_AL("v[24:25]", "v38") EDGE_AL("v[26:27]", "v39")
                               EDGE_AL("v[28:29]", "v36") EDGE_AL("v[30:31]", "v37") EDGE_AL("v[24:25]", "v38") EDGE_AL("v[26:27]", "v39") ".endr\n s_nop 0\n")

// (e) instructions with ONE VGPR source (the decoder's own forms): alone, and 1 : 3 beside float64
#define ANDOR1_8 "v_and_or_b32 v20,v36,s28,1.0\n v_and_or_b32 v21,v37,s28,1.0\n v_and_or_b32 v22,v38,s28,1.0\n v_and_or_b32 v23,v39,s28,1.0\n" \
                 "v_and_or_b32 v24,v36,s28,1.0\n v_and_or_b32 v25,v37,s28,1.0\n v_and_or_b32 v26,v38,s28,1.0\n v_and_or_b32 v27,v39,s28,1.0\n"
#define LSHL1_8 "v_lshlrev_b32_e32 v20,1,v20\n v_lshlrev_b32_e32 v21,1,v21\n v_lshlrev_b32_e32 v22,1,v22\n v_lshlrev_b32_e32 v23,1,v23\n" \
                "v_lshlrev_b32_e32 v24,1,v24\n v_lshlrev_b32_e32 v25,1,v25\n v_lshlrev_b32_e32 v26,1,v26\n v_lshlrev_b32_e32 v27,1,v27\n"
PURE(andor1, ANDOR1_8) PURE(lshl1, LSHL1_8)
#define IL13(NAME, A, B) DEF(NAME, ".rept 256\n" A "\n v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_fma_f64 v[28:29],v[28:29],v[40:41],v[42:43]\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n" \
                                  B "\n v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_fma_f64 v[30:31],v[30:31],v[40:41],v[42:43]\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n .endr\n")
IL13(il_andor1_fma3, "v_and_or_b32 v36,v36,s28,1.0", "v_and_or_b32 v38,v38,s28,1.0")
IL13(il_lshl1_fma3, "v_lshlrev_b32_e32 v36,1,v36", "v_lshlrev_b32_e32 v38,1,v38")
IL13(il_movs_fma3, "v_mov_b32 v36,s28", "v_mov_b32 v38,s28")
IL13(il_mov1_fma3, "v_mov_b32 v36,v37", "v_mov_b32 v38,v39")
IL13(il_align2_fma3, "v_alignbit_b32 v36,v36,v37,31", "v_alignbit_b32 v38,v38,v39,31")
IL13(il_cnd_fma3, "v_cndmask_b32_e64 v36,v37,v38,s[20:21]", "v_cndmask_b32_e64 v39,v37,v38,s[20:21]")

typedef void (*kern_t)(unsigned long long*, double*, int, int);
struct Case { const char* name; kern_t k; int per_pass; const char* what; };

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs\n", p.gcnArchName, ncu);
  unsigned long long* cyc;
  double* sink;
  hipMalloc(&cyc, sizeof(unsigned long long) * ncu * 64);
  hipMalloc(&sink, sizeof(double) * ncu * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  Case cases[] = {
#define C(N, P, W) {#N, k_##N, P, W}
      C(fma64, 2048, "v_fma_f64"), C(add64, 2048, "v_add_f64"), C(min64, 2048, "v_min/max_f64"), C(cmpx64, 2048, "v_cmpx_eq_f64_e32"),
      C(mov64, 2048, "v_mov_b64"), C(lshl64, 2048, "v_lshlrev_b64"),
      C(xor_e32, 2048, "v_xor_b32 VOP2 (4 B)"), C(xor_e64, 2048, "v_xor_b32 VOP3 (8 B)"), C(addu_e32, 2048, "v_add_u32 VOP2"),
      C(andor, 2048, "v_and_or_b32 (VOP3, 3 sources)"), C(bfi, 2048, "v_bfi_b32"), C(lshlor, 2048, "v_lshl_or_b32"), C(alignbit, 2048, "v_alignbit_b32"),
      C(cnd_e32, 2048, "v_cndmask_b32 VOP2 / vcc"), C(cnd_e64, 2048, "v_cndmask_b32 VOP3 / sgpr pair"), C(mov32, 2048, "v_mov_b32 vgpr"),
      C(movi32, 2048, "v_mov_b32 inline constant"), C(cmpxu32, 2048, "v_cmpx_eq_u32_e32"), C(cmpu32, 2048, "v_cmp_eq_u32_e32 -> vcc"),
      C(il_xor_fma, 2048, "xor / fma_f64 alternating 1:1"),
      C(ph_xor_8, 2048, "8 xor, 8 fma_f64, ..."), C(ph_xor_16, 2048, "16 / 16"), C(ph_xor_32, 2048, "32 / 32"), C(ph_xor_64, 2048, "64 / 64"),
      C(ph_xor_128, 2048, "128 / 128"), C(ph_xor_256, 2048, "256 / 256"), C(ph_xor_1024, 2048, "1024 / 1024"),
      C(ph_andor_32, 2048, "32 and_or / 32 fma_f64"), C(ph_andor_128, 2048, "128 / 128"),
      C(ph_cnd32_32, 2048, "32 cndmask_e32 / 32 fma_f64"), C(ph_cnd32_128, 2048, "128 / 128"),
      C(ph_cnd64_32, 2048, "32 cndmask_e64 / 32 fma_f64"), C(ph_cnd64_128, 2048, "128 / 128"),
      C(ph_mov_32, 2048, "32 mov imm / 32 fma_f64"), C(ph_mov_128, 2048, "128 / 128"),
      C(ph_xor_nop_128, 2048, "128 xor / 128 s_nop (units incl. the s_nop)"),
      C(fmac64_e32, 2048, "v_fmac_f64_e32 (4 B)"), C(fma64_4k, 4096, "v_fma_f64, 32 KB body"),
      C(il_movi_fma, 2048, "mov imm / fma_f64 alternating 1:1"), C(il_movi_fma3, 2048, "1 mov imm : 3 fma_f64, interleaved"),
      C(il_addu_fma3, 2048, "1 add_u32 : 3 fma_f64, interleaved"), C(ph_movi_8_24, 2048, "8 mov imm, 24 fma_f64, ..."),
      C(ph_movi_16_48, 2048, "16 mov imm, 48 fma_f64, ..."), C(ph_addu_16_48, 2048, "16 add_u32, 48 fma_f64, ..."),
      C(andor1, 2048, "v_and_or_b32 v, v, s, 1.0 (ONE VGPR source: the decoder's unit build)"), C(lshl1, 2048, "v_lshlrev_b32_e32 v, 1, v (one VGPR source)"),
      C(il_andor1_fma3, 2048, "1 and_or (one VGPR source) : 3 fma_f64"), C(il_lshl1_fma3, 2048, "1 lshlrev (one VGPR source) : 3 fma_f64"),
      C(il_movs_fma3, 2048, "1 v_mov from an SGPR : 3 fma_f64"), C(il_mov1_fma3, 2048, "1 v_mov from a VGPR : 3 fma_f64"),
      C(il_align2_fma3, 2048, "1 alignbit (two VGPR sources) : 3 fma_f64"), C(il_cnd_fma3, 2048, "1 cndmask_e64 (two VGPR sources + mask) : 3 fma_f64"),
      C(fma64_mis, 2048, "v_fma_f64 at addresses = 4 mod 8"), C(andor_mis, 2048, "v_and_or_b32 at addresses = 4 mod 8"),
      C(edge_aligned, 1920, "decoder edge, every 8-byte instruction 8-byte aligned; per VALU"),
      C(edge_misaligned, 1920, "decoder edge, every 8-byte instruction at 4 mod 8; per VALU"),
      C(edge_now, 1920, "decoder edge as it is: 15 VALU (8 f64 + 7 32-bit) + 2 SALU per edge, interleaved; per VALU"),
      C(edge_grouped, 1920, "the same instructions grouped by class over 8 edges; per VALU"),
  };
  printf("cycles per instruction per SIMD (all waves done; s_memtime); 'st' = waves of a SIMD staggered by 1/W of a phase pair\n");
  printf("%-16s %7s %7s %7s %7s | %7s %7s %7s | %9s  what\n", "case", "W=1", "W=2", "W=3", "W=4", "W=2 st", "W=3 st", "W=4 st", "ms @ W=3");
  for (auto& c : cases) {
    printf("%-16s", c.name);
    double ms3 = 0, mhz3 = 0;
    const int iters = (int)(2097152 / c.per_pass);         // 2 M instructions per wave
    for (int pass = 0; pass < 2; ++pass) {
      for (int W = (pass ? 2 : 1); W <= 4; ++W) {
        const int threads = 256 * W;
        int stagger = 0;
        if (pass) {
          int R = 0;
          const char* u = strrchr(c.name, '_');
          if (!strncmp(c.name, "ph_", 3) && u) R = atoi(u + 1);
          if (!strncmp(c.name, "edge_", 5)) R = 64;
          if (!R) { printf(" %7s", "-"); continue; }
          stagger = 2 * R / W;
        }
        hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, sink, 2, stagger);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, sink, iters, stagger);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(ncu * 4 * W), wg(ncu);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        for (int b = 0; b < ncu; ++b) wg[b] = *std::max_element(h.begin() + b * 4 * W, h.begin() + (b + 1) * 4 * W);
        std::sort(wg.begin(), wg.end());
        printf(" %7.2f", (double)wg[ncu / 2] / ((double)iters * c.per_pass * W));
        if (W == 3 && !pass) {
          ms3 = ms;
          std::vector<unsigned long long> q(ncu);
          hipMemcpy(q.data(), cyc + ncu * 16, ncu * 8, hipMemcpyDeviceToHost);
          std::sort(q.begin(), q.end());
          mhz3 = (double)wg[ncu / 2] / ((double)q[ncu / 2] / 100.0);
        }
      }
      if (!pass) printf(" |");
    }
    printf(" | %9.3f  %s\n", ms3, c.what);
    fflush(stdout);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return 0;
}
