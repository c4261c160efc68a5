// Micro-benchmark (dev tool, round 3): questions the decoder's issue model left open (gfx950).
//   1. a pure v_fma_f64 stream priced by WALL CLOCK in TFLOP/s for the whole chip (sanity line for the cycle tables:
//      s_memtime cycles and the clock they are converted with are both checked against s_memrealtime, 100 MHz);
//   2. straight-line code against a loop: the same 8-byte VOP3 / float64 mix as a 256 B loop body and as 4 KB ... 128 KB
//      of straight-line code per pass (the decoder's iteration is ~28 KB of straight-line code streamed by 12 waves per CU);
//   3. a 64-bit select as two v_cndmask against the same select done with EXEC masking (two v_fma_f64 under complementary
//      masks), and what SALU instructions cost a VALU-bound stream.
//   hipcc --offload-arch=gfx950 -O2 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <set>

#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", \
             "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "s20", "s21", \
             "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "vcc", "scc", "memory"

// BODY runs `iters` times; registers v20..v51 hold finite doubles / small integers set up in front.
#define DEF_FIXED(NAME, BODY)                                                                                   \
  __global__ void k_##NAME(unsigned long long* cyc, unsigned long long* rt, unsigned* hwid, double* sink, int iters) {                  \
    unsigned long long t0, t1, q0, q1;                                                                          \
    unsigned lane = threadIdx.x & 63u;                                                                          \
    asm volatile("v_and_b32 v48, 7, %0\n v_mov_b32 v49, 3\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0x3ff00000\n"    \
                 "v_cvt_f64_u32 v[20:21], %0\n v_cvt_f64_u32 v[22:23], %0\n v_cvt_f64_u32 v[24:25], %0\n"      \
                 "v_cvt_f64_u32 v[26:27], %0\n v_cvt_f64_u32 v[28:29], %0\n v_cvt_f64_u32 v[30:31], %0\n"      \
                 "v_cvt_f64_u32 v[32:33], %0\n v_cvt_f64_u32 v[34:35], %0\n v_cvt_f64_u32 v[36:37], %0\n"      \
                 "v_cvt_f64_u32 v[38:39], %0\n v_cvt_f64_u32 v[40:41], %0\n v_cvt_f64_u32 v[42:43], %0\n"      \
                 "v_cvt_f64_u32 v[44:45], %0\n v_cvt_f64_u32 v[46:47], %0\n"                                   \
                 "s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x33333333\n s_mov_b64 s[24:25], 0\n s_mov_b64 s[26:27], 1\n" \
                 "s_mov_b32 s28, 0x80000000\n" ::"v"(lane) : CLOB);                                             \
    asm volatile("s_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0)::"memory");   \
    for (int i = 0; i < iters; ++i) asm volatile(BODY ::: CLOB);                                                \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1)::"memory"); \
    unsigned hw, xcc;                                                                                           \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = 0.0;                                                          \
    if ((threadIdx.x & 63) == 0) {                                                                              \
      cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;                                         \
      rt[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = q1 - q0;                                          \
      hwid[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (hw & 0xffffffu) | (xcc << 24);                 \
    }                                                                                                           \
  }

// ---- 1. pure float64 FMA, 8 independent accumulators
#define FMA8 "v_fma_f64 v[20:21],v[20:21],v[40:41],v[42:43]\n v_fma_f64 v[22:23],v[22:23],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[24:25],v[24:25],v[40:41],v[42:43]\n v_fma_f64 v[26:27],v[26:27],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[28:29],v[28:29],v[40:41],v[42:43]\n v_fma_f64 v[30:31],v[30:31],v[40:41],v[42:43]\n" \
             "v_fma_f64 v[32:33],v[32:33],v[40:41],v[42:43]\n v_fma_f64 v[34:35],v[34:35],v[40:41],v[42:43]\n"
DEF_FIXED(fma_f64, ".rept 4\n" FMA8 ".endr\n")
#define ADD8 "v_add_f32 v20,v20,v40\n v_add_f32 v21,v21,v40\n v_add_f32 v22,v22,v40\n v_add_f32 v23,v23,v40\n" \
             "v_add_f32 v24,v24,v40\n v_add_f32 v25,v25,v40\n v_add_f32 v26,v26,v40\n v_add_f32 v27,v27,v40\n"
DEF_FIXED(add_f32, ".rept 4\n" ADD8 ".endr\n")

// ---- 2. the decoder's kind of mix (all 8-byte encodings), as a loop body of 32 instructions and as straight-line code
#define MIX8 "v_add_f64 v[20:21],v[24:25],v[30:31]\n v_and_or_b32 v32,v36,v41,v42\n v_min_f64 v[22:23],v[26:27],v[28:29]\n" \
             "v_alignbit_b32 v33,v37,v42,31\n v_max_f64 v[34:35],v[38:39],v[40:41]\n v_and_or_b32 v43,v44,v45,v46\n"          \
             "v_add_f64 v[46:47],v[24:25],v[28:29]\n v_cndmask_b32_e64 v36,v37,v38,s[20:21]\n"
DEF_FIXED(mix_32, ".rept 4\n" MIX8 ".endr\n")          // 256 B
DEF_FIXED(mix_512, ".rept 64\n" MIX8 ".endr\n")        // 4 KB
DEF_FIXED(mix_2k, ".rept 256\n" MIX8 ".endr\n")        // 16 KB
DEF_FIXED(mix_4k, ".rept 512\n" MIX8 ".endr\n")        // 32 KB
DEF_FIXED(mix_6k, ".rept 768\n" MIX8 ".endr\n")        // 48 KB
DEF_FIXED(mix_12k, ".rept 1536\n" MIX8 ".endr\n")      // 96 KB (beyond the 64 KB instruction cache)
// ... and a 4-byte VOP2 stream (half the bytes per instruction)
#define XOR8 "v_xor_b32 v20,v20,v40\n v_xor_b32 v21,v21,v40\n v_xor_b32 v22,v22,v40\n v_xor_b32 v23,v23,v40\n" \
             "v_xor_b32 v24,v24,v40\n v_xor_b32 v25,v25,v40\n v_xor_b32 v26,v26,v40\n v_xor_b32 v27,v27,v40\n"
DEF_FIXED(xor_32, ".rept 4\n" XOR8 ".endr\n")
DEF_FIXED(xor_4k, ".rept 512\n" XOR8 ".endr\n")        // 16 KB
DEF_FIXED(xor_8k, ".rept 1024\n" XOR8 ".endr\n")       // 32 KB

// ---- 3. one "edge" of the read pass: is-it-the-argmin compare, 64-bit select of the magnitude, subtract
// (a) as the decoder does it now: compare into an SGPR pair, two v_cndmask, sign insert, v_add_f64  (5 VALU)
#define SEL_CND(T, U) "v_cmp_eq_u32 s[22:23], v48, v49\n v_and_or_b32 v51, " U ", s28, v51\n"                       \
                      "v_cndmask_b32_e64 v44, v40, v42, s[22:23]\n v_cndmask_b32_e64 v45, v41, v43, s[22:23]\n"      \
                      "v_add_f64 " T ", " T ", -v[44:45]\n"
DEF_FIXED(sel_cnd, SEL_CND("v[20:21]", "v32") SEL_CND("v[22:23]", "v33") SEL_CND("v[24:25]", "v34") SEL_CND("v[26:27]", "v35")
                       SEL_CND("v[28:29]", "v32") SEL_CND("v[30:31]", "v33") SEL_CND("v[34:35]", "v34") SEL_CND("v[36:37]", "v35"))
// (b) unit-sign FMA under complementary EXEC masks: compare, unit, 2 FMA (4 VALU) + 2 SALU
#define SEL_EXEC(T, U) "v_cmp_eq_u32 s[22:23], v48, v49\n v_and_or_b32 v51, " U ", s28, v51\n"                      \
                       "s_mov_b64 exec, s[22:23]\n v_fma_f64 " T ", -v[50:51], v[42:43], " T "\n"                    \
                       "s_not_b64 exec, exec\n v_fma_f64 " T ", -v[50:51], v[40:41], " T "\n"
DEF_FIXED(sel_exec, SEL_EXEC("v[20:21]", "v32") SEL_EXEC("v[22:23]", "v33") SEL_EXEC("v[24:25]", "v34") SEL_EXEC("v[26:27]", "v35")
                        SEL_EXEC("v[28:29]", "v32") SEL_EXEC("v[30:31]", "v33") SEL_EXEC("v[34:35]", "v34") SEL_EXEC("v[36:37]", "v35")
                        "s_mov_b64 exec, -1\n")
// (c) all lanes first, the argmin lanes again: compare, unit, FMA, masked FMA + 2 SALU
#define SEL_EXEC2(T, U) "v_cmp_eq_u32 s[22:23], v48, v49\n v_and_or_b32 v51, " U ", s28, v51\n"                     \
                        "v_fma_f64 v[44:45], -v[50:51], v[40:41], " T "\n s_mov_b64 exec, s[22:23]\n"                \
                        "v_fma_f64 v[44:45], -v[50:51], v[42:43], " T "\n s_mov_b64 exec, -1\n v_mov_b64 " T ", v[44:45]\n"
DEF_FIXED(sel_exec2, SEL_EXEC2("v[20:21]", "v32") SEL_EXEC2("v[22:23]", "v33") SEL_EXEC2("v[24:25]", "v34") SEL_EXEC2("v[26:27]", "v35")
                         SEL_EXEC2("v[28:29]", "v32") SEL_EXEC2("v[30:31]", "v33") SEL_EXEC2("v[34:35]", "v34") SEL_EXEC2("v[36:37]", "v35"))
// (d) the compares batched in front (8 SGPR pairs would be needed; here the same pair: timing only), then masks
#define SEL_EXEC3(T, U) "v_and_or_b32 v51, " U ", s28, v51\n"                                                       \
                        "s_mov_b64 exec, s[22:23]\n v_fma_f64 " T ", -v[50:51], v[42:43], " T "\n"                   \
                        "s_not_b64 exec, exec\n v_fma_f64 " T ", -v[50:51], v[40:41], " T "\n"
DEF_FIXED(sel_exec3, "s_mov_b64 exec, -1\n .rept 8\n v_cmp_eq_u32 s[22:23], v48, v49\n .endr\n"
                     SEL_EXEC3("v[20:21]", "v32") SEL_EXEC3("v[22:23]", "v33") SEL_EXEC3("v[24:25]", "v34") SEL_EXEC3("v[26:27]", "v35")
                     SEL_EXEC3("v[28:29]", "v32") SEL_EXEC3("v[30:31]", "v33") SEL_EXEC3("v[34:35]", "v34") SEL_EXEC3("v[36:37]", "v35")
                     "s_mov_b64 exec, -1\n")
// SALU beside a VALU-bound stream: 0 / 1 / 2 scalar instructions per float64 add
DEF_FIXED(f64_salu0, ".rept 4\n v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[22:23],v[22:23],v[40:41]\n v_add_f64 v[24:25],v[24:25],v[40:41]\n v_add_f64 v[26:27],v[26:27],v[40:41]\n"
                     "v_add_f64 v[28:29],v[28:29],v[40:41]\n v_add_f64 v[30:31],v[30:31],v[40:41]\n v_add_f64 v[32:33],v[32:33],v[40:41]\n v_add_f64 v[34:35],v[34:35],v[40:41]\n .endr\n")
#define AS1(T) "v_add_f64 " T "," T ",v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n"
#define AS2(T) "v_add_f64 " T "," T ",v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n s_andn2_b64 s[22:23], s[20:21], s[24:25]\n"
DEF_FIXED(f64_salu1, ".rept 4\n" AS1("v[20:21]") AS1("v[22:23]") AS1("v[24:25]") AS1("v[26:27]") AS1("v[28:29]") AS1("v[30:31]") AS1("v[32:33]") AS1("v[34:35]") ".endr\n")
DEF_FIXED(f64_salu2, ".rept 4\n" AS2("v[20:21]") AS2("v[22:23]") AS2("v[24:25]") AS2("v[26:27]") AS2("v[28:29]") AS2("v[30:31]") AS2("v[32:33]") AS2("v[34:35]") ".endr\n")
// a masked v_mov_b32 (VOP1) and v_bcnt / v_lshl_or (per-row bookkeeping of the new scheme)
DEF_FIXED(mov_b32, ".rept 4\n v_mov_b32 v20, 5\n v_mov_b32 v21, 5\n v_mov_b32 v22, 5\n v_mov_b32 v23, 5\n v_mov_b32 v24, 5\n v_mov_b32 v25, 5\n v_mov_b32 v26, 5\n v_mov_b32 v27, 5\n .endr\n")


// ---- 4. straight-line (512 x 8 instructions, no loop bias) streams of ONE class, and two-class alternations
#define SL(NAME, A8) DEF_FIXED(NAME, ".rept 512\n" A8 ".endr\n")
SL(sl_fma64, FMA8)
SL(sl_add64, "v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[22:23],v[22:23],v[40:41]\n v_add_f64 v[24:25],v[24:25],v[40:41]\n v_add_f64 v[26:27],v[26:27],v[40:41]\n"
             "v_add_f64 v[28:29],v[28:29],v[40:41]\n v_add_f64 v[30:31],v[30:31],v[40:41]\n v_add_f64 v[32:33],v[32:33],v[40:41]\n v_add_f64 v[34:35],v[34:35],v[40:41]\n")
SL(sl_min64, "v_min_f64 v[20:21],v[20:21],v[40:41]\n v_max_f64 v[22:23],v[22:23],v[40:41]\n v_min_f64 v[24:25],v[24:25],v[40:41]\n v_max_f64 v[26:27],v[26:27],v[40:41]\n"
             "v_min_f64 v[28:29],v[28:29],v[40:41]\n v_max_f64 v[30:31],v[30:31],v[40:41]\n v_min_f64 v[32:33],v[32:33],v[40:41]\n v_max_f64 v[34:35],v[34:35],v[40:41]\n")
SL(sl_andor, "v_and_or_b32 v20,v36,v41,v42\n v_and_or_b32 v21,v36,v41,v42\n v_and_or_b32 v22,v36,v41,v42\n v_and_or_b32 v23,v36,v41,v42\n"
             "v_and_or_b32 v24,v36,v41,v42\n v_and_or_b32 v25,v36,v41,v42\n v_and_or_b32 v26,v36,v41,v42\n v_and_or_b32 v27,v36,v41,v42\n")
SL(sl_cnd64, "v_cndmask_b32_e64 v20,v37,v38,s[20:21]\n v_cndmask_b32_e64 v21,v37,v38,s[20:21]\n v_cndmask_b32_e64 v22,v37,v38,s[20:21]\n v_cndmask_b32_e64 v23,v37,v38,s[20:21]\n"
             "v_cndmask_b32_e64 v24,v37,v38,s[20:21]\n v_cndmask_b32_e64 v25,v37,v38,s[20:21]\n v_cndmask_b32_e64 v26,v37,v38,s[20:21]\n v_cndmask_b32_e64 v27,v37,v38,s[20:21]\n")
SL(sl_cmp64, "v_cmp_eq_f64 s[22:23],|v[20:21]|,v[40:41]\n v_cmp_eq_f64 s[24:25],|v[22:23]|,v[40:41]\n v_cmp_eq_f64 s[22:23],|v[24:25]|,v[40:41]\n v_cmp_eq_f64 s[24:25],|v[26:27]|,v[40:41]\n"
             "v_cmp_eq_f64 s[22:23],|v[28:29]|,v[40:41]\n v_cmp_eq_f64 s[24:25],|v[30:31]|,v[40:41]\n v_cmp_eq_f64 s[22:23],|v[32:33]|,v[40:41]\n v_cmp_eq_f64 s[24:25],|v[34:35]|,v[40:41]\n")
SL(sl_xor_add64, "v_xor_b32 v20,v20,v40\n v_add_f64 v[22:23],v[22:23],v[40:41]\n v_xor_b32 v21,v21,v40\n v_add_f64 v[24:25],v[24:25],v[40:41]\n"
                 "v_xor_b32 v26,v26,v40\n v_add_f64 v[28:29],v[28:29],v[40:41]\n v_xor_b32 v27,v27,v40\n v_add_f64 v[30:31],v[30:31],v[40:41]\n")
SL(sl_xor3_add64, "v_xor_b32 v20,v20,v40\n v_xor_b32 v21,v21,v40\n v_xor_b32 v26,v26,v40\n v_add_f64 v[22:23],v[22:23],v[40:41]\n"
                  "v_xor_b32 v27,v27,v40\n v_xor_b32 v32,v32,v40\n v_xor_b32 v33,v33,v40\n v_add_f64 v[28:29],v[28:29],v[40:41]\n")
SL(sl_add64_salu, "v_add_f64 v[20:21],v[20:21],v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n v_add_f64 v[22:23],v[22:23],v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n"
                  "v_add_f64 v[24:25],v[24:25],v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n v_add_f64 v[26:27],v[26:27],v[40:41]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n")
SL(sl_add64_nop, "v_add_f64 v[20:21],v[20:21],v[40:41]\n s_nop 0\n v_add_f64 v[22:23],v[22:23],v[40:41]\n s_nop 0\n"
                 "v_add_f64 v[24:25],v[24:25],v[40:41]\n s_nop 0\n v_add_f64 v[26:27],v[26:27],v[40:41]\n s_nop 0\n")
SL(sl_dep_add64, "v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n"
                 "v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n v_add_f64 v[20:21],v[20:21],v[40:41]\n")
SL(sl_dep_chain, "v_cndmask_b32_e64 v44,v40,v42,s[20:21]\n v_cndmask_b32_e64 v45,v41,v43,s[20:21]\n v_and_or_b32 v45,v32,s28,v45\n v_add_f64 v[20:21],v[20:21],v[44:45]\n"
                 "v_cndmask_b32_e64 v46,v40,v42,s[20:21]\n v_cndmask_b32_e64 v47,v41,v43,s[20:21]\n v_and_or_b32 v47,v33,s28,v47\n v_add_f64 v[22:23],v[22:23],v[46:47]\n")
SL(sl_hi_regs, "v_add_f64 v[20:21],v[24:25],v[30:31]\n v_and_or_b32 v32,v36,v41,v42\n v_min_f64 v[22:23],v[26:27],v[28:29]\n"
               "v_alignbit_b32 v33,v37,v42,31\n v_max_f64 v[34:35],v[38:39],v[40:41]\n v_and_or_b32 v43,v44,v45,v46\n"
               "v_add_f64 v[46:47],v[24:25],v[28:29]\n v_cndmask_b32_e64 v36,v37,v38,s[20:21]\n")

SL(sl_fmac64_sgpr, "v_fmac_f64_e32 v[20:21],s[20:21],v[40:41]\n v_fmac_f64_e32 v[22:23],s[20:21],v[42:43]\n v_fmac_f64_e32 v[24:25],s[24:25],v[40:41]\n v_fmac_f64_e32 v[26:27],s[24:25],v[42:43]\n"
                     "v_fmac_f64_e32 v[28:29],s[20:21],v[44:45]\n v_fmac_f64_e32 v[30:31],s[20:21],v[46:47]\n v_fmac_f64_e32 v[32:33],s[24:25],v[44:45]\n v_fmac_f64_e32 v[34:35],s[24:25],v[46:47]\n")
SL(sl_fmac64_vgpr, "v_fmac_f64_e32 v[20:21],v[36:37],v[40:41]\n v_fmac_f64_e32 v[22:23],v[36:37],v[42:43]\n v_fmac_f64_e32 v[24:25],v[38:39],v[40:41]\n v_fmac_f64_e32 v[26:27],v[38:39],v[42:43]\n"
                     "v_fmac_f64_e32 v[28:29],v[36:37],v[44:45]\n v_fmac_f64_e32 v[30:31],v[36:37],v[46:47]\n v_fmac_f64_e32 v[32:33],v[38:39],v[44:45]\n v_fmac_f64_e32 v[34:35],v[38:39],v[46:47]\n")

// clock check: s_memtime against s_memrealtime (100 MHz)
__global__ void k_clock(unsigned long long* out, int iters) {
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) asm volatile(".rept 32\n v_fma_f64 %0, %0, %0, %0\n .endr\n" : "+v"(a));
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (a == 12345.678) out[2] = 1;
}

typedef void (*kern_t)(unsigned long long*, unsigned long long*, unsigned*, double*, int);
struct Case { const char* name; kern_t k; int per_iter; int unit; const char* what; };

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
  unsigned long long* cyc;
  unsigned* hwid;
  unsigned long long* rt;
  double* sink;
  hipMalloc(&cyc, sizeof(unsigned long long) * ncu * 64);
  hipMalloc(&hwid, sizeof(unsigned) * ncu * 64);
  hipMalloc(&rt, sizeof(unsigned long long) * ncu * 64);
  hipMalloc(&sink, sizeof(double) * ncu * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, cyc, 200000);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("clock check: s_memtime advanced %llu while s_memrealtime (100 MHz) advanced %llu  =>  s_memtime runs at %.1f MHz\n", h[0], h[1],
           (double)h[0] / ((double)h[1] / 100.0));
  }
  Case cases[] = {
#define C(N, P, U, W) {#N, k_##N, P, U, W}
      C(fma_f64, 32, 1, "pure v_fma_f64"), C(add_f32, 32, 1, "pure v_add_f32"),
      C(mix_32, 32, 1, "f64/VOP3 mix, 256 B loop"), C(mix_512, 512, 1, "same mix, 4 KB straight-line"), C(mix_2k, 2048, 1, "16 KB"),
      C(mix_4k, 4096, 1, "32 KB"), C(mix_6k, 6144, 1, "48 KB"), C(mix_12k, 12288, 1, "96 KB"),
      C(xor_32, 32, 1, "VOP2 xor, 128 B loop"), C(xor_4k, 4096, 1, "VOP2 xor 16 KB"), C(xor_8k, 8192, 1, "VOP2 xor 32 KB"),
      C(sel_cnd, 8, 8, "per edge: cmp + unit + 2 cndmask + add_f64"), C(sel_exec, 8, 8, "per edge: cmp + unit + 2 fma under EXEC/~EXEC"),
      C(sel_exec2, 8, 8, "per edge: cmp + unit + fma + masked fma + mov_b64"), C(sel_exec3, 8, 8, "per edge: 8 cmps batched + unit + 2 masked fma"),
      C(f64_salu0, 32, 1, "add_f64"), C(f64_salu1, 32, 1, "add_f64 + 1 SALU"), C(f64_salu2, 32, 1, "add_f64 + 2 SALU"), C(mov_b32, 32, 1, "v_mov_b32 imm"),
      C(sl_fma64, 4096, 1, "straight-line v_fma_f64"), C(sl_add64, 4096, 1, "straight-line v_add_f64"), C(sl_min64, 4096, 1, "straight-line v_min/max_f64"),
      C(sl_andor, 4096, 1, "straight-line v_and_or_b32"), C(sl_cnd64, 4096, 1, "straight-line v_cndmask_e64 sgpr"), C(sl_cmp64, 4096, 1, "straight-line v_cmp_eq_f64 -> sgpr"),
      C(sl_xor_add64, 4096, 1, "straight-line xor / add_f64 alternating"), C(sl_xor3_add64, 4096, 1, "straight-line 3 xor : 1 add_f64"),
      C(sl_add64_salu, 4096, 1, "straight-line add_f64 / s_or alternating (per instruction)"), C(sl_add64_nop, 4096, 1, "straight-line add_f64 / s_nop alternating"),
      C(sl_fmac64_sgpr, 4096, 1, "straight-line v_fmac_f64 with an SGPR-pair multiplier"), C(sl_fmac64_vgpr, 4096, 1, "straight-line v_fmac_f64, all VGPR"),
      C(sl_dep_add64, 4096, 1, "straight-line dependent add_f64 chain"), C(sl_dep_chain, 4096, 1, "straight-line 2cnd->and_or->add_f64 chains"),
  };
  printf("%-12s %8s %8s %8s %8s   cycles (s_memtime) per unit per SIMD at W waves/SIMD | wall-clock at W=3\n", "case", "W=1", "W=2", "W=3", "W=4");
  for (auto& c : cases) {
    printf("%-12s", c.name);
    double wall3 = 0, mhz3 = 0;
    size_t ncu_seen = 0;
    const long total = 4L * 1024 * 1024;                 // instructions (units) per wave
    const int iters = (int)(total / c.per_iter / (c.unit == 8 ? 6 : 1));   // (an 'edge' unit is ~6 instructions)
    for (int W = 1; W <= 4; ++W) {
      const int threads = 256 * W;
      hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, rt, hwid, sink, 2);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, rt, hwid, sink, iters);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(ncu * 4 * W);
      std::vector<unsigned> hw(ncu * 4 * W);
      hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      hipMemcpy(hw.data(), hwid, hw.size() * 4, hipMemcpyDeviceToHost);
      if (!strcmp(c.name, "sl_add64") || !strcmp(c.name, "sl_xor_add64") || !strcmp(c.name, "sl_fmac64_sgpr") || !strcmp(c.name, "sl_fmac64_vgpr")) {      // per-wave view: is the issue bandwidth shared evenly?
        fprintf(stderr, "%s W=%d, workgroup 0: cycles per instruction of each wave [simd.waveslot]:", c.name, W);
        for (int i = 0; i < 4 * W; ++i)
          fprintf(stderr, " %u.%u:%.2f", (hw[i] >> 4) & 3u, hw[i] & 15u, (double)h[i] / ((double)iters * c.per_iter));
        std::vector<unsigned long long> hs(h);
        std::sort(hs.begin(), hs.end());
        fprintf(stderr, "   | all waves: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f\n", (double)hs[0] / ((double)iters * c.per_iter),
                (double)hs[hs.size() / 4] / ((double)iters * c.per_iter), (double)hs[hs.size() / 2] / ((double)iters * c.per_iter),
                (double)hs[3 * hs.size() / 4] / ((double)iters * c.per_iter), (double)hs.back() / ((double)iters * c.per_iter));
      }
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      printf(" %8.2f", med / ((double)iters * c.per_iter * W));
      if (W == 3) {
        std::vector<unsigned long long> hr(ncu * 4 * W);
        hipMemcpy(hr.data(), rt, hr.size() * 8, hipMemcpyDeviceToHost);
        std::sort(hr.begin(), hr.end());
        mhz3 = med / ((double)hr[hr.size() / 2] / 100.0);
        wall3 = ms;
        std::set<unsigned> cus;
        for (unsigned x : hw) cus.insert(((x >> 8) & 0xffu) | ((x >> 24) << 8));   // cu_id, sh_id, se_id + xcc_id
        ncu_seen = cus.size();
      }
    }
    const double units = (double)iters * c.per_iter * 12.0 * ncu;       // wave-level units executed at W=3
    printf("   | %.3f ms, s_memtime/s_memrealtime = %.0f MHz", wall3, mhz3);
    if (!strcmp(c.name, "fma_f64")) printf(", %.1f TFLOP/s (datasheet vector f64: 78.6)", units * 64 * 2 / (wall3 * 1e-3) * 1e-12);
    if (!strcmp(c.name, "add_f32")) printf(", %.1f T lane-add/s", units * 64 / (wall3 * 1e-3) * 1e-12);
    printf("  [%zu distinct (xcc,se,sh,cu)]  %s\n", ncu_seen, c.what);
  }
  // sustained float64 FMA rate of the whole chip by wall clock (a quarter of a second per point: past the clock ramp)
  for (int W = 2; W <= 4; ++W) {
    const int iters = 20000, threads = 256 * W;
    hipLaunchKernelGGL(k_sl_fma64, dim3(ncu), dim3(threads), 0, 0, cyc, rt, hwid, sink, 200);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_sl_fma64, dim3(ncu), dim3(threads), 0, 0, cyc, rt, hwid, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu * 4 * W), hr(ncu * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), rt, hr.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    std::sort(hr.begin(), hr.end());
    const double instr = (double)iters * 4096 * 4 * W * ncu;
    printf("sustained v_fma_f64, %d waves/SIMD on %d CUs: %.1f ms wall clock => %.1f TFLOP/s (datasheet vector f64 78.6); %.2f s_memtime cycles per "
           "instruction per SIMD; s_memtime/s_memrealtime = %.0f MHz under this load\n", W, ncu, ms, instr * 128 / (ms * 1e-3) * 1e-12,
           (double)h[h.size() / 2] / ((double)iters * 4096 * W), (double)h[h.size() / 2] / ((double)hr[hr.size() / 2] / 100.0));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return 0;
}
