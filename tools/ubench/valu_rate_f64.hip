// Micro-benchmark (dev tool): issue rate of the float64 VALU / LDS instructions the exact LDPC decoder is made of (gfx950).
// Same method as valu_rate.hip: one workgroup per CU with W waves per SIMD, a long stream of independent instructions
// of one kind between two s_memtime stamps.  Prints cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate_f64 valu_rate_f64.hip && ./valu_rate_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define DEF_KERNEL(NAME, ASM8)                                                                              \
  __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters) {                              \
    __shared__ double lds[2048];                                                                            \
    double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,  \
           r7 = r0 + 7;                                                                                     \
    double s = 1.0001 + blockIdx.x * 1e-7, u = 0.5;                                                         \
    unsigned a = (threadIdx.x & 255) * 8;                                                                   \
    lds[threadIdx.x & 2047] = r0;                                                                           \
    __syncthreads();                                                                                        \
    unsigned long long t0, t1;                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                             \
    for (int i = 0; i < iters; ++i) {                                                                       \
      asm volatile(ASM8 ASM8 ASM8 ASM8                                                                      \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)         \
                   : "v"(s), "v"(u), "v"(a)                                                                 \
                   : "vcc", "s20", "s21", "s22", "memory");                                                 \
    }                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");     \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + lds[threadIdx.x & 2047]; \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;         \
  }

#define OP2(OP) OP " %0,%0,%8\n " OP " %1,%1,%8\n " OP " %2,%2,%8\n " OP " %3,%3,%8\n " OP " %4,%4,%8\n " OP " %5,%5,%8\n " OP " %6,%6,%8\n " OP " %7,%7,%8\n "
DEF_KERNEL(add_f64, OP2("v_add_f64"))
DEF_KERNEL(mul_f64, OP2("v_mul_f64"))
DEF_KERNEL(min_f64, OP2("v_min_f64"))
DEF_KERNEL(max_f64, OP2("v_max_f64"))
DEF_KERNEL(add_f64_abs, "v_add_f64 %0,|%0|,%8\n v_add_f64 %1,|%1|,%8\n v_add_f64 %2,|%2|,%8\n v_add_f64 %3,|%3|,%8\n v_add_f64 %4,|%4|,%8\n v_add_f64 %5,|%5|,%8\n v_add_f64 %6,|%6|,%8\n v_add_f64 %7,|%7|,%8\n ")
DEF_KERNEL(min_f64_abs, "v_min_f64 %0,|%0|,%8\n v_min_f64 %1,|%1|,%8\n v_min_f64 %2,|%2|,%8\n v_min_f64 %3,|%3|,%8\n v_min_f64 %4,|%4|,%8\n v_min_f64 %5,|%5|,%8\n v_min_f64 %6,|%6|,%8\n v_min_f64 %7,|%7|,%8\n ")
DEF_KERNEL(fma_f64, "v_fma_f64 %0,%0,%8,%9\n v_fma_f64 %1,%1,%8,%9\n v_fma_f64 %2,%2,%8,%9\n v_fma_f64 %3,%3,%8,%9\n v_fma_f64 %4,%4,%8,%9\n v_fma_f64 %5,%5,%8,%9\n v_fma_f64 %6,%6,%8,%9\n v_fma_f64 %7,%7,%8,%9\n ")
DEF_KERNEL(cmp_eq_f64_sgpr, "v_cmp_eq_f64 s[20:21],%0,%8\n v_cmp_eq_f64 s[20:21],%1,%8\n v_cmp_eq_f64 s[20:21],%2,%8\n v_cmp_eq_f64 s[20:21],%3,%8\n v_cmp_eq_f64 s[20:21],%4,%8\n v_cmp_eq_f64 s[20:21],%5,%8\n v_cmp_eq_f64 s[20:21],%6,%8\n v_cmp_eq_f64 s[20:21],%7,%8\n ")
DEF_KERNEL(cmp_lt_f64_vcc, "v_cmp_lt_f64 vcc,%0,%8\n v_cmp_lt_f64 vcc,%1,%8\n v_cmp_lt_f64 vcc,%2,%8\n v_cmp_lt_f64 vcc,%3,%8\n v_cmp_lt_f64 vcc,%4,%8\n v_cmp_lt_f64 vcc,%5,%8\n v_cmp_lt_f64 vcc,%6,%8\n v_cmp_lt_f64 vcc,%7,%8\n ")
DEF_KERNEL(cmp_eq_u64_sgpr, "v_cmp_eq_u64 s[20:21],%0,%8\n v_cmp_eq_u64 s[20:21],%1,%8\n v_cmp_eq_u64 s[20:21],%2,%8\n v_cmp_eq_u64 s[20:21],%3,%8\n v_cmp_eq_u64 s[20:21],%4,%8\n v_cmp_eq_u64 s[20:21],%5,%8\n v_cmp_eq_u64 s[20:21],%6,%8\n v_cmp_eq_u64 s[20:21],%7,%8\n ")
DEF_KERNEL(cmp_lt_u64_sgpr, "v_cmp_lt_u64 s[20:21],%0,%8\n v_cmp_lt_u64 s[20:21],%1,%8\n v_cmp_lt_u64 s[20:21],%2,%8\n v_cmp_lt_u64 s[20:21],%3,%8\n v_cmp_lt_u64 s[20:21],%4,%8\n v_cmp_lt_u64 s[20:21],%5,%8\n v_cmp_lt_u64 s[20:21],%6,%8\n v_cmp_lt_u64 s[20:21],%7,%8\n ")
DEF_KERNEL(lshlrev_b64, "v_lshlrev_b64 %0,1,%0\n v_lshlrev_b64 %1,1,%1\n v_lshlrev_b64 %2,1,%2\n v_lshlrev_b64 %3,1,%3\n v_lshlrev_b64 %4,1,%4\n v_lshlrev_b64 %5,1,%5\n v_lshlrev_b64 %6,1,%6\n v_lshlrev_b64 %7,1,%7\n ")
DEF_KERNEL(mov_b64, "v_mov_b64 %0,%8\n v_mov_b64 %1,%8\n v_mov_b64 %2,%8\n v_mov_b64 %3,%8\n v_mov_b64 %4,%8\n v_mov_b64 %5,%8\n v_mov_b64 %6,%8\n v_mov_b64 %7,%8\n ")
// LDS: 8-byte reads / writes at consecutive lanes (the decoder's access pattern)
DEF_KERNEL(ds_read_b64, "ds_read_b64 %0,%10\n ds_read_b64 %1,%10 offset:2048\n ds_read_b64 %2,%10 offset:4096\n ds_read_b64 %3,%10 offset:6144\n ds_read_b64 %4,%10 offset:8192\n ds_read_b64 %5,%10 offset:10240\n ds_read_b64 %6,%10 offset:12288\n ds_read_b64 %7,%10 offset:14336\n s_waitcnt lgkmcnt(0)\n ")
DEF_KERNEL(ds_write_b64, "ds_write_b64 %10,%0\n ds_write_b64 %10,%1 offset:2048\n ds_write_b64 %10,%2 offset:4096\n ds_write_b64 %10,%3 offset:6144\n ds_write_b64 %10,%4 offset:8192\n ds_write_b64 %10,%5 offset:10240\n ds_write_b64 %10,%6 offset:12288\n ds_write_b64 %10,%7 offset:14336\n s_waitcnt lgkmcnt(0)\n ")


// the decoder's 32-bit companions (selects, sign-word bookkeeping, index compares): same harness on 32-bit registers
#define DEF_KERNEL32(NAME, ASM8)                                                                            \
  __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters) {                              \
    unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,\
             r7 = r0 + 7;                                                                                   \
    unsigned s = 0x10001u + blockIdx.x, u = 0x55aa55aau;                                                    \
    unsigned long long t0, t1;                                                                              \
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b64 vcc, s[20:21]\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory", "s20", "s21", "vcc"); \
    for (int i = 0; i < iters; ++i) {                                                                       \
      asm volatile(ASM8 ASM8 ASM8 ASM8                                                                      \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)         \
                   : "v"(s), "v"(u)                                                                         \
                   : "s22", "s23", "memory");                                                               \
    }                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");     \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (double)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);          \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;         \
  }
#define R8(FMT_A, FMT_B) FMT_A "%0" FMT_B "\n " FMT_A "%1" FMT_B "\n " FMT_A "%2" FMT_B "\n " FMT_A "%3" FMT_B "\n " FMT_A "%4" FMT_B "\n " FMT_A "%5" FMT_B "\n " FMT_A "%6" FMT_B "\n " FMT_A "%7" FMT_B "\n "
DEF_KERNEL32(cndmask_vcc, "v_cndmask_b32 %0,%0,%8,vcc\n v_cndmask_b32 %1,%1,%8,vcc\n v_cndmask_b32 %2,%2,%8,vcc\n v_cndmask_b32 %3,%3,%8,vcc\n v_cndmask_b32 %4,%4,%8,vcc\n v_cndmask_b32 %5,%5,%8,vcc\n v_cndmask_b32 %6,%6,%8,vcc\n v_cndmask_b32 %7,%7,%8,vcc\n ")
DEF_KERNEL32(cndmask_sgpr, "v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %1,%1,%8,s[20:21]\n v_cndmask_b32 %2,%2,%8,s[20:21]\n v_cndmask_b32 %3,%3,%8,s[20:21]\n v_cndmask_b32 %4,%4,%8,s[20:21]\n v_cndmask_b32 %5,%5,%8,s[20:21]\n v_cndmask_b32 %6,%6,%8,s[20:21]\n v_cndmask_b32 %7,%7,%8,s[20:21]\n ")
DEF_KERNEL32(and_or_b32, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %2,%2,%8,%9\n v_and_or_b32 %3,%3,%8,%9\n v_and_or_b32 %4,%4,%8,%9\n v_and_or_b32 %5,%5,%8,%9\n v_and_or_b32 %6,%6,%8,%9\n v_and_or_b32 %7,%7,%8,%9\n ")
DEF_KERNEL32(xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %2,%2,%8\n v_xor_b32 %3,%3,%8\n v_xor_b32 %4,%4,%8\n v_xor_b32 %5,%5,%8\n v_xor_b32 %6,%6,%8\n v_xor_b32 %7,%7,%8\n ")
DEF_KERNEL32(alignbit_b32, "v_alignbit_b32 %0,%0,%8,7\n v_alignbit_b32 %1,%1,%8,7\n v_alignbit_b32 %2,%2,%8,7\n v_alignbit_b32 %3,%3,%8,7\n v_alignbit_b32 %4,%4,%8,7\n v_alignbit_b32 %5,%5,%8,7\n v_alignbit_b32 %6,%6,%8,7\n v_alignbit_b32 %7,%7,%8,7\n ")
DEF_KERNEL32(add_u32, "v_add_u32 %0,%0,%8\n v_add_u32 %1,%1,%8\n v_add_u32 %2,%2,%8\n v_add_u32 %3,%3,%8\n v_add_u32 %4,%4,%8\n v_add_u32 %5,%5,%8\n v_add_u32 %6,%6,%8\n v_add_u32 %7,%7,%8\n ")
DEF_KERNEL32(cmp_eq_u32_vcc, "v_cmp_eq_u32 vcc,%0,%8\n v_cmp_eq_u32 vcc,%1,%8\n v_cmp_eq_u32 vcc,%2,%8\n v_cmp_eq_u32 vcc,%3,%8\n v_cmp_eq_u32 vcc,%4,%8\n v_cmp_eq_u32 vcc,%5,%8\n v_cmp_eq_u32 vcc,%6,%8\n v_cmp_eq_u32 vcc,%7,%8\n ")
DEF_KERNEL32(cmp_eq_u32_sgpr, "v_cmp_eq_u32 s[22:23],%0,%8\n v_cmp_eq_u32 s[22:23],%1,%8\n v_cmp_eq_u32 s[22:23],%2,%8\n v_cmp_eq_u32 s[22:23],%3,%8\n v_cmp_eq_u32 s[22:23],%4,%8\n v_cmp_eq_u32 s[22:23],%5,%8\n v_cmp_eq_u32 s[22:23],%6,%8\n v_cmp_eq_u32 s[22:23],%7,%8\n ")
DEF_KERNEL32(bfi_b32, "v_bfi_b32 %0,%8,%0,%9\n v_bfi_b32 %1,%8,%1,%9\n v_bfi_b32 %2,%8,%2,%9\n v_bfi_b32 %3,%8,%3,%9\n v_bfi_b32 %4,%8,%4,%9\n v_bfi_b32 %5,%8,%5,%9\n v_bfi_b32 %6,%8,%6,%9\n v_bfi_b32 %7,%8,%7,%9\n ")
// a cmp writing an SGPR pair immediately consumed by a cndmask (the decoder's select idiom)
DEF_KERNEL32(cmp_then_cndmask, "v_cmp_eq_u32 s[22:23],%0,%8\n v_cndmask_b32 %1,%1,%9,s[22:23]\n v_cmp_eq_u32 s[22:23],%2,%8\n v_cndmask_b32 %3,%3,%9,s[22:23]\n v_cmp_eq_u32 s[22:23],%4,%8\n v_cndmask_b32 %5,%5,%9,s[22:23]\n v_cmp_eq_u32 s[22:23],%6,%8\n v_cndmask_b32 %7,%7,%9,s[22:23]\n ")
// is it the VCC register or the VOP2 encoding?  (a) VOP3 encoding reading vcc; (b) VOP2 with vcc written by a VALU compare
DEF_KERNEL32(cndmask_e64_vcc, "v_cndmask_b32_e64 %0,%0,%8,vcc\n v_cndmask_b32_e64 %1,%1,%8,vcc\n v_cndmask_b32_e64 %2,%2,%8,vcc\n v_cndmask_b32_e64 %3,%3,%8,vcc\n v_cndmask_b32_e64 %4,%4,%8,vcc\n v_cndmask_b32_e64 %5,%5,%8,vcc\n v_cndmask_b32_e64 %6,%6,%8,vcc\n v_cndmask_b32_e64 %7,%7,%8,vcc\n ")
DEF_KERNEL32(cmpvcc_cndmask_e32, "v_cmp_eq_u32 vcc,%0,%8\n v_cndmask_b32_e32 %1,%1,%9,vcc\n v_cmp_eq_u32 vcc,%2,%8\n v_cndmask_b32_e32 %3,%3,%9,vcc\n v_cmp_eq_u32 vcc,%4,%8\n v_cndmask_b32_e32 %5,%5,%9,vcc\n v_cmp_eq_u32 vcc,%6,%8\n v_cndmask_b32_e32 %7,%7,%9,vcc\n ")
DEF_KERNEL32(cmpvcc_2cndmask_e32, "v_cmp_eq_u32 vcc,%0,%8\n v_cndmask_b32_e32 %1,%1,%9,vcc\n v_cndmask_b32_e32 %2,%2,%9,vcc\n v_cndmask_b32_e32 %3,%3,%9,vcc\n v_cmp_eq_u32 vcc,%4,%8\n v_cndmask_b32_e32 %5,%5,%9,vcc\n v_cndmask_b32_e32 %6,%6,%9,vcc\n v_cndmask_b32_e32 %7,%7,%9,vcc\n ")
DEF_KERNEL32(cndmask_e32_mix, "v_cndmask_b32_e32 %0,%0,%8,vcc\n v_xor_b32 %1,%1,%8\n v_cndmask_b32_e32 %2,%2,%8,vcc\n v_xor_b32 %3,%3,%8\n v_cndmask_b32_e32 %4,%4,%8,vcc\n v_xor_b32 %5,%5,%8\n v_cndmask_b32_e32 %6,%6,%8,vcc\n v_xor_b32 %7,%7,%8\n ")
// dependent chains: every instruction consumes the previous result (8 per block on ONE register) -> issue-to-use latency
DEF_KERNEL(dep_add_f64, "v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %0,%0,%8\n ")
DEF_KERNEL(dep2_add_f64, "v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n ")
DEF_KERNEL(dep4_add_f64, "v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n v_add_f64 %2,%2,%8\n v_add_f64 %3,%3,%8\n v_add_f64 %0,%0,%8\n v_add_f64 %1,%1,%8\n v_add_f64 %2,%2,%8\n v_add_f64 %3,%3,%8\n ")
DEF_KERNEL32(dep_xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n ")
DEF_KERNEL32(dep_and_or_b32, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n ")
DEF_KERNEL32(dep_cndmask_sgpr, "v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %0,%0,%8,s[20:21]\n ")
// the decoder's pass-2 chain on one edge: compare -> select -> sign -> (the f64 add is in the f64 harness above)
DEF_KERNEL32(dep_cmp_cnd_andor, "v_cmp_eq_u32 vcc,%0,%8\n s_nop 0\n v_cndmask_b32_e32 %1,%1,%9,vcc\n v_and_or_b32 %0,%1,%8,%9\n v_cmp_eq_u32 vcc,%0,%8\n s_nop 0\n v_cndmask_b32_e32 %1,%1,%9,vcc\n v_and_or_b32 %0,%1,%8,%9\n ")
// VGPR bank conflicts: the same instruction stream with both sources in the same register bank (register number mod 4) and
// in different banks.  Fixed registers (declared clobbered); values are whatever the registers hold (timing only).
#define DEF_KERNEL_FIXED(NAME, ASM8)                                                                        \
  __global__ void k_##NAME(unsigned long long* cyc, double* sink, int iters) {                              \
    unsigned long long t0, t1;                                                                              \
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b64 vcc, s[20:21]\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory", "s20", "s21", "vcc"); \
    for (int i = 0; i < iters; ++i) {                                                                       \
      asm volatile(ASM8 ASM8 ASM8 ASM8 ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", \
                   "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory"); \
    }                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");     \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = 0.0;                                                      \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;         \
  }
DEF_KERNEL_FIXED(add_f64_samebank, "v_add_f64 v[20:21],v[24:25],v[28:29]\n v_add_f64 v[22:23],v[26:27],v[30:31]\n v_add_f64 v[32:33],v[36:37],v[40:41]\n v_add_f64 v[34:35],v[38:39],v[42:43]\n v_add_f64 v[20:21],v[24:25],v[28:29]\n v_add_f64 v[22:23],v[26:27],v[30:31]\n v_add_f64 v[32:33],v[36:37],v[40:41]\n v_add_f64 v[34:35],v[38:39],v[42:43]\n ")
DEF_KERNEL_FIXED(add_f64_diffbank, "v_add_f64 v[20:21],v[24:25],v[30:31]\n v_add_f64 v[22:23],v[26:27],v[28:29]\n v_add_f64 v[32:33],v[36:37],v[42:43]\n v_add_f64 v[34:35],v[38:39],v[40:41]\n v_add_f64 v[20:21],v[24:25],v[30:31]\n v_add_f64 v[22:23],v[26:27],v[28:29]\n v_add_f64 v[32:33],v[36:37],v[42:43]\n v_add_f64 v[34:35],v[38:39],v[40:41]\n ")
DEF_KERNEL_FIXED(cnd_e32_samebank, "v_cndmask_b32_e32 v20,v24,v28,vcc\n v_xor_b32 v21,v25,v29\n v_cndmask_b32_e32 v22,v26,v30,vcc\n v_xor_b32 v23,v27,v31\n v_cndmask_b32_e32 v32,v36,v40,vcc\n v_xor_b32 v33,v37,v41\n v_cndmask_b32_e32 v34,v38,v42,vcc\n v_xor_b32 v35,v39,v43\n ")
DEF_KERNEL_FIXED(cnd_e32_diffbank, "v_cndmask_b32_e32 v20,v24,v29,vcc\n v_xor_b32 v21,v25,v30\n v_cndmask_b32_e32 v22,v26,v31,vcc\n v_xor_b32 v23,v27,v28\n v_cndmask_b32_e32 v32,v36,v41,vcc\n v_xor_b32 v33,v37,v42\n v_cndmask_b32_e32 v34,v38,v43,vcc\n v_xor_b32 v35,v39,v40\n ")
DEF_KERNEL_FIXED(andor_samebank, "v_and_or_b32 v20,v24,v28,v32\n v_and_or_b32 v21,v25,v29,v33\n v_and_or_b32 v22,v26,v30,v34\n v_and_or_b32 v23,v27,v31,v35\n v_and_or_b32 v36,v40,v44,v24\n v_and_or_b32 v37,v41,v45,v25\n v_and_or_b32 v38,v42,v46,v26\n v_and_or_b32 v39,v43,v47,v27\n ")
DEF_KERNEL_FIXED(andor_diffbank, "v_and_or_b32 v20,v24,v29,v34\n v_and_or_b32 v21,v25,v30,v35\n v_and_or_b32 v22,v26,v31,v32\n v_and_or_b32 v23,v27,v28,v33\n v_and_or_b32 v36,v40,v45,v26\n v_and_or_b32 v37,v41,v46,v27\n v_and_or_b32 v38,v42,v47,v24\n v_and_or_b32 v39,v43,v44,v25\n ")
// mixes: does interleaving encodings / data types cost more than the sum of the parts?
DEF_KERNEL_FIXED(mix_f64_vop2, "v_add_f64 v[20:21],v[24:25],v[30:31]\n v_xor_b32 v32,v36,v41\n v_min_f64 v[22:23],v[26:27],v[28:29]\n v_add_u32 v33,v37,v42\n v_max_f64 v[34:35],v[38:39],v[40:41]\n v_xor_b32 v43,v44,v45\n v_add_f64 v[46:47],v[24:25],v[28:29]\n v_add_u32 v36,v37,v38\n ")
DEF_KERNEL_FIXED(mix_f64_vop3, "v_add_f64 v[20:21],v[24:25],v[30:31]\n v_and_or_b32 v32,v36,v41,v42\n v_min_f64 v[22:23],v[26:27],v[28:29]\n v_alignbit_b32 v33,v37,v42,31\n v_max_f64 v[34:35],v[38:39],v[40:41]\n v_and_or_b32 v43,v44,v45,v46\n v_add_f64 v[46:47],v[24:25],v[28:29]\n v_cndmask_b32_e64 v36,v37,v38,s[20:21]\n ")
// one edge of pass 2 as the compiler emits it (registers renamed), twice
DEF_KERNEL_FIXED(mix_pass2_edge, "v_cmp_eq_f64 vcc,|v[20:21]|,v[22:23]\n v_alignbit_b32 v24,v25,v26,31\n s_nop 0\n v_cndmask_b32_e32 v27,v28,v29,vcc\n v_cndmask_b32_e32 v30,v31,v32,vcc\n v_and_or_b32 v33,v26,s20,v27\n v_add_f64 v[34:35],v[20:21],v[32:33]\n v_cndmask_b32_e64 v36,v37,14,vcc\n v_xor_b32 v26,v38,v39\n "
                                 "v_cmp_eq_f64 vcc,|v[40:41]|,v[22:23]\n v_alignbit_b32 v25,v24,v26,31\n s_nop 0\n v_cndmask_b32_e32 v27,v28,v29,vcc\n v_cndmask_b32_e32 v30,v31,v32,vcc\n v_and_or_b32 v43,v26,s20,v27\n v_add_f64 v[44:45],v[40:41],v[42:43]\n v_cndmask_b32_e64 v37,v36,13,vcc\n v_xor_b32 v26,v38,v21\n ")

typedef void (*kern_t)(unsigned long long*, double*, int);
struct Case { const char* name; kern_t k; int per_iter; };

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
  const int iters = 1000;
  unsigned long long* cyc;
  double* sink;
  hipMalloc(&cyc, sizeof(unsigned long long) * ncu * 64);
  hipMalloc(&sink, sizeof(double) * ncu * 1024);
  Case cases[] = {
#define C(N, P) {#N, k_##N, P}
      C(add_f64, 32), C(mul_f64, 32), C(min_f64, 32), C(max_f64, 32), C(add_f64_abs, 32), C(min_f64_abs, 32), C(fma_f64, 32),
      C(cmp_eq_f64_sgpr, 32), C(cmp_lt_f64_vcc, 32), C(cmp_eq_u64_sgpr, 32), C(cmp_lt_u64_sgpr, 32), C(lshlrev_b64, 32),
      C(mov_b64, 32), C(ds_read_b64, 32), C(ds_write_b64, 32), C(cndmask_vcc, 32), C(cndmask_sgpr, 32), C(and_or_b32, 32),
      C(xor_b32, 32), C(alignbit_b32, 32), C(add_u32, 32), C(cmp_eq_u32_vcc, 32), C(cmp_eq_u32_sgpr, 32), C(bfi_b32, 32),
      C(cmp_then_cndmask, 32), C(cndmask_e64_vcc, 32), C(cmpvcc_cndmask_e32, 32), C(cmpvcc_2cndmask_e32, 32), C(cndmask_e32_mix, 32),
      C(dep_add_f64, 32), C(dep2_add_f64, 32), C(dep4_add_f64, 32), C(dep_xor_b32, 32), C(dep_and_or_b32, 32), C(dep_cndmask_sgpr, 32),
      C(dep_cmp_cnd_andor, 24), C(add_f64_samebank, 32), C(add_f64_diffbank, 32), C(cnd_e32_samebank, 32), C(cnd_e32_diffbank, 32),
      C(andor_samebank, 32), C(andor_diffbank, 32), C(mix_f64_vop2, 32), C(mix_f64_vop3, 32), C(mix_pass2_edge, 64),
  };
  printf("%-20s %8s %8s %8s %8s   cycles per wave64 instruction per SIMD at W waves/SIMD\n", "op", "W=1", "W=2", "W=3", "W=4");
  for (auto& c : cases) {
    printf("%-20s", c.name);
    for (int W = 1; W <= 4; ++W) {
      const int threads = 256 * W;
      hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, sink, 10);
      hipLaunchKernelGGL(c.k, dim3(ncu), dim3(threads), 0, 0, cyc, sink, iters);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(ncu * 4 * W);
      hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2];
      printf(" %8.2f", med / ((double)iters * c.per_iter * W));
    }
    printf("\n");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return 0;
}
