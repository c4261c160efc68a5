// Micro-benchmark (dev tool): issue rate of the VALU instructions the LDPC decoder is made of, on gfx950.
// One workgroup per CU with W waves per SIMD; every wave runs a long stream of independent instructions of one kind
// and stamps s_memtime around it.  Prints cycles per wave64 instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL(NAME, ASM8)                                                                          \
  __global__ void k_##NAME(unsigned long long* cyc, float* sink, int iters) {                           \
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, \
          r7 = r0 + 7;                                                                                  \
    float s = 1.0001f + blockIdx.x * 1e-7f, u = 0.5f;                                                   \
    unsigned long long t0, t1;                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                         \
    for (int i = 0; i < iters; ++i) {                                                                   \
      asm volatile(ASM8 ASM8 ASM8 ASM8                                                                  \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)     \
                   : "v"(s), "v"(u)                                                                     \
                   : "vcc", "s20", "s21", "s22");                                                                \
    }                                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                         \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;     \
  }

// 8 independent instructions per ASM8
DEF_KERNEL(add_f32, "v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n v_add_f32 %4,%4,%8\n v_add_f32 %5,%5,%8\n v_add_f32 %6,%6,%8\n v_add_f32 %7,%7,%8\n")
DEF_KERNEL(sub_abs_f32, "v_sub_f32 %0,|%0|,%8\n v_sub_f32 %1,|%1|,%8\n v_sub_f32 %2,|%2|,%8\n v_sub_f32 %3,|%3|,%8\n v_sub_f32 %4,|%4|,%8\n v_sub_f32 %5,|%5|,%8\n v_sub_f32 %6,|%6|,%8\n v_sub_f32 %7,|%7|,%8\n")
DEF_KERNEL(fma_f32, "v_fma_f32 %0,%0,%8,%9\n v_fma_f32 %1,%1,%8,%9\n v_fma_f32 %2,%2,%8,%9\n v_fma_f32 %3,%3,%8,%9\n v_fma_f32 %4,%4,%8,%9\n v_fma_f32 %5,%5,%8,%9\n v_fma_f32 %6,%6,%8,%9\n v_fma_f32 %7,%7,%8,%9\n")
DEF_KERNEL(min_f32, "v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %2,%2,%8\n v_min_f32 %3,%3,%8\n v_min_f32 %4,%4,%8\n v_min_f32 %5,%5,%8\n v_min_f32 %6,%6,%8\n v_min_f32 %7,%7,%8\n")
DEF_KERNEL(med3_f32, "v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %2,%2,%8,%9\n v_med3_f32 %3,%3,%8,%9\n v_med3_f32 %4,%4,%8,%9\n v_med3_f32 %5,%5,%8,%9\n v_med3_f32 %6,%6,%8,%9\n v_med3_f32 %7,%7,%8,%9\n")
DEF_KERNEL(min3_f32, "v_min3_f32 %0,%0,%8,%9\n v_min3_f32 %1,%1,%8,%9\n v_min3_f32 %2,%2,%8,%9\n v_min3_f32 %3,%3,%8,%9\n v_min3_f32 %4,%4,%8,%9\n v_min3_f32 %5,%5,%8,%9\n v_min3_f32 %6,%6,%8,%9\n v_min3_f32 %7,%7,%8,%9\n")
DEF_KERNEL(xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %2,%2,%8\n v_xor_b32 %3,%3,%8\n v_xor_b32 %4,%4,%8\n v_xor_b32 %5,%5,%8\n v_xor_b32 %6,%6,%8\n v_xor_b32 %7,%7,%8\n")
DEF_KERNEL(bfi_b32, "v_bfi_b32 %0,%8,%0,%9\n v_bfi_b32 %1,%8,%1,%9\n v_bfi_b32 %2,%8,%2,%9\n v_bfi_b32 %3,%8,%3,%9\n v_bfi_b32 %4,%8,%4,%9\n v_bfi_b32 %5,%8,%5,%9\n v_bfi_b32 %6,%8,%6,%9\n v_bfi_b32 %7,%8,%7,%9\n")
DEF_KERNEL(alignbit, "v_alignbit_b32 %0,%0,%8,31\n v_alignbit_b32 %1,%1,%8,31\n v_alignbit_b32 %2,%2,%8,31\n v_alignbit_b32 %3,%3,%8,31\n v_alignbit_b32 %4,%4,%8,31\n v_alignbit_b32 %5,%5,%8,31\n v_alignbit_b32 %6,%6,%8,31\n v_alignbit_b32 %7,%7,%8,31\n")
DEF_KERNEL(lshlrev, "v_lshlrev_b32 %0,3,%0\n v_lshlrev_b32 %1,3,%1\n v_lshlrev_b32 %2,3,%2\n v_lshlrev_b32 %3,3,%3\n v_lshlrev_b32 %4,3,%4\n v_lshlrev_b32 %5,3,%5\n v_lshlrev_b32 %6,3,%6\n v_lshlrev_b32 %7,3,%7\n")
DEF_KERNEL(and_or, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %2,%2,%8,%9\n v_and_or_b32 %3,%3,%8,%9\n v_and_or_b32 %4,%4,%8,%9\n v_and_or_b32 %5,%5,%8,%9\n v_and_or_b32 %6,%6,%8,%9\n v_and_or_b32 %7,%7,%8,%9\n")
DEF_KERNEL(lshl_or, "v_lshl_or_b32 %0,%0,1,%9\n v_lshl_or_b32 %1,%1,1,%9\n v_lshl_or_b32 %2,%2,1,%9\n v_lshl_or_b32 %3,%3,1,%9\n v_lshl_or_b32 %4,%4,1,%9\n v_lshl_or_b32 %5,%5,1,%9\n v_lshl_or_b32 %6,%6,1,%9\n v_lshl_or_b32 %7,%7,1,%9\n")
DEF_KERNEL(cndmask_vcc, "v_cndmask_b32 %0,%0,%8,vcc\n v_cndmask_b32 %1,%1,%8,vcc\n v_cndmask_b32 %2,%2,%8,vcc\n v_cndmask_b32 %3,%3,%8,vcc\n v_cndmask_b32 %4,%4,%8,vcc\n v_cndmask_b32 %5,%5,%8,vcc\n v_cndmask_b32 %6,%6,%8,vcc\n v_cndmask_b32 %7,%7,%8,vcc\n")
DEF_KERNEL(cmp_vcc, "v_cmp_lt_f32 vcc,%0,%8\n v_cmp_lt_f32 vcc,%1,%8\n v_cmp_lt_f32 vcc,%2,%8\n v_cmp_lt_f32 vcc,%3,%8\n v_cmp_lt_f32 vcc,%4,%8\n v_cmp_lt_f32 vcc,%5,%8\n v_cmp_lt_f32 vcc,%6,%8\n v_cmp_lt_f32 vcc,%7,%8\n")
DEF_KERNEL(cmp_sgpr, "v_cmp_lt_f32 s[20:21],%0,%8\n v_cmp_lt_f32 s[20:21],%1,%8\n v_cmp_lt_f32 s[20:21],%2,%8\n v_cmp_lt_f32 s[20:21],%3,%8\n v_cmp_lt_f32 s[20:21],%4,%8\n v_cmp_lt_f32 s[20:21],%5,%8\n v_cmp_lt_f32 s[20:21],%6,%8\n v_cmp_lt_f32 s[20:21],%7,%8\n")
DEF_KERNEL(cmp_cnd_pair, "v_cmp_lt_f32 vcc,%0,%8\n v_cndmask_b32 %0,%0,%9,vcc\n v_cmp_lt_f32 vcc,%1,%8\n v_cndmask_b32 %1,%1,%9,vcc\n v_cmp_lt_f32 vcc,%2,%8\n v_cndmask_b32 %2,%2,%9,vcc\n v_cmp_lt_f32 vcc,%3,%8\n v_cndmask_b32 %3,%3,%9,vcc\n")
DEF_KERNEL(cmp_cnd_sgpr_pair, "v_cmp_lt_f32 s[20:21],%0,%8\n v_cndmask_b32 %0,%0,%9,s[20:21]\n v_cmp_lt_f32 s[20:21],%1,%8\n v_cndmask_b32 %1,%1,%9,s[20:21]\n v_cmp_lt_f32 s[20:21],%2,%8\n v_cndmask_b32 %2,%2,%9,s[20:21]\n v_cmp_lt_f32 s[20:21],%3,%8\n v_cndmask_b32 %3,%3,%9,s[20:21]\n")
DEF_KERNEL(mov_b32, "v_mov_b32 %0,%8\n v_mov_b32 %1,%8\n v_mov_b32 %2,%8\n v_mov_b32 %3,%8\n v_mov_b32 %4,%8\n v_mov_b32 %5,%8\n v_mov_b32 %6,%8\n v_mov_b32 %7,%8\n")
DEF_KERNEL(pk_add_f16, "v_pk_add_f16 %0,%0,%8\n v_pk_add_f16 %1,%1,%8\n v_pk_add_f16 %2,%2,%8\n v_pk_add_f16 %3,%3,%8\n v_pk_add_f16 %4,%4,%8\n v_pk_add_f16 %5,%5,%8\n v_pk_add_f16 %6,%6,%8\n v_pk_add_f16 %7,%7,%8\n")
DEF_KERNEL(pk_min_f16, "v_pk_min_f16 %0,%0,%8\n v_pk_min_f16 %1,%1,%8\n v_pk_min_f16 %2,%2,%8\n v_pk_min_f16 %3,%3,%8\n v_pk_min_f16 %4,%4,%8\n v_pk_min_f16 %5,%5,%8\n v_pk_min_f16 %6,%6,%8\n v_pk_min_f16 %7,%7,%8\n")
DEF_KERNEL(pk_min_i16, "v_pk_min_i16 %0,%0,%8\n v_pk_min_i16 %1,%1,%8\n v_pk_min_i16 %2,%2,%8\n v_pk_min_i16 %3,%3,%8\n v_pk_min_i16 %4,%4,%8\n v_pk_min_i16 %5,%5,%8\n v_pk_min_i16 %6,%6,%8\n v_pk_min_i16 %7,%7,%8\n")
DEF_KERNEL(pk_sub_i16, "v_pk_sub_i16 %0,%0,%8\n v_pk_sub_i16 %1,%1,%8\n v_pk_sub_i16 %2,%2,%8\n v_pk_sub_i16 %3,%3,%8\n v_pk_sub_i16 %4,%4,%8\n v_pk_sub_i16 %5,%5,%8\n v_pk_sub_i16 %6,%6,%8\n v_pk_sub_i16 %7,%7,%8\n")
DEF_KERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0,%0,%8\n v_pk_mul_lo_u16 %1,%1,%8\n v_pk_mul_lo_u16 %2,%2,%8\n v_pk_mul_lo_u16 %3,%3,%8\n v_pk_mul_lo_u16 %4,%4,%8\n v_pk_mul_lo_u16 %5,%5,%8\n v_pk_mul_lo_u16 %6,%6,%8\n v_pk_mul_lo_u16 %7,%7,%8\n")
DEF_KERNEL(pk_ashrrev_i16, "v_pk_ashrrev_i16 %0,15,%0\n v_pk_ashrrev_i16 %1,15,%1\n v_pk_ashrrev_i16 %2,15,%2\n v_pk_ashrrev_i16 %3,15,%3\n v_pk_ashrrev_i16 %4,15,%4\n v_pk_ashrrev_i16 %5,15,%5\n v_pk_ashrrev_i16 %6,15,%6\n v_pk_ashrrev_i16 %7,15,%7\n")
DEF_KERNEL(perm_b32, "v_perm_b32 %0,%0,%8,%9\n v_perm_b32 %1,%1,%8,%9\n v_perm_b32 %2,%2,%8,%9\n v_perm_b32 %3,%3,%8,%9\n v_perm_b32 %4,%4,%8,%9\n v_perm_b32 %5,%5,%8,%9\n v_perm_b32 %6,%6,%8,%9\n v_perm_b32 %7,%7,%8,%9\n")
DEF_KERNEL(sub_u32, "v_sub_u32 %0,%0,%8\n v_sub_u32 %1,%1,%8\n v_sub_u32 %2,%2,%8\n v_sub_u32 %3,%3,%8\n v_sub_u32 %4,%4,%8\n v_sub_u32 %5,%5,%8\n v_sub_u32 %6,%6,%8\n v_sub_u32 %7,%7,%8\n")
DEF_KERNEL(min_u32, "v_min_u32 %0,%0,%8\n v_min_u32 %1,%1,%8\n v_min_u32 %2,%2,%8\n v_min_u32 %3,%3,%8\n v_min_u32 %4,%4,%8\n v_min_u32 %5,%5,%8\n v_min_u32 %6,%6,%8\n v_min_u32 %7,%7,%8\n")

DEF_KERNEL(dep1_add_f32, "v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %0,%0,%8\n ")
DEF_KERNEL(dep2_add_f32, "v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n ")
DEF_KERNEL(dep4_add_f32, "v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n v_add_f32 %0,%0,%8\n v_add_f32 %1,%1,%8\n v_add_f32 %2,%2,%8\n v_add_f32 %3,%3,%8\n ")
DEF_KERNEL(dep1_med3_f32, "v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %0,%0,%8,%9\n ")
DEF_KERNEL(dep2_med3_f32, "v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n ")
DEF_KERNEL(dep4_med3_f32, "v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %2,%2,%8,%9\n v_med3_f32 %3,%3,%8,%9\n v_med3_f32 %0,%0,%8,%9\n v_med3_f32 %1,%1,%8,%9\n v_med3_f32 %2,%2,%8,%9\n v_med3_f32 %3,%3,%8,%9\n ")
DEF_KERNEL(dep1_xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %0,%0,%8\n ")
DEF_KERNEL(dep2_xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n ")
DEF_KERNEL(dep4_xor_b32, "v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %2,%2,%8\n v_xor_b32 %3,%3,%8\n v_xor_b32 %0,%0,%8\n v_xor_b32 %1,%1,%8\n v_xor_b32 %2,%2,%8\n v_xor_b32 %3,%3,%8\n ")
DEF_KERNEL(dep1_min_f32, "v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %0,%0,%8\n ")
DEF_KERNEL(dep2_min_f32, "v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n ")
DEF_KERNEL(dep4_min_f32, "v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %2,%2,%8\n v_min_f32 %3,%3,%8\n v_min_f32 %0,%0,%8\n v_min_f32 %1,%1,%8\n v_min_f32 %2,%2,%8\n v_min_f32 %3,%3,%8\n ")
DEF_KERNEL(dep1_and_or_b32, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n ")
DEF_KERNEL(dep2_and_or_b32, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n ")
DEF_KERNEL(dep4_and_or_b32, "v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %2,%2,%8,%9\n v_and_or_b32 %3,%3,%8,%9\n v_and_or_b32 %0,%0,%8,%9\n v_and_or_b32 %1,%1,%8,%9\n v_and_or_b32 %2,%2,%8,%9\n v_and_or_b32 %3,%3,%8,%9\n ")

DEF_KERNEL(add_u32, "v_add_u32 %0,%0,%8\n v_add_u32 %1,%1,%8\n v_add_u32 %2,%2,%8\n v_add_u32 %3,%3,%8\n v_add_u32 %4,%4,%8\n v_add_u32 %5,%5,%8\n v_add_u32 %6,%6,%8\n v_add_u32 %7,%7,%8\n ")
DEF_KERNEL(and_b32, "v_and_b32 %0,%0,%8\n v_and_b32 %1,%1,%8\n v_and_b32 %2,%2,%8\n v_and_b32 %3,%3,%8\n v_and_b32 %4,%4,%8\n v_and_b32 %5,%5,%8\n v_and_b32 %6,%6,%8\n v_and_b32 %7,%7,%8\n ")
DEF_KERNEL(or_b32, "v_or_b32 %0,%0,%8\n v_or_b32 %1,%1,%8\n v_or_b32 %2,%2,%8\n v_or_b32 %3,%3,%8\n v_or_b32 %4,%4,%8\n v_or_b32 %5,%5,%8\n v_or_b32 %6,%6,%8\n v_or_b32 %7,%7,%8\n ")
DEF_KERNEL(mul_f32, "v_mul_f32 %0,%0,%8\n v_mul_f32 %1,%1,%8\n v_mul_f32 %2,%2,%8\n v_mul_f32 %3,%3,%8\n v_mul_f32 %4,%4,%8\n v_mul_f32 %5,%5,%8\n v_mul_f32 %6,%6,%8\n v_mul_f32 %7,%7,%8\n ")
DEF_KERNEL(sub_f32, "v_sub_f32 %0,%0,%8\n v_sub_f32 %1,%1,%8\n v_sub_f32 %2,%2,%8\n v_sub_f32 %3,%3,%8\n v_sub_f32 %4,%4,%8\n v_sub_f32 %5,%5,%8\n v_sub_f32 %6,%6,%8\n v_sub_f32 %7,%7,%8\n ")
DEF_KERNEL(max_f32, "v_max_f32 %0,%0,%8\n v_max_f32 %1,%1,%8\n v_max_f32 %2,%2,%8\n v_max_f32 %3,%3,%8\n v_max_f32 %4,%4,%8\n v_max_f32 %5,%5,%8\n v_max_f32 %6,%6,%8\n v_max_f32 %7,%7,%8\n ")
DEF_KERNEL(lshlrev_v, "v_lshlrev_b32 %0,%9,%0\n v_lshlrev_b32 %1,%9,%1\n v_lshlrev_b32 %2,%9,%2\n v_lshlrev_b32 %3,%9,%3\n v_lshlrev_b32 %4,%9,%4\n v_lshlrev_b32 %5,%9,%5\n v_lshlrev_b32 %6,%9,%6\n v_lshlrev_b32 %7,%9,%7\n ")
DEF_KERNEL(add3_u32, "v_add3_u32 %0,%0,%8,%9\n v_add3_u32 %1,%1,%8,%9\n v_add3_u32 %2,%2,%8,%9\n v_add3_u32 %3,%3,%8,%9\n v_add3_u32 %4,%4,%8,%9\n v_add3_u32 %5,%5,%8,%9\n v_add3_u32 %6,%6,%8,%9\n v_add3_u32 %7,%7,%8,%9\n ")
DEF_KERNEL(lshl_add_u32, "v_lshl_add_u32 %0,%0,1,%9\n v_lshl_add_u32 %1,%1,1,%9\n v_lshl_add_u32 %2,%2,1,%9\n v_lshl_add_u32 %3,%3,1,%9\n v_lshl_add_u32 %4,%4,1,%9\n v_lshl_add_u32 %5,%5,1,%9\n v_lshl_add_u32 %6,%6,1,%9\n v_lshl_add_u32 %7,%7,1,%9\n ")
DEF_KERNEL(cndmask_sgpr, "v_cndmask_b32 %0,%0,%8,s[20:21]\n v_cndmask_b32 %1,%1,%8,s[20:21]\n v_cndmask_b32 %2,%2,%8,s[20:21]\n v_cndmask_b32 %3,%3,%8,s[20:21]\n v_cndmask_b32 %4,%4,%8,s[20:21]\n v_cndmask_b32 %5,%5,%8,s[20:21]\n v_cndmask_b32 %6,%6,%8,s[20:21]\n v_cndmask_b32 %7,%7,%8,s[20:21]\n ")
DEF_KERNEL(bfe_u32, "v_bfe_u32 %0,%0,3,5\n v_bfe_u32 %1,%1,3,5\n v_bfe_u32 %2,%2,3,5\n v_bfe_u32 %3,%3,3,5\n v_bfe_u32 %4,%4,3,5\n v_bfe_u32 %5,%5,3,5\n v_bfe_u32 %6,%6,3,5\n v_bfe_u32 %7,%7,3,5\n ")
DEF_KERNEL(cmp_eq_u32_sgpr, "v_cmp_eq_u32 s[20:21],%0,%8\n v_cmp_eq_u32 s[20:21],%1,%8\n v_cmp_eq_u32 s[20:21],%2,%8\n v_cmp_eq_u32 s[20:21],%3,%8\n v_cmp_eq_u32 s[20:21],%4,%8\n v_cmp_eq_u32 s[20:21],%5,%8\n v_cmp_eq_u32 s[20:21],%6,%8\n v_cmp_eq_u32 s[20:21],%7,%8\n ")
DEF_KERNEL(add_f32_abs, "v_add_f32 %0,|%0|,%8\n v_add_f32 %1,|%1|,%8\n v_add_f32 %2,|%2|,%8\n v_add_f32 %3,|%3|,%8\n v_add_f32 %4,|%4|,%8\n v_add_f32 %5,|%5|,%8\n v_add_f32 %6,|%6|,%8\n v_add_f32 %7,|%7|,%8\n ")
DEF_KERNEL(add_f32_e64, "v_add_f32_e64 %0,%0,%8\n v_add_f32_e64 %1,%1,%8\n v_add_f32_e64 %2,%2,%8\n v_add_f32_e64 %3,%3,%8\n v_add_f32_e64 %4,%4,%8\n v_add_f32_e64 %5,%5,%8\n v_add_f32_e64 %6,%6,%8\n v_add_f32_e64 %7,%7,%8\n ")
DEF_KERNEL(xor_e64, "v_xor_b32_e64 %0,%0,%8\n v_xor_b32_e64 %1,%1,%8\n v_xor_b32_e64 %2,%2,%8\n v_xor_b32_e64 %3,%3,%8\n v_xor_b32_e64 %4,%4,%8\n v_xor_b32_e64 %5,%5,%8\n v_xor_b32_e64 %6,%6,%8\n v_xor_b32_e64 %7,%7,%8\n ")
DEF_KERNEL(min_f32_lit, "v_min_f32 %0,1.0,%0\n v_min_f32 %1,1.0,%1\n v_min_f32 %2,1.0,%2\n v_min_f32 %3,1.0,%3\n v_min_f32 %4,1.0,%4\n v_min_f32 %5,1.0,%5\n v_min_f32 %6,1.0,%6\n v_min_f32 %7,1.0,%7\n ")
DEF_KERNEL(add_f32_lit, "v_add_f32 %0,1.0,%0\n v_add_f32 %1,1.0,%1\n v_add_f32 %2,1.0,%2\n v_add_f32 %3,1.0,%3\n v_add_f32 %4,1.0,%4\n v_add_f32 %5,1.0,%5\n v_add_f32 %6,1.0,%6\n v_add_f32 %7,1.0,%7\n ")
DEF_KERNEL(ashrrev, "v_ashrrev_i32 %0,31,%0\n v_ashrrev_i32 %1,31,%1\n v_ashrrev_i32 %2,31,%2\n v_ashrrev_i32 %3,31,%3\n v_ashrrev_i32 %4,31,%4\n v_ashrrev_i32 %5,31,%5\n v_ashrrev_i32 %6,31,%6\n v_ashrrev_i32 %7,31,%7\n ")
DEF_KERNEL(add_f32_sgpr, "v_add_f32 %0,s22,%0\n v_add_f32 %1,s22,%1\n v_add_f32 %2,s22,%2\n v_add_f32 %3,s22,%3\n v_add_f32 %4,s22,%4\n v_add_f32 %5,s22,%5\n v_add_f32 %6,s22,%6\n v_add_f32 %7,s22,%7\n ")
DEF_KERNEL(min_f32_sgpr, "v_min_f32 %0,s22,%0\n v_min_f32 %1,s22,%1\n v_min_f32 %2,s22,%2\n v_min_f32 %3,s22,%3\n v_min_f32 %4,s22,%4\n v_min_f32 %5,s22,%5\n v_min_f32 %6,s22,%6\n v_min_f32 %7,s22,%7\n ")

typedef float float2v __attribute__((ext_vector_type(2)));
#define DEF_PK(NAME, OP)                                                                                \
  __global__ void k_##NAME(unsigned long long* cyc, float* sink, int iters) {                           \
    float2v r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, \
            r6 = r0 + 6.f, r7 = r0 + 7.f;                                                               \
    float2v s = {1.0001f + blockIdx.x * 1e-7f, 0.999f}, u = {0.5f, 0.25f};                              \
    unsigned long long t0, t1;                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                         \
    for (int i = 0; i < iters; ++i) {                                                                   \
      asm volatile(OP OP OP OP                                                                          \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)     \
                   : "v"(s), "v"(u));                                                                   \
    }                                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                         \
    float2v a = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                                  \
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y;                                            \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;     \
  }
DEF_PK(pk_add_f32, "v_pk_add_f32 %0,%0,%8\n v_pk_add_f32 %1,%1,%8\n v_pk_add_f32 %2,%2,%8\n v_pk_add_f32 %3,%3,%8\n v_pk_add_f32 %4,%4,%8\n v_pk_add_f32 %5,%5,%8\n v_pk_add_f32 %6,%6,%8\n v_pk_add_f32 %7,%7,%8\n")
DEF_PK(pk_mul_f32, "v_pk_mul_f32 %0,%0,%8\n v_pk_mul_f32 %1,%1,%8\n v_pk_mul_f32 %2,%2,%8\n v_pk_mul_f32 %3,%3,%8\n v_pk_mul_f32 %4,%4,%8\n v_pk_mul_f32 %5,%5,%8\n v_pk_mul_f32 %6,%6,%8\n v_pk_mul_f32 %7,%7,%8\n")
DEF_PK(pk_fma_f32, "v_pk_fma_f32 %0,%0,%8,%9\n v_pk_fma_f32 %1,%1,%8,%9\n v_pk_fma_f32 %2,%2,%8,%9\n v_pk_fma_f32 %3,%3,%8,%9\n v_pk_fma_f32 %4,%4,%8,%9\n v_pk_fma_f32 %5,%5,%8,%9\n v_pk_fma_f32 %6,%6,%8,%9\n v_pk_fma_f32 %7,%7,%8,%9\n")
DEF_PK(pk_mov_b32, "v_pk_mov_b32 %0,%8,%9\n v_pk_mov_b32 %1,%8,%9\n v_pk_mov_b32 %2,%8,%9\n v_pk_mov_b32 %3,%8,%9\n v_pk_mov_b32 %4,%8,%9\n v_pk_mov_b32 %5,%8,%9\n v_pk_mov_b32 %6,%8,%9\n v_pk_mov_b32 %7,%8,%9\n")

typedef void (*kern_t)(unsigned long long*, float*, int);
struct Case { const char* name; kern_t k; int per_iter; };

int main(int argc, char** argv) {
  int ncu = 256;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  ncu = p.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
  const int iters = 2000;
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, sizeof(unsigned long long) * ncu * 64);
  hipMalloc(&sink, sizeof(float) * ncu * 1024);
  Case cases[] = {
#define C(N, P) {#N, k_##N, P}
      C(add_f32, 32), C(sub_abs_f32, 32), C(fma_f32, 32), C(min_f32, 32), C(med3_f32, 32), C(min3_f32, 32), C(xor_b32, 32),
      C(bfi_b32, 32), C(alignbit, 32), C(lshlrev, 32), C(and_or, 32), C(lshl_or, 32), C(cndmask_vcc, 32),
      C(cmp_vcc, 32), C(cmp_sgpr, 32), C(cmp_cnd_pair, 32), C(cmp_cnd_sgpr_pair, 32), C(mov_b32, 32),
      C(pk_add_f16, 32), C(pk_min_f16, 32), C(pk_min_i16, 32), C(pk_sub_i16, 32), C(pk_mul_lo_u16, 32),
      C(pk_ashrrev_i16, 32), C(perm_b32, 32), C(sub_u32, 32), C(min_u32, 32),
      C(add_u32, 32), C(and_b32, 32), C(or_b32, 32), C(mul_f32, 32), C(sub_f32, 32), C(max_f32, 32), C(lshlrev_v, 32), C(add3_u32, 32), C(lshl_add_u32, 32), C(cndmask_sgpr, 32), C(bfe_u32, 32), C(cmp_eq_u32_sgpr, 32), C(add_f32_abs, 32), C(add_f32_e64, 32), C(xor_e64, 32), C(min_f32_lit, 32), C(add_f32_lit, 32), C(ashrrev, 32), C(add_f32_sgpr, 32), C(min_f32_sgpr, 32), C(pk_add_f32, 32), C(pk_mul_f32, 32), C(pk_fma_f32, 32), C(pk_mov_b32, 32),
      C(dep1_add_f32, 32), C(dep2_add_f32, 32), C(dep4_add_f32, 32), C(dep1_med3_f32, 32), C(dep2_med3_f32, 32), C(dep4_med3_f32, 32), C(dep1_xor_b32, 32), C(dep2_xor_b32, 32), C(dep4_xor_b32, 32), C(dep1_min_f32, 32), C(dep2_min_f32, 32), C(dep4_min_f32, 32), C(dep1_and_or_b32, 32), C(dep2_and_or_b32, 32), C(dep4_and_or_b32, 32),
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
