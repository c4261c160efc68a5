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
      C(mov_b64, 32), C(ds_read_b64, 32), C(ds_write_b64, 32),
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
