// Micro-benchmark (dev tool, round 3): do scalar stores work on gfx950 and what do they cost?  Each wave keeps a private
// stash of N 64-bit masks in global memory: per "iteration" it reloads every mask (s_load_dwordx2), combines it with a
// per-iteration value and stores it back (s_store_dwordx2) -- the pattern the LDPC decoder would use to carry the argmin masks
// of one iteration's write pass to the next iteration's read pass without a VALU instruction.  Checks the final contents.
//   hipcc --offload-arch=gfx950 -O2 -o sstore_probe sstore_probe.hip && ./sstore_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

constexpr int N = 160;   // masks per wave

__global__ void k_stash(unsigned long long* stash, unsigned long long* cyc, int iters, int with_valu) {
  const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  unsigned long long* mine = stash + (size_t)wave * N;
  unsigned long long base = (unsigned long long)mine;
  base = __builtin_amdgcn_readfirstlane((unsigned)base) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32);
  unsigned long long t0, t1;
  double a = threadIdx.x, b = 1.000001;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < N; k += 4) {
      unsigned long long m0, m1, m2, m3;
      asm volatile("s_load_dwordx2 %0, %4, %5\n s_load_dwordx2 %1, %4, %6\n s_load_dwordx2 %2, %4, %7\n s_load_dwordx2 %3, %4, %8\n s_waitcnt lgkmcnt(0)"
                   : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "s"(base), "i"(0), "i"(8), "i"(16), "i"(24) : "memory");
      if (with_valu) asm volatile(".rept 16\n v_fma_f64 %0, %0, %1, %1\n .endr" : "+v"(a) : "v"(b));
      m0 += 1; m1 += 2; m2 += 3; m3 += 4;
      asm volatile("s_store_dwordx2 %0, %4, %5\n s_store_dwordx2 %1, %4, %6\n s_store_dwordx2 %2, %4, %7\n s_store_dwordx2 %3, %4, %8"
                   :: "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(base), "i"(0), "i"(8), "i"(16), "i"(24) : "memory");
      base += 32;
    }
    base -= 8 * N;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
  if (a == 1.2345) cyc[0] = 0;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount, waves = ncu * 12;
  unsigned long long *stash, *cyc;
  hipMalloc(&stash, sizeof(unsigned long long) * waves * N);
  hipMalloc(&cyc, sizeof(unsigned long long) * waves);
  for (int with_valu = 0; with_valu <= 1; ++with_valu) {
    const int iters = 200;
    hipMemset(stash, 0, sizeof(unsigned long long) * waves * N);
    hipLaunchKernelGGL(k_stash, dim3(ncu), dim3(768), 0, 0, stash, cyc, iters, with_valu);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves * N), c(waves);
    hipMemcpy(h.data(), stash, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < waves; ++w)
      for (int k = 0; k < N; ++k)
        if (h[(size_t)w * N + k] != (unsigned long long)iters * (k % 4 + 1)) ++bad;
    unsigned long long mx = 0;
    double mean = 0;
    for (auto v : c) { mx = v > mx ? v : mx; mean += (double)v; }
    mean /= waves;
    printf("%s: %d waves x %d masks x %d iterations: %ld wrong values; cycles per (load + store) pair per wave: mean %.1f, slowest wave %.1f%s\n",
           with_valu ? "with 4 v_fma_f64 per mask" : "scalar only", waves, N, iters, bad, mean / ((double)iters * N), (double)mx / ((double)iters * N),
           with_valu ? "  (the FMAs alone need 16 cycles per mask per wave x 3 waves per SIMD = 48)" : "");
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return 0;
}
