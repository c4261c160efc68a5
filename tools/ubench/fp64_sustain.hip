// Micro-benchmark (dev tool): SUSTAINED float64 throughput and shader clock of a vector-FMA stream against an MFMA stream on
// gfx950 -- does the matrix pipe keep its clock where the vector pipe throttles (nrx_chan.hip's channel filter runs at 1.95 GHz)?
// Each kernel: 256 x 2 workgroups of 512 threads (4 waves per SIMD), independent accumulator chains, `iters` rounds; launched
// back to back for ~150 ms; prints TFLOP/s from HIP events and the clock from s_memtime / s_memrealtime (100 MHz).
//   hipcc --offload-arch=gfx950 -O2 -o fp64_sustain fp64_sustain.hip && ./fp64_sustain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) k_vfma(double* sink, unsigned long long* clk, int iters) {
  double a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = threadIdx.x * 1e-3 + k;
  const double s = 1.0000001, u = 1e-9 * blockIdx.x;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = __builtin_fma(a[k], s, u);
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += a[k];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(512) k_mfma(double* sink, unsigned long long* clk, int iters) {
  d4 c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = (d4){0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-6, b = 1e-3 * (1 + (blockIdx.x & 3));
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += c[k].x + c[k].y + c[k].z + c[k].w;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int grid = 512, threads = 512, launches = 40;
  double* sink; unsigned long long* clk;
  CK(hipMalloc(&sink, sizeof(double) * grid * threads));
  CK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * grid));
  unsigned long long h[2 * 512];
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    // per round and wave: vector 8 x 16 FMAs x 64 lanes x 2 flop; matrix 16 MFMAs x 2048 flop
    const double flop_round_wave = which == 0 ? 8.0 * 16 * 64 * 2 : 16.0 * 2048;
    const int iters = which == 0 ? 6000 : 3000;
    for (int w = 0; w < 2; ++w) {      // warm-up launch, then the timed series
      const int n = w == 0 ? 2 : launches;
      CK(hipEventRecord(e0));
      for (int l = 0; l < n; ++l) {
        if (which == 0) hipLaunchKernelGGL(k_vfma, dim3(grid), dim3(threads), 0, 0, sink, clk, iters);
        else hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(threads), 0, 0, sink, clk, iters);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      if (w == 1) {
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        double st = 0, sr = 0;
        for (int i = 0; i < grid; ++i) { st += (double)h[2 * i]; sr += (double)h[2 * i + 1]; }
        const double flops = flop_round_wave * iters * (threads / 64.0) * grid * n;
        printf("%s: %d launches, %.2f ms each, %.1f TFLOP/s sustained, shader clock %.0f MHz (last launch)\n",
               which == 0 ? "v_fma_f64          " : "v_mfma_f64_16x16x4", n, ms / n, flops / (ms * 1e-3) / 1e12, st / (sr / 100.0));
      }
    }
  }
  return 0;
}
