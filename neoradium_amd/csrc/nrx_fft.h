// Workgroup FFT in LDS (radix-2 decimation-in-frequency with 4 stages fused per pass, in place, bit-reversed
// output) for gfx950.
// Shared by OFDM modulation/demodulation and the CIR -> channel-matrix transform.
#pragma once
#include "nrx_common.h"
#include "nrx_cplx.h"

namespace nrx {

// Twiddles: one device-resident table W[k] = exp(-2*pi*i*k/FFT_TW_N), k < FFT_TW_N/2, evaluated once with sincospi
// in float64 (exact quadrant handling); an n-point transform uses every (FFT_TW_N/n)-th entry (power-of-two scaling
// of the argument is exact, so the values equal exp(-2*pi*i*k/n) computed directly).  Kept in global memory (64 KB,
// L1/L2 resident) so that an LDS FFT needs only its data buffer: two 4096-point float64 workgroups per CU.
constexpr int FFT_TW_N = 8192;
// Device pointer of the table; the first call fills it on `stream` and waits for that once (nrx_fft_tab.hip).
const cx<double>* fft_twiddle_table(hipStream_t stream);

// LDS layout of an FFT buffer: logical element i lives at i + (i >> 4), one element of padding per 16.  In the last fused
// pass every thread owns 16 CONSECUTIVE elements (lane stride 256 B: all lanes on the same four banks, the pass ran
// serialised); with the padding the lane stride is 17 elements and eight neighbouring lanes cover all 32 banks.
// Every access to a buffer handed to fft_dif_lds goes through fft_idx; its size is fft_lds_elems(n).
__host__ __device__ __forceinline__ constexpr int fft_idx(int i) { return i + (i >> 4); }
__host__ __device__ __forceinline__ constexpr size_t fft_lds_elems(size_t n) { return n + (n >> 4); }

// LG consecutive radix-2 DIF stages (starting at stage s) fused in registers: a thread loads the 2^LG points that
// only interact with each other during those stages, runs the butterflies, stores them back.  Same arithmetic and
// same (bit-reversed) result order as stage-by-stage radix-2, but one LDS round trip and one barrier per LG stages.
template <typename T, int LG>
__device__ __forceinline__ void fft_dif_fused(cx<T>* buf, const cx<double>* __restrict__ tw, int tws, int n, int log2n,
                                              int s, bool inverse) {
  constexpr int P = 1 << LG;
  const int lq = log2n - s - LG;   // log2 of the point spacing q
  const int q = 1 << lq;
  for (int gi = threadIdx.x; gi < (n >> LG); gi += blockDim.x) {
    const int lo = gi & (q - 1);
    const int a = ((gi >> lq) << (lq + LG)) | lo;
    // the P - 1 distinct twiddles of the fused stages (stage t uses P >> (t+1) of them, each for 2^t butterflies): all
    // fetched up front, so their latency is paid once per pass instead of once per dependent batch of butterflies
    // (the loads are fenced from what follows: left to itself the scheduler sinks every one of them to its first use to
    //  save registers -- load, wait, multiply, P - 1 dependent round trips per pass)
    cx<double> wd[P - 1];
#pragma unroll
    for (int t = 0; t < LG; ++t) {
      const int half = P >> (t + 1);
#pragma unroll
      for (int r = 0; r < half; ++r) {
        const int j = (lo + (r << lq)) << (s + t);
        wd[(P - 2 * half) + r] = tw[(size_t)j * tws];                             // offsets 0, P/2, 3P/4, ...
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    cx<T> wt[P - 1];
#pragma unroll
    for (int m = 0; m < P - 1; ++m) wt[m] = cx<T>((T)wd[m].re, inverse ? -(T)wd[m].im : (T)wd[m].im);
    cx<T> x[P];
#pragma unroll
    for (int m = 0; m < P; ++m) x[m] = buf[fft_idx(a + (m << lq))];
#pragma unroll
    for (int t = 0; t < LG; ++t) {
      const int half = P >> (t + 1);   // partner distance in m
#pragma unroll
      for (int m = 0; m < P; ++m) {
        if (m & half) continue;        // m is the upper element of its pair
        const cx<T> w = wt[(P - 2 * half) + (m & (half - 1))];
        const cx<T> u = x[m], v = x[m + half];
        x[m] = u + v;
        x[m + half] = (u - v) * w;
      }
    }
#pragma unroll
    for (int m = 0; m < P; ++m) buf[fft_idx(a + (m << lq))] = x[m];
  }
  __syncthreads();
}

// In-place DIF FFT of the (padded, see fft_idx) buffer: X[k] ends up at buf[fft_idx(bitrev(k))].  inverse: conjugated twiddles, no scaling.
// All threads of the workgroup must call it; it ends with a barrier.
template <typename T>
__device__ __forceinline__ void fft_dif_lds(cx<T>* buf, const cx<double>* __restrict__ tw, int n, int log2n,
                                            bool inverse) {
  const int tws = FFT_TW_N / n;
  int s = 0;
  while (log2n - s >= 4) {
    fft_dif_fused<T, 4>(buf, tw, tws, n, log2n, s, inverse);
    s += 4;
  }
  const int rem = log2n - s;
  if (rem == 3) fft_dif_fused<T, 3>(buf, tw, tws, n, log2n, s, inverse);
  else if (rem == 2) fft_dif_fused<T, 2>(buf, tw, tws, n, log2n, s, inverse);
  else if (rem == 1) fft_dif_fused<T, 1>(buf, tw, tws, n, log2n, s, inverse);
}

__device__ __forceinline__ int fft_bitrev(int k, int log2n) { return (int)(__brev((unsigned)k) >> (32 - log2n)); }

}  // namespace nrx
