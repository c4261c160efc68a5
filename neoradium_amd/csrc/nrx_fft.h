// Workgroup FFT in LDS (radix-2 decimation-in-frequency, in place, bit-reversed output) for gfx950.
// Shared by OFDM modulation/demodulation and the CIR -> channel-matrix transform.
#pragma once
#include "nrx_cplx.h"

namespace nrx {

// tw[k] = exp(-2*pi*i*k/n), k < n/2, evaluated with sincospi in float64 (exact quadrant handling).
template <typename T>
__device__ __forceinline__ void fft_fill_twiddles(cx<T>* tw, int n) {
  for (int k = threadIdx.x; k < n / 2; k += blockDim.x) {
    double s, c;
    sincospi(2.0 * (double)k / (double)n, &s, &c);
    tw[k] = cx<T>((T)c, (T)(-s));
  }
}

// In-place DIF FFT of buf[0..n): X[k] ends up at buf[bitrev(k)].  inverse: conjugated twiddles, no scaling.
// All threads of the workgroup must call it; it ends with a barrier.
template <typename T>
__device__ __forceinline__ void fft_dif_lds(cx<T>* buf, const cx<T>* tw, int n, int log2n, bool inverse) {
  for (int s = 0; s < log2n; ++s) {
    const int hb = log2n - 1 - s;  // log2 of the half span
    const int h = 1 << hb;
    for (int i = threadIdx.x; i < n / 2; i += blockDim.x) {
      const int j = i & (h - 1);
      const int a = ((i >> hb) << (hb + 1)) + j, b = a + h;
      const cx<T> u = buf[a], v = buf[b];
      cx<T> w = tw[j << s];
      if (inverse) w.im = -w.im;
      buf[a] = u + v;
      buf[b] = (u - v) * w;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int fft_bitrev(int k, int log2n) { return (int)(__brev((unsigned)k) >> (32 - log2n)); }

}  // namespace nrx
