// The shared FFT twiddle table (see nrx_fft.h).
#include <mutex>
#include "nrx_fft.h"

namespace nrx {

__device__ cx<double> g_fft_tw[FFT_TW_N / 2];

__global__ void fft_tw_init_kernel() {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < FFT_TW_N / 2) {
    double s, c;
    sincospi(2.0 * (double)k / (double)FFT_TW_N, &s, &c);
    g_fft_tw[k] = cx<double>(c, -s);
  }
}

// One table per device (a __device__ symbol has one address per device); filled on first use on that device.  A failed
// initialisation (e.g. a synchronisation refused during stream capture) is not remembered: the next call tries again.
const cx<double>* fft_twiddle_table(hipStream_t stream) {
  constexpr int MAX_DEV = 16;
  static std::mutex mu;
  static const cx<double>* table[MAX_DEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!table[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fft_tw)) != hipSuccess || !p) return nullptr;
    hipLaunchKernelGGL(fft_tw_init_kernel, dim3(FFT_TW_N / 2 / 256), dim3(256), 0, stream);
    if (hipGetLastError() != hipSuccess) return nullptr;
    if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;   // one-time per device: later calls may come on any stream
    table[dev] = (const cx<double>*)p;
  }
  return table[dev];
}

}  // namespace nrx
