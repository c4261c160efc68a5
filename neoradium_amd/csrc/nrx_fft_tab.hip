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

const cx<double>* fft_twiddle_table(hipStream_t stream) {
  static std::once_flag once;
  static const cx<double>* table = nullptr;
  std::call_once(once, [&]() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fft_tw)) != hipSuccess || !p) return;
    hipLaunchKernelGGL(fft_tw_init_kernel, dim3(FFT_TW_N / 2 / 256), dim3(256), 0, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return;   // one-time: later calls may come on any stream
    table = (const cx<double>*)p;
  });
  return table;
}

}  // namespace nrx
