// "Perfect" channel state of the BLER harness without a channel matrix in memory (gfx950, float64).
//
// The harness's perfect CSI is  Hest = channelMatrix @ precoder  (PDSCH-BLER.ipynb cell 2) with
//   channelMatrix[c][k] = FFT_nfft( cir[c] advanced by chanOffset )[(k - K/2) mod nfft]          (channelmodel.py:362-400)
//   cir[c][r][t][l]     = sum_p gains[c][r][t][p] * coeff[p][l]                                   (channelmodel.py:343)
// The transform is linear in the path gains and the coefficient matrix is a constant of the channel, so with
//   S_p[k] = sum_j taps[p][j] * exp(-2 pi i k' (off_p + j) / nfft),  k' = (k - K/2) mod nfft      (nrx_td_path_spectra_bins_f64, once per link)
//   Hest[c][k][r][l] = exp(+2 pi i k' chanOffset / nfft) * sum_p gfold[c][r][l][p] * S_p[k]
// where gfold already holds the wideband precoder (nrx_fold_precoder_f64: sum_t gains[c][r][t][p] F[t][l]).  One thread per RE forms its
// Nr x Nl matrix from the P path spectra (P x Nr x Nl complex multiply-adds; the (item, symbol)'s gains are workgroup-uniform and go
// through LDS) and runs the MMSE solve of nrx_mmse.h on it: Grid.equalize (grid.py:626-694) on perfect CSI with NOTHING of the
// (L, K, Nr, Nt) channel matrix written or read -- at 273 PRB, 4 x 4 the matrix is 11.7 MB per slot, and the three kernels it went
// through (FFT per (symbol, antenna pair) with 16-byte strided stores, H @ F, equaliser) took 8.6 ms per 256 slots.
// Same values as the FFT route up to rounding (|diff| ~ 1e-15 of the largest entry); the FFT route stays for the frequency-domain link
// (which needs the matrix itself to apply the channel), per-PRG precoders and details=True.
#include "nrx_common.h"
#include "nrx_fft.h"
#include "nrx_mmse.h"

namespace {
using nrx::cx;
typedef cx<double> cd;

// W_nfft^m = exp(-2 pi i m / nfft), 0 <= m < nfft, from the half-circle table W_8192^k (k < 4096)
__device__ __forceinline__ cd w_nfft(const cd* __restrict__ tw, int m, int nfft) {
  const int s = nrx::FFT_TW_N / nfft;
  const bool upper = m >= nfft / 2;
  const cd w = tw[(size_t)(upper ? m - nfft / 2 : m) * s];
  return upper ? cd(-w.re, -w.im) : w;
}

__global__ void __launch_bounds__(256)
path_spectra_bins_kernel(const double* __restrict__ taps, const int32_t* __restrict__ tap_off, int n_paths, int flen, int K, int nfft,
                         const cd* __restrict__ tw, cd* __restrict__ spec) {
  const int64_t total = (int64_t)n_paths * K;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(g / K), k = (int)(g - (int64_t)p * K);
    const int kp = (k - K / 2 + nfft) & (nfft - 1);
    const int off = tap_off[p];
    double ar = 0.0, ai = 0.0;
    for (int j = 0; j < flen; ++j) {
      const int m = (int)(((int64_t)kp * (off + j)) & (nfft - 1));
      const cd w = w_nfft(tw, m, nfft);
      const double c = taps[(size_t)p * flen + j];
      ar += c * w.re;
      ai += c * w.im;
    }
    spec[g] = cd(ar, ai);
  }
}

template <int NR, int NL>
__global__ void __launch_bounds__(128, 2)
mmse_paths_kernel(const cd* __restrict__ rx, const cd* __restrict__ gains, int n_sets, int n_paths, const cd* __restrict__ spec,
                  const int32_t* __restrict__ chan_off, const double* __restrict__ noise_var, int nv_stride, int L, int K, int nfft,
                  const cd* __restrict__ tw, uint32_t sym_mask, cd* __restrict__ eq, double* __restrict__ scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* gl = (cd*)smem;                       // [NR * NL][n_paths]: the gains of this (item, symbol)
  const int l = blockIdx.y % L, b = blockIdx.y / L;
  if (!((sym_mask >> (l & 31)) & 1u)) return;          // a symbol the caller does not want equalised (whole workgroup)
  const cd* gb = gains + ((size_t)b * n_sets + l) * NR * NL * n_paths;
  for (int i = threadIdx.x; i < NR * NL * n_paths; i += blockDim.x) gl[i] = gb[i];
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  cd H[NR][NL];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int q = 0; q < NL; ++q) H[r][q] = cd(0.0, 0.0);
  cd c = spec[k];
  for (int p = 0; p < n_paths; ++p) {
    const cd cn = spec[(size_t)(p + 1 < n_paths ? p + 1 : p) * K + k];       // (the next path's spectrum is in flight)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int q = 0; q < NL; ++q) {
        const cd gg = gl[(r * NL + q) * n_paths + p];
        H[r][q].re = fma(gg.re, c.re, fma(-gg.im, c.im, H[r][q].re));
        H[r][q].im = fma(gg.re, c.im, fma(gg.im, c.re, H[r][q].im));
      }
    c = cn;
  }
  // the circular advance by chanOffset (channelmodel.py:389-393): exp(+2 pi i k' o / nfft)
  const int kp = (k - K / 2 + nfft) & (nfft - 1);
  const int o = chan_off[b];
  const cd w = w_nfft(tw, (int)(((int64_t)kp * o) & (nfft - 1)), nfft);
  const cd ph(w.re, -w.im);
  cd y[NR];
  const size_t lk = (size_t)L * K;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    y[r] = rx[((size_t)b * NR + r) * lk + (size_t)l * K + k];
#pragma unroll
    for (int q = 0; q < NL; ++q) H[r][q] = H[r][q] * ph;
  }
  double nv = noise_var[(size_t)b * nv_stride];
  nv = nv > 1e-8 ? nv : 1e-8;  // grid.py:676
  cd xh[NL];
  double sc[NL];
  nrx::mmse_solve<NR, NL>(H, y, nv, xh, sc);
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    eq[((size_t)b * NL + q) * lk + (size_t)l * K + k] = xh[q];
    scale[((size_t)b * NL + q) * lk + (size_t)l * K + k] = sc[q];
  }
}

template <int NR>
int32_t launch_nl(int nl, dim3 grid, size_t lds, hipStream_t st, const cd* rx, const cd* gains, int n_sets, int n_paths, const cd* spec,
                  const int32_t* off, const double* nv, int nv_stride, int L, int K, int nfft, const cd* tw, uint32_t mask, cd* eq, double* sc) {
#define NRX_MP_CASE(NL)                                                                                                              \
  case NL:                                                                                                                           \
    hipLaunchKernelGGL((mmse_paths_kernel<NR, NL>), grid, dim3(128), lds, st, rx, gains, n_sets, n_paths, spec, off, nv, nv_stride, L, K, \
                       nfft, tw, mask, eq, sc);                                                                                      \
    return NRX_OK;
  switch (nl) {
    NRX_MP_CASE(1)
    NRX_MP_CASE(2)
    NRX_MP_CASE(3)
    NRX_MP_CASE(4)
    default: return NRX_E_UNSUPPORTED;
  }
#undef NRX_MP_CASE
}

}  // namespace

extern "C" int32_t nrx_td_path_spectra_bins_f64(const double* taps, const int32_t* tap_off, int32_t n_paths, int32_t flen, int32_t K,
                                                int32_t nfft, void* spec, void* stream) {
  NRX_REQUIRE(taps && tap_off && spec, NRX_E_ARG, "nrx_td_path_spectra_bins: NULL buffer");
  NRX_REQUIRE(n_paths >= 1 && flen >= 1 && K >= 1 && K <= nfft && nfft >= 64 && nfft <= nrx::FFT_TW_N && (nfft & (nfft - 1)) == 0, NRX_E_ARG,
              "nrx_td_path_spectra_bins: bad sizes");
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_td_path_spectra_bins: FFT twiddle table unavailable");
  hipLaunchKernelGGL(path_spectra_bins_kernel, dim3(nrx::stream_grid((long)n_paths * K, 256)), dim3(256), 0, (hipStream_t)stream, taps, tap_off,
                     n_paths, flen, K, nfft, tw, (cd*)spec);
  NRX_CHECK_LAUNCH("nrx_td_path_spectra_bins");
  return NRX_OK;
}

extern "C" int32_t nrx_mmse_equalize_paths_f64(const void* rx, const void* gains, int32_t n_sets, const void* spec, const int32_t* chan_off,
                                               const double* noise_var, int32_t nv_stride, int32_t n_rx, int32_t n_layers, int32_t n_paths,
                                               int32_t L, int32_t K, int32_t nfft, uint32_t sym_mask, void* eq, void* scale, int32_t n_batch,
                                               void* stream) {
  NRX_REQUIRE(rx && gains && spec && chan_off && noise_var && eq && scale, NRX_E_ARG, "nrx_mmse_equalize_paths: NULL buffer");
  NRX_REQUIRE(L >= 1 && L <= 32 && n_sets >= L && K >= 1 && K <= nfft && n_paths >= 1 && n_batch >= 0 && nfft >= 64 && nfft <= nrx::FFT_TW_N &&
              (nfft & (nfft - 1)) == 0, NRX_E_ARG, "nrx_mmse_equalize_paths: bad sizes");
  const size_t lds = sizeof(cd) * (size_t)n_rx * n_layers * n_paths;
  if (!((n_rx == 1 || n_rx == 2 || n_rx == 4) && n_layers >= 1 && n_layers <= 4 && lds <= 64 * 1024)) {
    ::nrx::set_error("nrx_mmse_equalize_paths: built for Nr in {1, 2, 4}, 1..4 layers (Nr %d, layers %d, %d paths)", n_rx, n_layers, n_paths);
    return NRX_E_UNSUPPORTED;
  }
  if (n_batch == 0) return NRX_OK;
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_mmse_equalize_paths: FFT twiddle table unavailable");
  const dim3 grid((K + 127) / 128, (unsigned)(n_batch * L));
  hipStream_t st = (hipStream_t)stream;
  int32_t rc = NRX_E_UNSUPPORTED;
  switch (n_rx) {
    case 1: rc = launch_nl<1>(n_layers, grid, lds, st, (const cd*)rx, (const cd*)gains, n_sets, n_paths, (const cd*)spec, chan_off, noise_var, nv_stride, L, K, nfft, tw, sym_mask, (cd*)eq, (double*)scale); break;
    case 2: rc = launch_nl<2>(n_layers, grid, lds, st, (const cd*)rx, (const cd*)gains, n_sets, n_paths, (const cd*)spec, chan_off, noise_var, nv_stride, L, K, nfft, tw, sym_mask, (cd*)eq, (double*)scale); break;
    case 4: rc = launch_nl<4>(n_layers, grid, lds, st, (const cd*)rx, (const cd*)gains, n_sets, n_paths, (const cd*)spec, chan_off, noise_var, nv_stride, L, K, nfft, tw, sym_mask, (cd*)eq, (double*)scale); break;
    default: break;
  }
  NRX_CHECK_LAUNCH("nrx_mmse_equalize_paths");
  return rc;
}
