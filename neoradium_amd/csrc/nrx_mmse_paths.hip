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

// ------------------------------------------------------------------------------ channel set-up from the path spectra
// chanOffset and the channel matrix at n_k subcarriers (what nrx_chan_setup_f64 produces by a direct DFT of the CIR over its cl taps),
// with the sums over taps taken out of the per-row work:
//   chanOffset = argmax_l sum_r | sum_p G[r][p] coeff[p][l] |,   G[r][p] = sum_{c<nc, t} gains[c][r][t][p]      (channelmodel.py:343-346)
//   H[c][k][rt] = exp(+2 pi i k' o / nfft) * sum_p gains[c][rt][p] * S_p[k0 + k]                                (channelmodel.py:362-400)
// One workgroup per item; P (r, t, k) products instead of cl (r, t, k) products per matrix entry: 20x fewer multiply-adds at CDL-C
// (24 paths, 335 taps).  Same values as nrx_chan_setup_f64 up to the order of the sums (~1e-16 relative).
constexpr int CSP_THREADS = 512;
__global__ void __launch_bounds__(CSP_THREADS)
chan_setup_paths_kernel(const cd* __restrict__ gains, const double* __restrict__ coeff, const cd* __restrict__ spec, int64_t spec_stride,
                        int n_t_total, int nc, int nr, int nt, int n_paths, int cl, int K, int nfft, int k0, int n_k,
                        int32_t* __restrict__ off_out, cd* __restrict__ H, const cd* __restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* G = (cd*)smem;                        // [nr][n_paths]
  __shared__ double bestv[CSP_THREADS / 64];
  __shared__ int besti[CSP_THREADS / 64];
  __shared__ int off_s;
  const int b = blockIdx.x, tid = threadIdx.x, n_rt = nr * nt;
  const cd* gb = gains + (size_t)b * n_t_total * n_rt * n_paths;
  for (int i = tid; i < nr * n_paths; i += CSP_THREADS) {
    const int r = i / n_paths, p = i - r * n_paths;
    cd s(0, 0);
    for (int c = 0; c < nc; ++c)
      for (int t = 0; t < nt; ++t) s = s + gb[((size_t)c * n_rt + r * nt + t) * n_paths + p];
    G[i] = s;
  }
  __syncthreads();
  double bv = -1.0;
  int bi = 0;
  for (int l = tid; l < cl; l += CSP_THREADS) {
    double tot = 0;
    for (int r = 0; r < nr; ++r) {
      double ar = 0, ai = 0;
      for (int p = 0; p < n_paths; ++p) {
        const double cf = coeff[(size_t)p * cl + l];
        ar += G[r * n_paths + p].re * cf;
        ai += G[r * n_paths + p].im * cf;
      }
      tot += hypot(ar, ai);
    }
    if (tot > bv) { bv = tot; bi = l; }
  }
  // first maximum (largest value, smallest tap among equals -- np.argmax)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { bestv[tid >> 6] = bv; besti[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    double v = -1.0;
    int idx = 0;
    for (int i = 0; i < CSP_THREADS / 64; ++i)
      if (bestv[i] > v || (bestv[i] == v && besti[i] < idx)) { v = bestv[i]; idx = besti[i]; }
    off_out[b] = idx;
    off_s = idx;
  }
  __syncthreads();
  const int o = off_s;
  for (int i = tid; i < nc * n_k * n_rt; i += CSP_THREADS) {       // H (nc, n_k, n_rt): rt fastest
    const int rt = i % n_rt, k = (i / n_rt) % n_k, c = i / (n_rt * n_k);
    const cd* gr = gb + ((size_t)c * n_rt + rt) * n_paths;
    double ar = 0, ai = 0;
    for (int p = 0; p < n_paths; ++p) {
      const cd g = gr[p], sp = spec[(size_t)p * spec_stride + k0 + k];
      ar = fma(g.re, sp.re, fma(-g.im, sp.im, ar));
      ai = fma(g.re, sp.im, fma(g.im, sp.re, ai));
    }
    const int kp = (k0 + k - K / 2 + nfft) & (nfft - 1);
    const cd w = w_nfft(tw, (int)(((int64_t)kp * o) & (nfft - 1)), nfft);
    H[(size_t)b * nc * n_k * n_rt + i] = cd(ar, ai) * cd(w.re, -w.im);
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

extern "C" int32_t nrx_chan_setup_paths_f64(const void* gains, const double* coeff, const void* spec, int64_t spec_stride, int32_t n_items,
                                            int32_t n_t, int32_t nc, int32_t n_rx, int32_t n_tx, int32_t n_paths, int32_t cl, int32_t K,
                                            int32_t nfft, int32_t k0, int32_t n_k, int32_t* chan_offset, void* H, void* stream) {
  NRX_REQUIRE(gains && coeff && spec && chan_offset && H, NRX_E_ARG, "nrx_chan_setup_paths: NULL buffer");
  NRX_REQUIRE(n_t >= 1 && nc >= 1 && nc <= n_t && n_rx >= 1 && n_tx >= 1 && n_paths >= 1 && cl >= 1 && n_items >= 0 && spec_stride >= K,
              NRX_E_ARG, "nrx_chan_setup_paths: bad sizes");
  NRX_REQUIRE(nfft >= 64 && nfft <= nrx::FFT_TW_N && (nfft & (nfft - 1)) == 0 && K > 0 && K <= nfft && k0 >= 0 && n_k >= 1 && k0 + n_k <= K, NRX_E_ARG,
              "nrx_chan_setup_paths: bad nfft / K / subcarrier range");
  const size_t lds = sizeof(cd) * (size_t)n_rx * n_paths;
  NRX_REQUIRE(lds <= 64 * 1024, NRX_E_UNSUPPORTED, "nrx_chan_setup_paths: Nr x paths too large (%zu B)", lds);
  if (n_items == 0) return NRX_OK;
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_chan_setup_paths: FFT twiddle table unavailable");
  hipLaunchKernelGGL(chan_setup_paths_kernel, dim3(n_items), dim3(CSP_THREADS), lds, (hipStream_t)stream, (const cd*)gains, coeff, (const cd*)spec,
                     spec_stride, n_t, nc, n_rx, n_tx, n_paths, cl, K, nfft, k0, n_k, chan_offset, (cd*)H, tw);
  NRX_CHECK_LAUNCH("nrx_chan_setup_paths");
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
  NRX_REQUIRE((int64_t)n_batch * L <= 65535, NRX_E_SHAPE, "nrx_mmse_equalize_paths: n_batch x L = %lld exceeds the grid's 65535 (split the batch)",
              (long long)n_batch * L);
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
