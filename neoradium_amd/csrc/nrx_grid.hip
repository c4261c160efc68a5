// Per-resource-element grid stages (gfx950): precoding, frequency-domain channel, MMSE equalisation,
// complex variance (noise scaling), AWGN.  One RE per lane, float64 arithmetic inside; no MFMA -- the
// matrices are 1x1 .. 8x8 per RE, not a contraction.
//
// Replaces reference grid.py:456-518 (precode), :978-1018 (applyChannel), :626-694 (equalize),
// :1040-1046 (getNoiseStd -> np.var), :1049-1187 + waveform.py:145-292 (addNoise), random.py:203 (awgn).
#include "nrx_common.h"
#include "nrx_cplx.h"
#include "nrx_mmse.h"
#include "nrx_rng.h"

namespace {
using nrx::cx;
typedef cx<double> cd;

// ------------------------------------------------------------------------------------------------ precode
// out[b][t][lk] = sum_n F[b][t][n] * grid[b][n][lk]            (wideband Nt x Nl precoder, grid.py:505-516)
template <typename T>
__global__ void __launch_bounds__(256)
precode_kernel(const cx<T>* __restrict__ grid, const cx<T>* __restrict__ f, int64_t f_stride, int nl, int nt, int lk,
               cx<T>* __restrict__ out, int n_batch) {
  const int64_t total = (int64_t)n_batch * lk;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / lk), i = (int)(g - (int64_t)b * lk);
    cd x[8];
    for (int n = 0; n < nl; ++n) x[n] = cd(grid[((size_t)b * nl + n) * lk + i]);
    const cx<T>* fb = f + (size_t)b * f_stride;
    for (int t = 0; t < nt; ++t) {
      cd acc(0, 0);
      for (int n = 0; n < nl; ++n) nrx::cmac(acc, cd(fb[t * nl + n]), x[n]);
      out[((size_t)b * nt + t) * lk + i] = cx<T>(acc);
    }
  }
}

// Per-PRG precoder (grid.py:482-493 with the (rbList, F) list of pdsch.py:1132-1165): subcarrier k uses matrix
// k2g[k] of item b; k2g[k] < 0 = no group covers the subcarrier -> the output is zero there, like the reference's
// zero-initialised (K, Nt, Nl) precoder tensor.
template <typename T>
__global__ void __launch_bounds__(256)
precode_prg_kernel(const cx<T>* __restrict__ grid, const cx<T>* __restrict__ f, int64_t f_stride,
                   const int32_t* __restrict__ k2g, int nl, int nt, int L, int K, cx<T>* __restrict__ out, int n_batch) {
  const int lk = L * K;
  const int64_t total = (int64_t)n_batch * lk;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / lk), i = (int)(g - (int64_t)b * lk);
    const int grp = k2g[i % K];
    if (grp < 0) {
      for (int t = 0; t < nt; ++t) out[((size_t)b * nt + t) * lk + i] = cx<T>(0, 0);
      continue;
    }
    cd x[8];
    for (int n = 0; n < nl; ++n) x[n] = cd(grid[((size_t)b * nl + n) * lk + i]);
    const cx<T>* fb = f + (size_t)b * f_stride + (size_t)grp * nt * nl;
    for (int t = 0; t < nt; ++t) {
      cd acc(0, 0);
      for (int n = 0; n < nl; ++n) nrx::cmac(acc, cd(fb[t * nl + n]), x[n]);
      out[((size_t)b * nt + t) * lk + i] = cx<T>(acc);
    }
  }
}

// Mean channel of each PRG (pdsch.py:1125-1127): Hm[b][g][e] = mean over the L symbols and the group's n_k subcarriers
// (contiguous, starting at k0) of H[b][l][k][e], e = (r, t).  One thread per (b, g, e).
__global__ void __launch_bounds__(256)
group_mean_kernel(const cd* __restrict__ H, int L, int K, int E, const int32_t* __restrict__ k0, const int32_t* __restrict__ nk,
                  int G, cd* __restrict__ Hm, int n_batch) {
  const int64_t total = (int64_t)n_batch * G * E;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(gi % E), g = (int)((gi / E) % G), b = (int)(gi / ((int64_t)E * G));
    const cd* src = H + (size_t)b * L * K * E + e;
    double sr = 0, si = 0;
    for (int l = 0; l < L; ++l)
      for (int k = 0; k < nk[g]; ++k) {
        const cd v = src[((size_t)l * K + k0[g] + k) * E];
        sr += v.re;
        si += v.im;
      }
    const double cnt = (double)L * (double)nk[g];
    Hm[gi] = cd(sr / cnt, si / cnt);
  }
}

// Hest[b][l][k][r][p] = sum_t H[b][l][k][r][t] * F[b][k2g[k]][t][p]   (perfect CSI with per-PRG precoders; zero
// where no group covers k)
__global__ void __launch_bounds__(256)
eff_channel_prg_kernel(const cd* __restrict__ H, const cd* __restrict__ F, int64_t f_stride, const int32_t* __restrict__ k2g,
                       int L, int K, int nr, int nt, int nl, cd* __restrict__ out, int64_t total) {
  const int lk = L * K;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(g % nl);
    const int r = (int)((g / nl) % nr);
    const int64_t i = g / ((int64_t)nl * nr);  // b*lk + re
    const int b = (int)(i / lk);
    const int grp = k2g[(int)(i % K)];
    cd acc(0, 0);
    if (grp >= 0) {
      const cd* h = H + ((size_t)i * nr + r) * nt;
      const cd* f = F + (size_t)b * f_stride + (size_t)grp * nt * nl;
      for (int t = 0; t < nt; ++t) nrx::cmac(acc, h[t], f[t * nl + p]);
    }
    out[g] = acc;
  }
}

// ------------------------------------------------------------------------------------------ applyChannel
// out[b][r][lk] = sum_t H[b][lk][r][t] * grid[b][t][lk]        (grid.py:1006-1011)
template <typename T>
__global__ void __launch_bounds__(256)
apply_fd_kernel(const cx<T>* __restrict__ grid, const cx<T>* __restrict__ h, int64_t h_stride, int nt, int nr, int lk,
                cx<T>* __restrict__ out, int n_batch) {
  const int64_t total = (int64_t)n_batch * lk;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / lk), i = (int)(g - (int64_t)b * lk);
    const cx<T>* hb = h + (size_t)b * h_stride + (size_t)i * nr * nt;
    for (int r = 0; r < nr; ++r) {
      cd acc(0, 0);
      for (int t = 0; t < nt; ++t) nrx::cmac(acc, cd(hb[r * nt + t]), cd(grid[((size_t)b * nt + t) * lk + i]));
      out[((size_t)b * nr + r) * lk + i] = cx<T>(acc);
    }
  }
}

// --------------------------------------------------------------------------------------------- equalize
// MMSE per RE (grid.py:669-688):  Ainv = (H^H H + s2 I)^-1,  xhat = Ainv H^H y,  llrScale = 1/Re diag(Ainv).
// The reference goes through pinv/SVD of the same Hermitian positive-definite matrix; here it is a Cholesky
// factorisation in float64, one RE per lane, everything in registers (NR, NL compile-time).
template <typename T, int NR, int NL>
__global__ void __launch_bounds__(256)
mmse_kernel(const cx<T>* __restrict__ rx, const cx<T>* __restrict__ hf, int64_t h_stride,
            const T* __restrict__ noise_var, int nv_stride, int lk, cx<T>* __restrict__ eq, T* __restrict__ scale,
            int n_batch) {
  const int64_t total = (int64_t)n_batch * lk;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / lk), i = (int)(g - (int64_t)b * lk);
    double nv = (double)noise_var[(size_t)b * nv_stride];
    nv = nv > 1e-8 ? nv : 1e-8;  // grid.py:676
    cd H[NR][NL], y[NR];
    const cx<T>* hb = hf + (size_t)b * h_stride + (size_t)i * NR * NL;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      y[r] = cd(rx[((size_t)b * NR + r) * lk + i]);
#pragma unroll
      for (int p = 0; p < NL; ++p) H[r][p] = cd(hb[r * NL + p]);
    }
    cd xh[NL];
    double sc[NL];
    nrx::mmse_solve<NR, NL>(H, y, nv, xh, sc);
#pragma unroll
    for (int p = 0; p < NL; ++p) {
      eq[((size_t)b * NL + p) * lk + i] = cx<T>(xh[p]);
      scale[((size_t)b * NL + p) * lk + i] = (T)sc[p];
    }
  }
}

// Wide-MIMO fallback (5..8 layers, two-codeword PDSCH; any Nr <= 8): the same Cholesky solve as nrx::mmse_solve with
// run-time sizes (local arrays, not unrolled).  Not on the throughput path.
template <typename T>
__global__ void __launch_bounds__(64)
mmse_wide_kernel(const cx<T>* __restrict__ rx, const cx<T>* __restrict__ hf, int64_t h_stride,
                 const T* __restrict__ noise_var, int nv_stride, int nr, int nl, int lk, cx<T>* __restrict__ eq,
                 T* __restrict__ scale, int n_batch) {
  constexpr int M8 = 8;
  const int64_t total = (int64_t)n_batch * lk;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / lk), i = (int)(g - (int64_t)b * lk);
    double nv = (double)noise_var[(size_t)b * nv_stride];
    nv = nv > 1e-8 ? nv : 1e-8;  // grid.py:676
    const cx<T>* hb = hf + (size_t)b * h_stride + (size_t)i * nr * nl;
    cd A[M8][M8], Lm[M8][M8], Mi[M8][M8], z[M8], u[M8];
    double dinv[M8];
    for (int p = 0; p < nl; ++p) {
      cd zz(0, 0);
      for (int r = 0; r < nr; ++r) nrx::cmacc(zz, cd(hb[r * nl + p]), cd(rx[((size_t)b * nr + r) * lk + i]));
      z[p] = zz;
      for (int q = 0; q <= p; ++q) {
        cd a(0, 0);
        for (int r = 0; r < nr; ++r) nrx::cmacc(a, cd(hb[r * nl + q]), cd(hb[r * nl + p]));
        A[p][q] = nrx::conj(a);
      }
      A[p][p].re += nv;
    }
    for (int j = 0; j < nl; ++j) {
      double d = A[j][j].re;
      for (int k = 0; k < j; ++k) d -= nrx::norm2(Lm[j][k]);
      const double ljj = sqrt(d);
      dinv[j] = 1.0 / ljj;
      Lm[j][j] = cd(ljj, 0);
      for (int r = j + 1; r < nl; ++r) {
        cd s2 = A[r][j];
        for (int k = 0; k < j; ++k) {
          s2.re -= Lm[r][k].re * Lm[j][k].re + Lm[r][k].im * Lm[j][k].im;
          s2.im -= Lm[r][k].im * Lm[j][k].re - Lm[r][k].re * Lm[j][k].im;
        }
        Lm[r][j] = s2 * dinv[j];
      }
    }
    for (int c = 0; c < nl; ++c) {
      Mi[c][c] = cd(dinv[c], 0);
      for (int r = c + 1; r < nl; ++r) {
        cd s2(0, 0);
        for (int k = c; k < r; ++k) nrx::cmac(s2, Lm[r][k], Mi[k][c]);
        Mi[r][c] = cd(-s2.re * dinv[r], -s2.im * dinv[r]);
      }
    }
    for (int r = 0; r < nl; ++r) {
      cd s2(0, 0);
      for (int c = 0; c <= r; ++c) nrx::cmac(s2, Mi[r][c], z[c]);
      u[r] = s2;
    }
    for (int p = 0; p < nl; ++p) {
      cd s2(0, 0);
      double dg = 0;
      for (int k = p; k < nl; ++k) {
        nrx::cmacc(s2, Mi[k][p], u[k]);
        dg += nrx::norm2(Mi[k][p]);
      }
      eq[((size_t)b * nl + p) * lk + i] = cx<T>(s2);
      scale[((size_t)b * nl + p) * lk + i] = (T)(1.0 / dg);
    }
  }
}

// ---------------------------------------------------------------------------------------- complex variance
// np.var of a complex array = mean |x - mean(x)|^2 (grid.py:1046, waveform.py:117): per batch item, float64,
// two-level reduction: (sum x, sum |x|^2) per workgroup into acc[b][workgroup][3], summed in workgroup order by
// var_finish_kernel -- no atomics, so the result (and with it the noise scaling of a slot) is reproducible bit for bit.
template <typename T>
__global__ void __launch_bounds__(256)
var_partial_kernel(const cx<T>* __restrict__ x, int64_t n_per, int64_t x_stride, const int32_t* __restrict__ gather,
                   int64_t n_gather, double* __restrict__ acc /* n_batch x gridDim.x x 3 */) {
  const int b = blockIdx.y;
  const int64_t n = gather ? n_gather : n_per;
  double sr = 0, si = 0, s2 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = gather ? (int64_t)gather[i] : i;
    const cx<T> v = x[(size_t)b * x_stride + j];
    sr += (double)v.re;
    si += (double)v.im;
    s2 += (double)v.re * (double)v.re + (double)v.im * (double)v.im;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_xor(sr, o, 64);
    si += __shfl_xor(si, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  __shared__ double red[3][4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = sr; red[1][w] = si; red[2][w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0, d = 0;
    for (int k = 0; k < 4; ++k) { a += red[0][k]; c += red[1][k]; d += red[2][k]; }
    double* o = acc + ((size_t)b * gridDim.x + blockIdx.x) * 3;
    o[0] = a;
    o[1] = c;
    o[2] = d;
  }
}

// sigma[b] = sqrt(var * mult / snr_lin[b])  and  noise_var_out[b] = sigma^2 * nv_mult (both optional outputs)
template <typename T>
__global__ void var_finish_kernel(const double* __restrict__ acc, int n_part, double n, int n_batch,
                                  T* __restrict__ var_out, const double* __restrict__ snr_lin, int snr_stride, double mult,
                                  T* __restrict__ sigma_out, T* __restrict__ nv_out, double nv_mult) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_batch) return;
  double sr = 0, si = 0, s2 = 0;
  for (int k = 0; k < n_part; ++k) {
    const double* p = acc + ((size_t)b * n_part + k) * 3;
    sr += p[0];
    si += p[1];
    s2 += p[2];
  }
  const double mr = sr / n, mi = si / n;
  double var = s2 / n - (mr * mr + mi * mi);
  if (var < 0) var = 0;
  if (var_out) var_out[b] = (T)var;
  if (snr_lin) {
    const double sg = sqrt(var * mult / snr_lin[(size_t)b * snr_stride]);
    if (sigma_out) sigma_out[b] = (T)sg;
    if (nv_out) nv_out[b] = (T)(sg * sg * nv_mult);
  }
}

// The same with one wave per item for many partials (lane l sums partials l, l+64, ...; then a fixed shuffle tree: reproducible).
__global__ void __launch_bounds__(64)
var_finish_wave_kernel(const double* __restrict__ acc, int n_part, double n, double* __restrict__ var_out,
                       const double* __restrict__ snr_lin, int snr_stride, double mult, double* __restrict__ sigma_out,
                       double* __restrict__ nv_out, double nv_mult) {
  const int b = blockIdx.x;
  double sr = 0, si = 0, s2 = 0;
  for (int k = threadIdx.x; k < n_part; k += 64) {
    const double* p = acc + ((size_t)b * n_part + k) * 3;
    sr += p[0];
    si += p[1];
    s2 += p[2];
  }
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_xor(sr, o, 64);
    si += __shfl_xor(si, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (threadIdx.x) return;
  const double mr = sr / n, mi = si / n;
  double var = s2 / n - (mr * mr + mi * mi);
  if (var < 0) var = 0;
  if (var_out) var_out[b] = var;
  if (snr_lin) {
    const double sg = sqrt(var * mult / snr_lin[(size_t)b * snr_stride]);
    if (sigma_out) sigma_out[b] = sg;
    if (nv_out) nv_out[b] = sg * sg * nv_mult;
  }
}

// ------------------------------------------------------------------------------------------------- noise
// out = x + (sigma[b]/sqrt(2)) * z   with z = standard-normal pairs supplied by the caller (host PCG64 stream
// in parity mode: random.py:203 awgn = normal(0, sigma/sqrt(2), shape+(2,))).
template <typename T>
__global__ void __launch_bounds__(256)
add_noise_kernel(const cx<T>* __restrict__ x, const cx<T>* __restrict__ z, const T* __restrict__ sigma,
                 int sigma_stride, int64_t n_per, cx<T>* __restrict__ out, int n_batch) {
  const int64_t total = (int64_t)n_batch * n_per;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / n_per);
    const double s = (double)sigma[(size_t)b * sigma_stride] / 1.4142135623730951;
    const cx<T> v = x[g], w = z[g];
    out[g] = cx<T>((T)((double)v.re + s * (double)w.re), (T)((double)v.im + s * (double)w.im));
  }
}

using nrx::philox4x32;

template <typename T, bool F64N>
__global__ void __launch_bounds__(256)
awgn_philox_kernel(const cx<T>* __restrict__ x, const T* __restrict__ sigma, int sigma_stride, int64_t n_per,
                   cx<T>* __restrict__ out, int n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset,
                   const int64_t* __restrict__ item_ids) {
  const int64_t total = (int64_t)n_batch * n_per;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / n_per);
    const int64_t e = g - (int64_t)b * n_per;
    out[g] = nrx::awgn_add<T, F64N>(x[g], (double)sigma[(size_t)b * sigma_stride], seed, stream_id,
                               (uint64_t)(item_ids ? item_ids[b] : batch_offset + b), e);
  }
}

// Synthetic transport blocks for the throughput mode: bit e of item b = one Philox output bit, keyed like awgn.
__global__ void __launch_bounds__(256)
random_bits_kernel(uint8_t* __restrict__ out, int64_t n_per, int n_batch, uint64_t seed, uint64_t stream_id,
                   int64_t batch_offset, const int64_t* __restrict__ item_ids) {
  const int64_t words = (n_per + 127) / 128;  // one Philox call = 128 bits
  const int64_t total = (int64_t)n_batch * words;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(g / words);
    const int64_t w = g - (int64_t)b * words;
    const uint64_t item = (uint64_t)(item_ids ? item_ids[b] : batch_offset + b);
    uint32_t c[4] = {(uint32_t)w, (uint32_t)((uint64_t)w >> 32), (uint32_t)item, (uint32_t)(item >> 32) ^ (uint32_t)stream_id};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint8_t* dst = out + (size_t)b * n_per + w * 128;
    const int64_t lim = n_per - w * 128 < 128 ? n_per - w * 128 : 128;
    if (lim == 128 && ((uintptr_t)dst & 7u) == 0) {
      // 8 bits -> 8 bytes per store: nibble * 0x00204081 puts bit k of the nibble at bit 8k
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const uint32_t byte = (c[q >> 2] >> (8 * (q & 3))) & 0xffu;
        uint2 v;
        v.x = ((byte & 15u) * 0x00204081u) & 0x01010101u;
        v.y = ((byte >> 4) * 0x00204081u) & 0x01010101u;
        *reinterpret_cast<uint2*>(dst + 8 * q) = v;
      }
    } else {
      for (int i = 0; i < lim; ++i) dst[i] = (uint8_t)((c[i >> 5] >> (i & 31)) & 1u);
    }
  }
}

template <typename T, int NR>
int32_t mmse_dispatch_nl(int nl, dim3 grid, hipStream_t st, const cx<T>* rx, const cx<T>* hf, int64_t h_stride,
                         const T* nv, int nv_stride, int lk, cx<T>* eq, T* sc, int n_batch) {
#define NRX_MMSE_CASE(NL)                                                                                          \
  case NL:                                                                                                         \
    hipLaunchKernelGGL((mmse_kernel<T, NR, NL>), grid, dim3(256), 0, st, rx, hf, h_stride, nv, nv_stride, lk, eq, sc, \
                       n_batch);                                                                                   \
    return NRX_OK;
  switch (nl) {
    NRX_MMSE_CASE(1)
    NRX_MMSE_CASE(2)
    NRX_MMSE_CASE(3)
    NRX_MMSE_CASE(4)
    default: return NRX_E_UNSUPPORTED;
  }
#undef NRX_MMSE_CASE
}

template <typename T>
int32_t mmse_entry(const void* rx, const void* hf, int64_t h_stride, const void* noise_var, int32_t nv_stride,
                   int32_t nr, int32_t nl, int32_t lk, void* eq, void* scale, int32_t n_batch, void* stream) {
  NRX_REQUIRE(rx && hf && noise_var && eq && scale, NRX_E_ARG, "nrx_mmse_equalize: NULL buffer");
  NRX_REQUIRE(nr >= 1 && nl >= 1 && lk >= 0 && n_batch >= 0, NRX_E_ARG, "nrx_mmse_equalize: bad sizes");
  if (lk == 0 || n_batch == 0) return NRX_OK;
  const dim3 grid(nrx::stream_grid((long)lk * n_batch, 256));
  hipStream_t st = (hipStream_t)stream;
  int32_t rc = NRX_E_UNSUPPORTED;
  switch (nr) {
    case 1: rc = mmse_dispatch_nl<T, 1>(nl, grid, st, (const cx<T>*)rx, (const cx<T>*)hf, h_stride, (const T*)noise_var, nv_stride, lk, (cx<T>*)eq, (T*)scale, n_batch); break;
    case 2: rc = mmse_dispatch_nl<T, 2>(nl, grid, st, (const cx<T>*)rx, (const cx<T>*)hf, h_stride, (const T*)noise_var, nv_stride, lk, (cx<T>*)eq, (T*)scale, n_batch); break;
    case 4: rc = mmse_dispatch_nl<T, 4>(nl, grid, st, (const cx<T>*)rx, (const cx<T>*)hf, h_stride, (const T*)noise_var, nv_stride, lk, (cx<T>*)eq, (T*)scale, n_batch); break;
    case 8: rc = mmse_dispatch_nl<T, 8>(nl, grid, st, (const cx<T>*)rx, (const cx<T>*)hf, h_stride, (const T*)noise_var, nv_stride, lk, (cx<T>*)eq, (T*)scale, n_batch); break;
    default: break;
  }
  if (rc == NRX_E_UNSUPPORTED && nr <= 8 && nl <= 8) {   // 5..8 layers (or an odd antenna count): run-time sizes
    hipLaunchKernelGGL(mmse_wide_kernel<T>, dim3(nrx::stream_grid((long)lk * n_batch, 64)), dim3(64), 0, st, (const cx<T>*)rx,
                       (const cx<T>*)hf, h_stride, (const T*)noise_var, nv_stride, nr, nl, lk, (cx<T>*)eq, (T*)scale, n_batch);
    rc = NRX_OK;
  }
  NRX_REQUIRE(rc != NRX_E_UNSUPPORTED, NRX_E_UNSUPPORTED,
              "nrx_mmse_equalize: (Nr=%d, layers=%d) not built (at most 8 x 8)", nr, nl);
  NRX_CHECK_LAUNCH("nrx_mmse_equalize");
  return rc;
}

}  // namespace

#define NRX_T2(NAME, IMPL)                                      \
  extern "C" int32_t NAME##_f32 IMPL(float) extern "C" int32_t NAME##_f64 IMPL(double)

extern "C" int32_t nrx_precode_f32(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk, void* out, int32_t n_batch, void* stream);
extern "C" int32_t nrx_precode_f64(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk, void* out, int32_t n_batch, void* stream);

template <typename T>
static int32_t precode_entry(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk,
                             void* out, int32_t n_batch, void* stream) {
  NRX_REQUIRE(grid && f && out, NRX_E_ARG, "nrx_precode: NULL buffer");
  NRX_REQUIRE(nl >= 1 && nl <= 8 && nt >= 1 && lk >= 0 && n_batch >= 0, NRX_E_ARG, "nrx_precode: bad sizes (layers 1..8)");
  if (lk == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(precode_kernel<T>, dim3(nrx::stream_grid((long)lk * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)grid, (const cx<T>*)f, f_stride, nl, nt, lk, (cx<T>*)out, n_batch);
  NRX_CHECK_LAUNCH("nrx_precode");
  return NRX_OK;
}
extern "C" int32_t nrx_precode_f32(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk, void* out, int32_t n_batch, void* stream) { return precode_entry<float>(grid, f, f_stride, nl, nt, lk, out, n_batch, stream); }
extern "C" int32_t nrx_precode_f64(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk, void* out, int32_t n_batch, void* stream) { return precode_entry<double>(grid, f, f_stride, nl, nt, lk, out, n_batch, stream); }

template <typename T>
static int32_t apply_fd_entry(const void* grid, const void* h, int64_t h_stride, int32_t nt, int32_t nr, int32_t lk,
                              void* out, int32_t n_batch, void* stream) {
  NRX_REQUIRE(grid && h && out, NRX_E_ARG, "nrx_apply_channel_fd: NULL buffer");
  NRX_REQUIRE(nt >= 1 && nr >= 1 && lk >= 0 && n_batch >= 0, NRX_E_ARG, "nrx_apply_channel_fd: bad sizes");
  if (lk == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(apply_fd_kernel<T>, dim3(nrx::stream_grid((long)lk * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)grid, (const cx<T>*)h, h_stride, nt, nr, lk, (cx<T>*)out, n_batch);
  NRX_CHECK_LAUNCH("nrx_apply_channel_fd");
  return NRX_OK;
}
extern "C" int32_t nrx_apply_channel_fd_f32(const void* grid, const void* h, int64_t h_stride, int32_t nt, int32_t nr, int32_t lk, void* out, int32_t n_batch, void* stream) { return apply_fd_entry<float>(grid, h, h_stride, nt, nr, lk, out, n_batch, stream); }
extern "C" int32_t nrx_apply_channel_fd_f64(const void* grid, const void* h, int64_t h_stride, int32_t nt, int32_t nr, int32_t lk, void* out, int32_t n_batch, void* stream) { return apply_fd_entry<double>(grid, h, h_stride, nt, nr, lk, out, n_batch, stream); }

extern "C" int32_t nrx_mmse_equalize_f32(const void* rx, const void* hf, int64_t h_stride, const void* noise_var, int32_t nv_stride, int32_t nr, int32_t nl, int32_t lk, void* eq, void* scale, int32_t n_batch, void* stream) { return mmse_entry<float>(rx, hf, h_stride, noise_var, nv_stride, nr, nl, lk, eq, scale, n_batch, stream); }
extern "C" int32_t nrx_mmse_equalize_f64(const void* rx, const void* hf, int64_t h_stride, const void* noise_var, int32_t nv_stride, int32_t nr, int32_t nl, int32_t lk, void* eq, void* scale, int32_t n_batch, void* stream) { return mmse_entry<double>(rx, hf, h_stride, noise_var, nv_stride, nr, nl, lk, eq, scale, n_batch, stream); }

// Per-item complex variance + (optionally) the noise std of addNoise(snrDb, useRxPower=True).
template <typename T>
static int32_t noise_level_entry(const void* x, int64_t n_per, int64_t x_stride, const int32_t* gather, int64_t n_gather,
                                 int32_t n_batch, double* acc_ws, void* var_out, const double* snr_lin, int32_t snr_stride,
                                 double mult, void* sigma_out, void* nv_out, double nv_mult, void* stream) {
  NRX_REQUIRE(x && acc_ws, NRX_E_ARG, "nrx_noise_level: NULL buffer (acc_ws = 192 doubles per batch item)");
  NRX_REQUIRE(n_per > 0 && n_batch >= 0, NRX_E_ARG, "nrx_noise_level: bad sizes");
  if (n_batch == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = gather ? n_gather : n_per;
  int gx = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(var_partial_kernel<T>, dim3(gx, n_batch), dim3(256), 0, st, (const cx<T>*)x, n_per, x_stride, gather,
                     n_gather, acc_ws);
  hipLaunchKernelGGL(var_finish_kernel<T>, dim3((n_batch + 63) / 64), dim3(64), 0, st, acc_ws, gx, (double)n, n_batch,
                     (T*)var_out, snr_lin, snr_stride, mult, (T*)sigma_out, (T*)nv_out, nv_mult);
  NRX_CHECK_LAUNCH("nrx_noise_level");
  return NRX_OK;
}
// The second half of nrx_noise_level_f64 on partial sums that another kernel left (nrx_apply_td_paths_pow_f64): acc (n_batch,
// n_part, 3) = (sum re, sum im, sum |x|^2) over `count` samples per item.
extern "C" int32_t nrx_noise_level_finish_f64(const double* acc, int32_t n_part, int64_t count, int32_t n_batch, void* var_out,
                                              const double* snr_lin, int32_t snr_stride, double mult, void* sigma_out, void* nv_out,
                                              double nv_mult, void* stream) {
  NRX_REQUIRE(acc && n_part >= 1 && count > 0 && n_batch >= 0, NRX_E_ARG, "nrx_noise_level_finish: bad argument");
  if (n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(var_finish_wave_kernel, dim3(n_batch), dim3(64), 0, (hipStream_t)stream, acc, n_part, (double)count,
                     (double*)var_out, snr_lin, snr_stride, mult, (double*)sigma_out, (double*)nv_out, nv_mult);
  NRX_CHECK_LAUNCH("nrx_noise_level_finish");
  return NRX_OK;
}
extern "C" int32_t nrx_noise_level_f32(const void* x, int64_t n_per, int64_t x_stride, const int32_t* gather, int64_t n_gather, int32_t n_batch, double* acc_ws, void* var_out, const double* snr_lin, int32_t snr_stride, double mult, void* sigma_out, void* nv_out, double nv_mult, void* stream) { return noise_level_entry<float>(x, n_per, x_stride, gather, n_gather, n_batch, acc_ws, var_out, snr_lin, snr_stride, mult, sigma_out, nv_out, nv_mult, stream); }
extern "C" int32_t nrx_noise_level_f64(const void* x, int64_t n_per, int64_t x_stride, const int32_t* gather, int64_t n_gather, int32_t n_batch, double* acc_ws, void* var_out, const double* snr_lin, int32_t snr_stride, double mult, void* sigma_out, void* nv_out, double nv_mult, void* stream) { return noise_level_entry<double>(x, n_per, x_stride, gather, n_gather, n_batch, acc_ws, var_out, snr_lin, snr_stride, mult, sigma_out, nv_out, nv_mult, stream); }

template <typename T>
static int32_t add_noise_entry(const void* x, const void* z, const void* sigma, int32_t sigma_stride, int64_t n_per,
                               void* out, int32_t n_batch, void* stream) {
  NRX_REQUIRE(x && z && sigma && out, NRX_E_ARG, "nrx_add_noise: NULL buffer");
  if (n_per == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(add_noise_kernel<T>, dim3(nrx::stream_grid((long)n_per * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)x, (const cx<T>*)z, (const T*)sigma, sigma_stride, n_per,
                     (cx<T>*)out, n_batch);
  NRX_CHECK_LAUNCH("nrx_add_noise");
  return NRX_OK;
}
extern "C" int32_t nrx_add_noise_f32(const void* x, const void* z, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out, int32_t n_batch, void* stream) { return add_noise_entry<float>(x, z, sigma, sigma_stride, n_per, out, n_batch, stream); }
extern "C" int32_t nrx_add_noise_f64(const void* x, const void* z, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out, int32_t n_batch, void* stream) { return add_noise_entry<double>(x, z, sigma, sigma_stride, n_per, out, n_batch, stream); }

template <typename T>
static int32_t awgn_entry(const void* x, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out,
                          int32_t n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids,
                          void* stream) {
  NRX_REQUIRE(x && sigma && out, NRX_E_ARG, "nrx_awgn: NULL buffer");
  if (n_per == 0 || n_batch == 0) return NRX_OK;
  auto kern = nrx::noise_f64() ? awgn_philox_kernel<T, true> : awgn_philox_kernel<T, false>;
  hipLaunchKernelGGL(kern, dim3(nrx::stream_grid((long)n_per * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)x, (const T*)sigma, sigma_stride, n_per, (cx<T>*)out, n_batch,
                     seed, stream_id, batch_offset, item_ids);
  NRX_CHECK_LAUNCH("nrx_awgn");
  return NRX_OK;
}
extern "C" int32_t nrx_awgn_f32(const void* x, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out, int32_t n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* stream) { return awgn_entry<float>(x, sigma, sigma_stride, n_per, out, n_batch, seed, stream_id, batch_offset, item_ids, stream); }
extern "C" int32_t nrx_awgn_f64(const void* x, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out, int32_t n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* stream) { return awgn_entry<double>(x, sigma, sigma_stride, n_per, out, n_batch, seed, stream_id, batch_offset, item_ids, stream); }

extern "C" int32_t nrx_random_bits(uint8_t* out, int64_t n_per, int32_t n_batch, uint64_t seed, uint64_t stream_id,
                                   int64_t batch_offset, const int64_t* item_ids, void* stream) {
  NRX_REQUIRE(out && n_per >= 0 && n_batch >= 0, NRX_E_ARG, "nrx_random_bits: bad argument");
  if (n_per == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(random_bits_kernel, dim3(nrx::stream_grid(((long)n_per + 127) / 128 * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, out, n_per, n_batch, seed, stream_id, batch_offset, item_ids);
  NRX_CHECK_LAUNCH("nrx_random_bits");
  return NRX_OK;
}


// ------------------------------------------------------------------------------------------- per-PRG precoding
template <typename T>
static int32_t precode_prg_entry(const void* grid, const void* f, int64_t f_stride, const int32_t* k2g, int32_t nl, int32_t nt,
                                 int32_t L, int32_t K, void* out, int32_t n_batch, void* stream) {
  NRX_REQUIRE(grid && f && k2g && out, NRX_E_ARG, "nrx_precode_prg: NULL buffer");
  NRX_REQUIRE(nl >= 1 && nl <= 8 && nt >= 1 && L >= 0 && K >= 0 && n_batch >= 0, NRX_E_ARG, "nrx_precode_prg: bad sizes (layers 1..8)");
  if ((int64_t)L * K == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(precode_prg_kernel<T>, dim3(nrx::stream_grid((long)L * K * n_batch, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)grid, (const cx<T>*)f, f_stride, k2g, nl, nt, L, K, (cx<T>*)out, n_batch);
  NRX_CHECK_LAUNCH("nrx_precode_prg");
  return NRX_OK;
}
extern "C" int32_t nrx_precode_prg_f32(const void* grid, const void* f, int64_t f_stride, const int32_t* k2g, int32_t nl, int32_t nt, int32_t L, int32_t K, void* out, int32_t n_batch, void* stream) { return precode_prg_entry<float>(grid, f, f_stride, k2g, nl, nt, L, K, out, n_batch, stream); }
extern "C" int32_t nrx_precode_prg_f64(const void* grid, const void* f, int64_t f_stride, const int32_t* k2g, int32_t nl, int32_t nt, int32_t L, int32_t K, void* out, int32_t n_batch, void* stream) { return precode_prg_entry<double>(grid, f, f_stride, k2g, nl, nt, L, K, out, n_batch, stream); }

extern "C" int32_t nrx_group_mean_f64(const void* H, int32_t n_items, int32_t L, int32_t K, int32_t E, const int32_t* k0,
                                      const int32_t* nk, int32_t G, void* Hm, void* stream) {
  NRX_REQUIRE(H && k0 && nk && Hm, NRX_E_ARG, "nrx_group_mean: NULL buffer");
  NRX_REQUIRE(L >= 1 && K >= 1 && E >= 1 && G >= 0 && n_items >= 0, NRX_E_ARG, "nrx_group_mean: bad sizes");
  if (G == 0 || n_items == 0) return NRX_OK;
  hipLaunchKernelGGL(group_mean_kernel, dim3(nrx::stream_grid((long)n_items * G * E, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const cd*)H, L, K, E, k0, nk, G, (cd*)Hm, n_items);
  NRX_CHECK_LAUNCH("nrx_group_mean");
  return NRX_OK;
}

extern "C" int32_t nrx_effective_channel_prg_f64(const void* H, const void* F, int64_t f_stride, const int32_t* k2g,
                                                 int32_t n_items, int32_t L, int32_t K, int32_t n_rx, int32_t n_tx,
                                                 int32_t n_layers, void* out, void* stream) {
  NRX_REQUIRE(H && F && k2g && out, NRX_E_ARG, "nrx_effective_channel_prg: NULL buffer");
  const int64_t total = (int64_t)n_items * L * K * n_rx * n_layers;
  if (total == 0) return NRX_OK;
  hipLaunchKernelGGL(eff_channel_prg_kernel, dim3(nrx::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const cd*)H, (const cd*)F, f_stride, k2g, L, K, n_rx, n_tx, n_layers, (cd*)out, total);
  NRX_CHECK_LAUNCH("nrx_effective_channel_prg");
  return NRX_OK;
}
