// CSI feedback: post-MMSE SINR of every (codebook entry, resource element, layer) (gfx950).
//
// Replaces reference csifeedback.py:419-433 CsiReport.getSINR -- the inner loop of the PMI / rank search
// (csifeedback.py:450-536): heff = H W, gamma_l = 1 / (noiseVar * [(heff^H heff + noiseVar I)^-1]_ll) - 1.  The reference
// reaches the diagonal of that inverse through an SVD of heff; it is the same Hermitian positive-definite matrix the
// equaliser factorises, so here it is a Cholesky factorisation per lane, [A^-1]_ll = |L^-1 e_l|^2.
// One lane per (entry, RE); the codebook (<= 32 x 8 per entry) and the channel rows are re-read through L2.
#include "nrx_common.h"
#include "nrx_cplx.h"

namespace {
using nrx::cx;
typedef cx<double> cd;
constexpr int MAXR = 8, MAXL = 8;

__global__ void __launch_bounds__(128)
csi_sinr_kernel(const cd* __restrict__ h, const cd* __restrict__ w, int n_re, int nr, int nt, int n_cb, int nl, double nv,
                double* __restrict__ out) {
  const int64_t total = (int64_t)n_cb * n_re;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int cb = (int)(gi / n_re), re = (int)(gi - (int64_t)cb * n_re);
    const cd* hh = h + (size_t)re * nr * nt;
    const cd* ww = w + (size_t)cb * nt * nl;
    cd he[MAXR][MAXL];
    for (int r = 0; r < nr; ++r)
      for (int l = 0; l < nl; ++l) {
        cd a(0, 0);
        for (int t = 0; t < nt; ++t) nrx::cmac(a, hh[r * nt + t], ww[t * nl + l]);
        he[r][l] = a;
      }
    // A = heff^H heff + nv I, lower triangle; Cholesky in place
    cd L[MAXL][MAXL];
    for (int p = 0; p < nl; ++p)
      for (int q = 0; q <= p; ++q) {
        cd a(0, 0);
        for (int r = 0; r < nr; ++r) nrx::cmacc(a, he[r][q], he[r][p]);   // conj(he[r][q]) he[r][p] = A[q][p]
        L[p][q] = nrx::conj(a);
      }
    for (int p = 0; p < nl; ++p) L[p][p].re += nv;
    double dinv[MAXL];
    for (int j = 0; j < nl; ++j) {
      double d = L[j][j].re;
      for (int k = 0; k < j; ++k) d -= nrx::norm2(L[j][k]);
      const double ljj = sqrt(d);
      dinv[j] = 1.0 / ljj;
      for (int r = j + 1; r < nl; ++r) {
        cd s = L[r][j];
        for (int k = 0; k < j; ++k) {
          s.re -= L[r][k].re * L[j][k].re + L[r][k].im * L[j][k].im;
          s.im -= L[r][k].im * L[j][k].re - L[r][k].re * L[j][k].im;
        }
        L[r][j] = cd(s.re * dinv[j], s.im * dinv[j]);
      }
    }
    for (int l = 0; l < nl; ++l) {       // y = L^-1 e_l (zero above l), [A^-1]_ll = |y|^2
      cd y[MAXL];
      double acc = dinv[l] * dinv[l];
      y[l] = cd(dinv[l], 0);
      for (int r = l + 1; r < nl; ++r) {
        cd s(0, 0);
        for (int k = l; k < r; ++k) {
          s.re -= L[r][k].re * y[k].re - L[r][k].im * y[k].im;
          s.im -= L[r][k].re * y[k].im + L[r][k].im * y[k].re;
        }
        y[r] = cd(s.re * dinv[r], s.im * dinv[r]);
        acc += nrx::norm2(y[r]);
      }
      out[((size_t)cb * n_re + re) * nl + l] = 1.0 / (nv * acc) - 1.0;
    }
  }
}
}  // namespace

extern "C" int32_t nrx_csi_sinr_f64(const void* h, int32_t n_re, int32_t nr, int32_t nt, const void* w, int32_t n_cb,
                                    int32_t nl, double noise_var, double* sinr, void* stream) {
  NRX_REQUIRE(h && w && sinr, NRX_E_ARG, "nrx_csi_sinr: NULL buffer");
  NRX_REQUIRE(n_re >= 0 && n_cb >= 0 && nt >= 1, NRX_E_ARG, "nrx_csi_sinr: bad sizes");
  NRX_REQUIRE(nr >= 1 && nr <= MAXR && nl >= 1 && nl <= MAXL, NRX_E_UNSUPPORTED,
              "nrx_csi_sinr: built for up to %d rx antennas and %d layers (got %d, %d)", MAXR, MAXL, nr, nl);
  NRX_REQUIRE(nl <= nr, NRX_E_ARG, "nrx_csi_sinr: %d layers on %d rx antennas", nl, nr);
  NRX_REQUIRE(noise_var > 0.0, NRX_E_ARG, "nrx_csi_sinr: the noise variance must be positive");
  if (n_re == 0 || n_cb == 0) return NRX_OK;
  hipLaunchKernelGGL(csi_sinr_kernel, dim3(nrx::stream_grid((long)n_cb * n_re, 128)), dim3(128), 0, (hipStream_t)stream,
                     (const cd*)h, (const cd*)w, n_re, nr, nt, n_cb, nl, noise_var, sinr);
  NRX_CHECK_LAUNCH("nrx_csi_sinr");
  return NRX_OK;
}
