// DMRS least-squares channel estimation with CDM de-spreading and linear interpolation (gfx950).
//
// Replaces reference grid.py:874-975 estimateChannelLS(polarInt=False, kernel='linear') ->
// grid.py:740-871 estimateChannelLsEx (channel part) + utils.py:26-35 interpolate('linear').
// One lane per (subcarrier, rx antenna, port); pilots/received pilots are re-read from L2 (each is used by the
// <= 2*k_cdm*spacing neighbouring subcarriers).  HBM-bound on the (L,K,Nr,P) output write.
#include "nrx_common.h"
#include "nrx_cplx.h"
#include "nrx_mmse.h"

namespace {
using nrx::cx;
typedef cx<double> cd;

struct ChestGeom {
  int32_t n_ds;        // DMRS symbols in the slot
  int32_t ds[8];       // their symbol indices
  int32_t l_cdm, k_cdm;
  int32_t n_k;         // pilots per port per DMRS symbol
  int32_t L, K, nr, P;
};

template <typename T>
__global__ void __launch_bounds__(256)
chest_ls_kernel(const cx<T>* __restrict__ rx, const cx<T>* __restrict__ pilots, const int32_t* __restrict__ pil_set,
                const int32_t* __restrict__ port_ks, ChestGeom g, cx<T>* __restrict__ hest, int n_batch, int hk_only) {
  const int rp = g.nr * g.P;
  const int64_t per = (int64_t)g.K * rp;
  const int64_t total = (int64_t)n_batch * per;
  const int n_j = g.n_k / g.k_cdm;      // CDM groups along frequency
  const int n_g = g.n_ds / g.l_cdm;     // estimates along time
  const int cdm = g.l_cdm * g.k_cdm;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(gi / per);
    const int64_t e = gi - (int64_t)b * per;
    const int k = (int)(e / rp);
    const int r = (int)(e - (int64_t)k * rp) / g.P, p = (int)(e - (int64_t)k * rp) % g.P;
    const int32_t* ks = port_ks + (size_t)p * g.n_k;
    const cx<T>* pil = pilots + ((size_t)(pil_set ? pil_set[b] : 0) * g.P + p) * g.n_ds * g.n_k;
    const cx<T>* rxb = rx + ((size_t)b * g.nr + r) * g.L * g.K;
    auto centre = [&](int j) {  // mean subcarrier index of CDM group j (grid.py:795-796)
      double s = 0;
      for (int q = 0; q < g.k_cdm; ++q) s += (double)ks[j * g.k_cdm + q];
      return s / (double)g.k_cdm;
    };
    // j = clip(searchsorted(centres, k, 'left'), 1, n_j-1): segment [j-1, j] inter/extrapolates k
    int j = 1;
    if (n_j > 1) {
      int lo = 0, hi = n_j;  // first index with centre >= k
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (centre(mid) < (double)k) lo = mid + 1; else hi = mid;
      }
      j = lo < 1 ? 1 : (lo > n_j - 1 ? n_j - 1 : lo);
    }
    cd hk[4];  // estimate at subcarrier k for each time group (n_g <= 4)
    for (int tg = 0; tg < n_g; ++tg) {
      auto group_mean = [&](int jj) {  // LS estimates rx/pilot averaged over the CDM group (grid.py:775-793)
        cd s(0, 0);
        for (int ll = 0; ll < g.l_cdm; ++ll) {
          const int di = tg * g.l_cdm + ll;
          for (int q = 0; q < g.k_cdm; ++q) {
            const int kk = ks[jj * g.k_cdm + q];
            s = s + nrx::cdiv(cd(rxb[(size_t)g.ds[di] * g.K + kk]), cd(pil[(size_t)di * g.n_k + jj * g.k_cdm + q]));
          }
        }
        return cd(s.re / (double)cdm, s.im / (double)cdm);
      };
      if (n_j == 1) { hk[tg] = group_mean(0); continue; }
      const cd y0 = group_mean(j - 1), y1 = group_mean(j);
      const double x0 = centre(j - 1), x1 = centre(j);
      const cd sl((y1.re - y0.re) / (x1 - x0), (y1.im - y0.im) / (x1 - x0));
      hk[tg] = cd(sl.re * ((double)k - x0) + y0.re, sl.im * ((double)k - x0) + y0.im);
    }
    if (hk_only) {   // fused path: keep the per-time-group estimates, the equaliser interpolates along symbols
      for (int tg = 0; tg < n_g; ++tg) hest[((size_t)b * n_g + tg) * per + e] = cx<T>(hk[tg]);
      continue;
    }
    // along symbols (grid.py:853-866): repeat a single estimate, else linear inter/extrapolation
    cx<T>* out = hest + (size_t)b * g.L * per + e;
    for (int l = 0; l < g.L; ++l) {
      cd v;
      if (n_g == 1) v = hk[0];
      else {
        auto lc = [&](int tg) {
          double s = 0;
          for (int ll = 0; ll < g.l_cdm; ++ll) s += (double)g.ds[tg * g.l_cdm + ll];
          return s / (double)g.l_cdm;
        };
        int tj = 0;
        while (tj < n_g && lc(tj) < (double)l) ++tj;
        tj = tj < 1 ? 1 : (tj > n_g - 1 ? n_g - 1 : tj);
        const double x0 = lc(tj - 1), x1 = lc(tj);
        const cd y0 = hk[tj - 1], y1 = hk[tj];
        const cd sl((y1.re - y0.re) / (x1 - x0), (y1.im - y0.im) / (x1 - x0));
        v = cd(sl.re * ((double)l - x0) + y0.re, sl.im * ((double)l - x0) + y0.im);
      }
      out[(size_t)l * per] = cx<T>(v);
    }
  }
}

template <typename T>
int32_t chest_entry(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                    const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                    int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream) {
  NRX_REQUIRE(rx && pilots && port_ks && dmrs_syms && hest, NRX_E_ARG, "nrx_chest_ls: NULL buffer");
  NRX_REQUIRE(n_ds >= 1 && n_ds <= 8 && l_cdm >= 1 && k_cdm >= 1 && n_k >= 1, NRX_E_ARG, "nrx_chest_ls: bad DMRS geometry");
  NRX_REQUIRE(n_k % k_cdm == 0 && n_ds % l_cdm == 0, NRX_E_UNSUPPORTED, "nrx_chest_ls: Partial CDMs are not supported in this version.");
  NRX_REQUIRE(n_ds / l_cdm <= 4, NRX_E_UNSUPPORTED, "nrx_chest_ls: more than 4 DMRS time groups");
  NRX_REQUIRE(L >= 1 && K >= 1 && nr >= 1 && P >= 1 && n_batch >= 0, NRX_E_ARG, "nrx_chest_ls: bad sizes");
  if (n_batch == 0) return NRX_OK;
  ChestGeom g;
  g.n_ds = n_ds;
  for (int i = 0; i < n_ds; ++i) {
    NRX_REQUIRE(dmrs_syms[i] >= 0 && dmrs_syms[i] < L, NRX_E_ARG, "nrx_chest_ls: DMRS symbol index out of range");
    g.ds[i] = dmrs_syms[i];
  }
  g.l_cdm = l_cdm; g.k_cdm = k_cdm; g.n_k = n_k; g.L = L; g.K = K; g.nr = nr; g.P = P;
  hipLaunchKernelGGL(chest_ls_kernel<T>, dim3(nrx::stream_grid((long)n_batch * K * nr * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)rx, (const cx<T>*)pilots, pil_set, port_ks, g, (cx<T>*)hest, n_batch, 0);
  NRX_CHECK_LAUNCH("nrx_chest_ls");
  return NRX_OK;
}

// Estimate + equalise without materialising the (L, K, Nr, P) estimate: chest_ls_kernel leaves the frequency-
// interpolated estimates of the (<= 2) DMRS time groups in `hk`; one thread per (item, subcarrier) then walks the L
// symbols, forms H_l with the same inter/extrapolation expressions as chest_ls_kernel (grid.py:853-866) and runs
// the same MMSE solve as mmse_kernel -- results are identical to nrx_chest_ls + nrx_mmse_equalize.
template <int NR, int NL>
__global__ void __launch_bounds__(128)
mmse_interp_kernel(const cd* __restrict__ rx, const cd* __restrict__ hk, ChestGeom g, const double* __restrict__ noise_var,
                   int nv_stride, cd* __restrict__ eq, double* __restrict__ scale, int n_batch) {
  const int n_g = g.n_ds / g.l_cdm;
  const int64_t total = (int64_t)n_batch * g.K;
  const int64_t lk = (int64_t)g.L * g.K;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(gi / g.K), k = (int)(gi - (int64_t)b * g.K);
    double nv = noise_var[(size_t)b * nv_stride];
    nv = nv > 1e-8 ? nv : 1e-8;  // grid.py:676
    cd y0[NR][NL], sl[NR][NL];
    double x0 = 0.0;
    {
      const cd* h0 = hk + (((size_t)b * n_g + 0) * g.K + k) * NR * NL;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int p = 0; p < NL; ++p) y0[r][p] = h0[r * NL + p];
      if (n_g == 2) {
        auto lc = [&](int tg) {
          double s = 0;
          for (int ll = 0; ll < g.l_cdm; ++ll) s += (double)g.ds[tg * g.l_cdm + ll];
          return s / (double)g.l_cdm;
        };
        x0 = lc(0);
        const double x1 = lc(1);
        const cd* h1 = hk + (((size_t)b * n_g + 1) * g.K + k) * NR * NL;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int p = 0; p < NL; ++p) {
            const cd y1 = h1[r * NL + p];
            sl[r][p] = cd((y1.re - y0[r][p].re) / (x1 - x0), (y1.im - y0[r][p].im) / (x1 - x0));
          }
      }
    }
    for (int l = 0; l < g.L; ++l) {
      cd H[NR][NL], y[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        y[r] = rx[((size_t)b * NR + r) * lk + (size_t)l * g.K + k];
#pragma unroll
        for (int p = 0; p < NL; ++p)
          H[r][p] = n_g == 2 ? cd(sl[r][p].re * ((double)l - x0) + y0[r][p].re, sl[r][p].im * ((double)l - x0) + y0[r][p].im)
                             : y0[r][p];
      }
      cd xh[NL];
      double sc[NL];
      nrx::mmse_solve<NR, NL>(H, y, nv, xh, sc);
#pragma unroll
      for (int p = 0; p < NL; ++p) {
        eq[((size_t)b * NL + p) * lk + (size_t)l * g.K + k] = xh[p];
        scale[((size_t)b * NL + p) * lk + (size_t)l * g.K + k] = sc[p];
      }
    }
  }
}
}  // namespace

extern "C" int32_t nrx_chest_ls_mmse_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                         const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                         int32_t L, int32_t K, int32_t nr, int32_t P, const double* noise_var,
                                         int32_t nv_stride, void* hk_ws, void* eq, double* scale, int32_t n_batch,
                                         void* stream) {
  NRX_REQUIRE(rx && pilots && port_ks && dmrs_syms && noise_var && hk_ws && eq && scale, NRX_E_ARG,
              "nrx_chest_ls_mmse: NULL buffer");
  NRX_REQUIRE(n_ds >= 1 && n_ds <= 8 && l_cdm >= 1 && k_cdm >= 1 && n_k >= 1, NRX_E_ARG, "nrx_chest_ls_mmse: bad DMRS geometry");
  NRX_REQUIRE(n_k % k_cdm == 0 && n_ds % l_cdm == 0, NRX_E_UNSUPPORTED, "nrx_chest_ls_mmse: Partial CDMs are not supported in this version.");
  NRX_REQUIRE(n_ds / l_cdm <= 2, NRX_E_UNSUPPORTED, "nrx_chest_ls_mmse: more than 2 DMRS time groups (use nrx_chest_ls + nrx_mmse_equalize)");
  NRX_REQUIRE(L >= 1 && K >= 1 && nr >= 1 && P >= 1 && n_batch >= 0, NRX_E_ARG, "nrx_chest_ls_mmse: bad sizes");
  if (n_batch == 0) return NRX_OK;
  ChestGeom g;
  g.n_ds = n_ds;
  for (int i = 0; i < n_ds; ++i) {
    NRX_REQUIRE(dmrs_syms[i] >= 0 && dmrs_syms[i] < L, NRX_E_ARG, "nrx_chest_ls_mmse: DMRS symbol index out of range");
    g.ds[i] = dmrs_syms[i];
  }
  g.l_cdm = l_cdm; g.k_cdm = k_cdm; g.n_k = n_k; g.L = L; g.K = K; g.nr = nr; g.P = P;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid2(nrx::stream_grid((long)n_batch * K, 128));
#define NRX_CM_CASE(NR, NL)                                                                                              \
  if (nr == NR && P == NL) {                                                                                             \
    hipLaunchKernelGGL(chest_ls_kernel<double>, dim3(nrx::stream_grid((long)n_batch * K * nr * P, 256)), dim3(256), 0, st, \
                       (const cd*)rx, (const cd*)pilots, pil_set, port_ks, g, (cd*)hk_ws, n_batch, 1);                   \
    hipLaunchKernelGGL((mmse_interp_kernel<NR, NL>), grid2, dim3(128), 0, st, (const cd*)rx, (const cd*)hk_ws, g,        \
                       noise_var, nv_stride, (cd*)eq, scale, n_batch);                                                   \
    NRX_CHECK_LAUNCH("nrx_chest_ls_mmse");                                                                               \
    return NRX_OK;                                                                                                       \
  }
  NRX_CM_CASE(1, 1) NRX_CM_CASE(2, 1) NRX_CM_CASE(2, 2) NRX_CM_CASE(4, 1) NRX_CM_CASE(4, 2) NRX_CM_CASE(4, 3) NRX_CM_CASE(4, 4)
  NRX_CM_CASE(8, 1) NRX_CM_CASE(8, 2) NRX_CM_CASE(8, 4)
#undef NRX_CM_CASE
  NRX_REQUIRE(false, NRX_E_UNSUPPORTED, "nrx_chest_ls_mmse: (Nr=%d, layers=%d) not built", nr, P);
}


extern "C" int32_t nrx_chest_ls_f32(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L, int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream) { return chest_entry<float>(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, hest, n_batch, stream); }
extern "C" int32_t nrx_chest_ls_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L, int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream) { return chest_entry<double>(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, hest, n_batch, stream); }
