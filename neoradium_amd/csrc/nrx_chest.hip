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

// POLAR: the CDM-group estimates come as (unwrapped angle, magnitude) pairs from chest_polar_prep_kernel +
// chest_unwrap_kernel (`pol`, [row = ((b*n_g + tg)*nr + r)*P + p][j][2]); angle and magnitude are inter/extrapolated
// separately along the subcarriers (utils.py:38-42 polarInterpolate), the symbol axis stays complex-linear.
// MEANS: the CDM-group means come precomputed from chest_polar_prep_kernel<false> (`pol`, same row layout, (re, im) pairs)
// instead of being re-derived by every lane -- each mean is used by the ~2*k_cdm*spacing subcarriers around it and costs
// l_cdm*k_cdm complex divisions; same operations in the same order, so the estimate is bit-identical.
template <typename T, bool POLAR, bool MEANS = false>
__global__ void __launch_bounds__(256)
chest_ls_kernel(const cx<T>* __restrict__ rx, const cx<T>* __restrict__ pilots, const int32_t* __restrict__ pil_set,
                const int32_t* __restrict__ port_ks, ChestGeom g, cx<T>* __restrict__ hest, int n_batch, int hk_only,
                cx<T>* __restrict__ hk_out, const double* __restrict__ pol) {
  const int rp = g.nr * g.P;
  const int64_t per = (int64_t)g.K * rp;
  const int64_t total = (int64_t)n_batch * per;
  const int n_j = g.n_k / g.k_cdm;      // CDM groups along frequency
  const int n_g = g.n_ds / g.l_cdm;     // estimates along time
  const int cdm = g.l_cdm * g.k_cdm;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    int b, k, r, p;
    int64_t e;
    if (total < (1ll << 31)) {           // 32-bit index arithmetic where it fits (a 64-bit division is ~100 instructions)
      const uint32_t g32 = (uint32_t)gi, per32 = (uint32_t)per;
      b = (int)(g32 / per32);
      const uint32_t e32 = g32 - (uint32_t)b * per32;
      k = (int)(e32 / (uint32_t)rp);
      const uint32_t q32 = e32 - (uint32_t)k * (uint32_t)rp;
      r = (int)(q32 / (uint32_t)g.P);
      p = (int)(q32 - (uint32_t)r * (uint32_t)g.P);
      e = e32;
    } else {
      b = (int)(gi / per);
      e = gi - (int64_t)b * per;
      k = (int)(e / rp);
      r = (int)(e - (int64_t)k * rp) / g.P;
      p = (int)(e - (int64_t)k * rp) % g.P;
    }
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
      // lo = first index with centre >= k (the centres ascend).  Found from a proportional guess and a walk to the exact
      // place instead of a bisection: ten dependent table reads per lane became two or three for regular pilot combs.
      const double c0 = centre(0), cN = centre(n_j - 1);
      int lo = cN > c0 ? (int)(((double)k - c0) / (cN - c0) * (double)(n_j - 1)) : 0;
      lo = lo < 0 ? 0 : (lo > n_j - 1 ? n_j - 1 : lo);
      while (lo < n_j && centre(lo) < (double)k) ++lo;
      while (lo > 0 && centre(lo - 1) >= (double)k) --lo;
      j = lo < 1 ? 1 : (lo > n_j - 1 ? n_j - 1 : lo);
    }
    cd hk[4];  // estimate at subcarrier k for each time group (n_g <= 4)
    for (int tg = 0; tg < n_g; ++tg) {
      auto group_mean = [&](int jj) {  // LS estimates rx/pilot averaged over the CDM group (grid.py:775-793)
        if constexpr (MEANS) {
          const double* q = pol + (((((size_t)b * n_g + tg) * g.nr + r) * g.P + p) * (size_t)n_j + jj) * 2;
          return cd(q[0], q[1]);
        }
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
      const double x0 = centre(j - 1), x1 = centre(j);
      if constexpr (POLAR) {
        const double* q = pol + ((((size_t)b * n_g + tg) * g.nr + r) * g.P + p) * (size_t)n_j * 2;
        const double t0 = q[2 * (j - 1)], a0 = q[2 * (j - 1) + 1], t1 = q[2 * j], a1 = q[2 * j + 1];
        const double tn = (t1 - t0) / (x1 - x0) * ((double)k - x0) + t0;
        const double an = (a1 - a0) / (x1 - x0) * ((double)k - x0) + a0;
        double sn, cs;
        sincos(tn, &sn, &cs);
        hk[tg] = cd(an * cs, an * sn);
      } else {
        const cd y0 = group_mean(j - 1), y1 = group_mean(j);
        const cd sl((y1.re - y0.re) / (x1 - x0), (y1.im - y0.im) / (x1 - x0));
        hk[tg] = cd(sl.re * ((double)k - x0) + y0.re, sl.im * ((double)k - x0) + y0.im);
      }
    }
    if (hk_out)      // estimates at the DMRS time groups, (n, n_g, K, nr, P): input of the noise estimate
      for (int tg = 0; tg < n_g; ++tg) hk_out[((size_t)b * n_g + tg) * per + e] = cx<T>(hk[tg]);
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


// The estimate at the DMRS time groups for the fused estimate + equalise path (chest_ls_kernel<double, false, true> with
// hk_only): one lane per (item, subcarrier, port) walks the receive antennas, so the segment search and the two centres -- which
// depend on (subcarrier, port) only -- are done once instead of once per antenna (the kernel is bound by its instruction count:
// ~700 per lane in the general form).  Same expressions in the same order: the same bits.
__global__ void __launch_bounds__(256)
chest_ls_hk_kernel(const int32_t* __restrict__ port_ks, ChestGeom g, cd* __restrict__ hk, int n_batch, const double* __restrict__ pol) {
  const int n_j = g.n_k / g.k_cdm, n_g = g.n_ds / g.l_cdm;
  const uint32_t per_b = (uint32_t)g.K * (uint32_t)g.P;
  const uint32_t total = (uint32_t)n_batch * per_b;
  const int64_t per = (int64_t)g.K * g.nr * g.P;
  for (uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += gridDim.x * blockDim.x) {
    const int b = (int)(gi / per_b);
    const uint32_t e0 = gi - (uint32_t)b * per_b;
    const int k = (int)(e0 / (uint32_t)g.P), p = (int)(e0 - (uint32_t)k * (uint32_t)g.P);
    const int32_t* ks = port_ks + (size_t)p * g.n_k;
    auto centre = [&](int j) {  // mean subcarrier index of CDM group j (grid.py:795-796)
      double s = 0;
      for (int q = 0; q < g.k_cdm; ++q) s += (double)ks[j * g.k_cdm + q];
      return s / (double)g.k_cdm;
    };
    int j = 1;
    if (n_j > 1) {
      const double c0 = centre(0), cN = centre(n_j - 1);
      int lo = cN > c0 ? (int)(((double)k - c0) / (cN - c0) * (double)(n_j - 1)) : 0;
      lo = lo < 0 ? 0 : (lo > n_j - 1 ? n_j - 1 : lo);
      while (lo < n_j && centre(lo) < (double)k) ++lo;
      while (lo > 0 && centre(lo - 1) >= (double)k) --lo;
      j = lo < 1 ? 1 : (lo > n_j - 1 ? n_j - 1 : lo);
    }
    const double x0 = n_j > 1 ? centre(j - 1) : 0.0, x1 = n_j > 1 ? centre(j) : 1.0;
    for (int tg = 0; tg < n_g; ++tg)
      for (int r = 0; r < g.nr; ++r) {
        const double* q = pol + ((((size_t)b * n_g + tg) * g.nr + r) * g.P + p) * (size_t)n_j * 2;
        cd v;
        if (n_j == 1) {
          v = cd(q[0], q[1]);
        } else {
          const cd y0(q[2 * (j - 1)], q[2 * (j - 1) + 1]), y1(q[2 * j], q[2 * j + 1]);
          const cd sl((y1.re - y0.re) / (x1 - x0), (y1.im - y0.im) / (x1 - x0));
          v = cd(sl.re * ((double)k - x0) + y0.re, sl.im * ((double)k - x0) + y0.im);
        }
        hk[((size_t)b * n_g + tg) * per + ((size_t)k * g.nr + r) * g.P + p] = v;
      }
  }
}

// ---- polar interpolation, step 1 (utils.py:39): angle and magnitude of every CDM-group estimate.
// pol[row][j] = (atan2(im, re), hypot(re, im)), row = ((b*n_g + tg)*nr + r)*P + p.  POLAR = false leaves (re, im) in the
// same layout: the CDM-group means themselves, input of the tap-table interpolators (nrx_interp_taps_f64).
template <bool POLAR>
__global__ void __launch_bounds__(256)
chest_polar_prep_kernel(const cd* __restrict__ rx, const cd* __restrict__ pilots, const int32_t* __restrict__ pil_set,
                        const int32_t* __restrict__ port_ks, ChestGeom g, double* __restrict__ pol, int n_batch) {
  const int n_j = g.n_k / g.k_cdm, n_g = g.n_ds / g.l_cdm, cdm = g.l_cdm * g.k_cdm;
  const int64_t total = (int64_t)n_batch * n_g * g.nr * g.P * n_j;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    int j, p, r, tg, b;
    if (total < (1ll << 31)) {           // 32-bit index arithmetic where it fits (a 64-bit division is ~100 instructions)
      const uint32_t g32 = (uint32_t)gi;
      uint32_t row = g32 / (uint32_t)n_j;
      j = (int)(g32 - row * (uint32_t)n_j);
      uint32_t q = row / (uint32_t)g.P;
      p = (int)(row - q * (uint32_t)g.P);
      row = q;
      q = row / (uint32_t)g.nr;
      r = (int)(row - q * (uint32_t)g.nr);
      b = (int)(q / (uint32_t)n_g);
      tg = (int)(q - (uint32_t)b * (uint32_t)n_g);
    } else {
      j = (int)(gi % n_j);
      const int64_t row = gi / n_j;
      p = (int)(row % g.P), r = (int)((row / g.P) % g.nr), tg = (int)((row / ((int64_t)g.P * g.nr)) % n_g);
      b = (int)(row / ((int64_t)g.P * g.nr * n_g));
    }
    const int32_t* ks = port_ks + (size_t)p * g.n_k;
    const cd* pil = pilots + ((size_t)(pil_set ? pil_set[b] : 0) * g.P + p) * g.n_ds * g.n_k;
    const cd* rxb = rx + ((size_t)b * g.nr + r) * g.L * g.K;
    cd s(0, 0);
    for (int ll = 0; ll < g.l_cdm; ++ll) {
      const int di = tg * g.l_cdm + ll;
      for (int q = 0; q < g.k_cdm; ++q)
        s = s + nrx::cdiv(rxb[(size_t)g.ds[di] * g.K + ks[j * g.k_cdm + q]], pil[(size_t)di * g.n_k + j * g.k_cdm + q]);
    }
    s = cd(s.re / (double)cdm, s.im / (double)cdm);
    pol[2 * gi] = POLAR ? atan2(s.im, s.re) : s.re;
    pol[2 * gi + 1] = POLAR ? hypot(s.re, s.im) : s.im;
  }
}

// ---- step 2: np.unwrap along the subcarrier axis, one thread per row, in NumPy's operation order (the phase
// corrections are accumulated sequentially like np.cumsum, so the result is the same double).
__global__ void __launch_bounds__(64) chest_unwrap_kernel(double* __restrict__ pol, int n_j, int64_t n_rows) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  double* q = pol + (size_t)row * n_j * 2;
  const double pi = 3.141592653589793, two_pi = 6.283185307179586;
  double prev = q[0], corr = 0.0;
  for (int j = 1; j < n_j; ++j) {
    const double cur = q[2 * j];
    const double dd = cur - prev;
    double m = fmod(dd + pi, two_pi);          // np.mod: result takes the sign of the divisor
    if (m != 0.0) { if (m < 0.0) m += two_pi; } else m = 0.0;
    double ddmod = m - pi;
    if (ddmod == -pi && dd > 0.0) ddmod = pi;
    double ph = ddmod - dd;
    if (fabs(dd) < pi) ph = 0.0;
    corr += ph;
    q[2 * j] = cur + corr;
    prev = cur;
  }
}

// ---- noise side output of estimateChannelLsEx (grid.py:808-837).
// (a) windowed channel impulse response: only the 2*rise taps under the raised-cosine window are non-zero,
//     cirw[row][t] = win[t] * (1/K) sum_k hk[row][k] e^{+2 pi i k m_t / K},  m_t = t (t < rise) or K - 2*rise + t.
//     tw[q] = e^{2 pi i q / K} (caller table); the index k*m mod K is advanced incrementally.
__global__ void __launch_bounds__(128)
chest_cir_kernel(const cd* __restrict__ hk, const cd* __restrict__ tw, const double* __restrict__ win, int rise, int K,
                 int rp, cd* __restrict__ cirw) {
  const int64_t row = blockIdx.x;                    // (b*n_g + tg)*rp + (r*P + p)
  const int t = blockIdx.y * blockDim.x + threadIdx.x;
  if (t >= 2 * rise) return;
  const int m = t < rise ? t : K - 2 * rise + t;
  const cd* h = hk + (size_t)(row / rp) * K * rp + (row % rp);
  cd acc(0, 0);
  int idx = 0;
  for (int k = 0; k < K; ++k) {
    nrx::cmac(acc, h[(size_t)k * rp], tw[idx]);
    idx += m;
    if (idx >= K) idx -= K;
  }
  const double w = win[t];
  cirw[(size_t)row * 2 * rise + t] = cd(acc.re / (double)K * w, acc.im / (double)K * w);
}

// (b) residuals at the pilots: delta = (rx/pilot at the port's own pilot q) - denoised estimate sampled at subcarrier
//     ks_last[q] -- the pilot subcarriers of the LAST port for every port (QUIRK grid.py:823, kept).
//     deltas[b][((p*n_ds + di)*n_k + q)*nr + r]
__global__ void __launch_bounds__(256)
chest_delta_kernel(const cd* __restrict__ rx, const cd* __restrict__ pilots, const int32_t* __restrict__ pil_set,
                   const int32_t* __restrict__ port_ks, const int32_t* __restrict__ ks_sample, ChestGeom g,
                   const cd* __restrict__ cirw, const cd* __restrict__ tw, int rise, cd* __restrict__ deltas, int n_batch) {
  const int n_g = g.n_ds / g.l_cdm;
  const int64_t per = (int64_t)g.P * g.n_ds * g.n_k * g.nr;
  const int64_t total = (int64_t)n_batch * per;
  const int32_t* ks_last = ks_sample ? ks_sample : port_ks + (size_t)(g.P - 1) * g.n_k;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(gi / per);
    int64_t e = gi - (int64_t)b * per;
    const int r = (int)(e % g.nr); e /= g.nr;
    const int q = (int)(e % g.n_k); e /= g.n_k;
    const int di = (int)(e % g.n_ds);
    const int p = (int)(e / g.n_ds);
    const int tg = di / g.l_cdm;
    const cd* pil = pilots + ((size_t)(pil_set ? pil_set[b] : 0) * g.P + p) * g.n_ds * g.n_k;
    const cd raw = nrx::cdiv(rx[(((size_t)b * g.nr + r) * g.L + g.ds[di]) * g.K + port_ks[(size_t)p * g.n_k + q]],
                             pil[(size_t)di * g.n_k + q]);
    const cd* c = cirw + ((((size_t)b * n_g + tg) * g.nr + r) * g.P + p) * 2 * rise;
    const int kq = ks_last[q];
    cd den(0, 0);
    for (int half = 0; half < 2; ++half) {
      const int m0 = half ? g.K - rise : 0;
      int idx = (int)(((int64_t)kq * m0) % g.K);
      for (int t = 0; t < rise; ++t) {
        const cd w = tw[idx];
        nrx::cmac(den, c[half * rise + t], cd(w.re, -w.im));   // e^{-2 pi i kq m / K}
        idx += kq;
        if (idx >= g.K) idx -= g.K;
      }
    }
    deltas[gi] = cd(raw.re - den.re, raw.im - den.im);
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
  hipLaunchKernelGGL((chest_ls_kernel<T, false>), dim3(nrx::stream_grid((long)n_batch * K * nr * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cx<T>*)rx, (const cx<T>*)pilots, pil_set, port_ks, g, (cx<T>*)hest, n_batch, 0,
                     (cx<T>*)nullptr, (const double*)nullptr);
  NRX_CHECK_LAUNCH("nrx_chest_ls");
  return NRX_OK;
}

// Estimate + equalise without materialising the (L, K, Nr, P) estimate: chest_ls_kernel leaves the frequency-
// interpolated estimates of the (<= 2) DMRS time groups in `hk`; one thread per (item, subcarrier) then walks the L
// symbols, forms H_l with the same inter/extrapolation expressions as chest_ls_kernel (grid.py:853-866) and runs
// the same MMSE solve as mmse_kernel -- results are identical to nrx_chest_ls + nrx_mmse_equalize.
template <int NR, int NL>
__global__ void __launch_bounds__(128, 2)   // two waves per SIMD (256 registers; the 4x4 solve wants 284: 29 words of scratch) beat one: 0.72 -> 0.67 ms
mmse_interp_kernel(const cd* __restrict__ rx, const cd* __restrict__ hk, ChestGeom g, const double* __restrict__ noise_var,
                   int nv_stride, cd* __restrict__ eq, double* __restrict__ scale, int n_batch, uint32_t sym_mask) {
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
      if (!((sym_mask >> (l & 31)) & 1u)) continue;       // a symbol the caller does not want equalised (it holds no data RE)
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
// ---- interpolation by tap tables: out[q] = sum_t w[q][t] * in[idx[q][t]] for every row, both components of the pair
// separately; polar pairs (angle, magnitude) are recombined as magnitude * e^{i angle} (utils.py:42).  The tables are the
// linear operator of the reference's interpolators (interp1d nearest / quadratic, RBFInterpolator with k nearest
// neighbours in one or two dimensions, utils.py:26-35, grid.py:853-861) for one set of sample positions; they depend
// on the pilot geometry only, so the caller builds them once.  One lane per (outer, q, inner) with `inner` fastest:
// rows are numbered outer*inner + in, table = in % n_tabs (the port is the fastest index of every layout used).
struct TapGeom {
  int64_t n_outer;
  int32_t inner, n_out, n_taps, n_tabs, polar;
  int64_t tab_stride, in_outer, in_inner, in_j, out_outer, out_inner, out_q;
};

__global__ void __launch_bounds__(256)
interp_taps_kernel(const cd* __restrict__ in, const int32_t* __restrict__ idx, const double* __restrict__ w, TapGeom t,
                   cd* __restrict__ out) {
  const int64_t per = (int64_t)t.n_out * t.inner;
  const int64_t total = t.n_outer * per;
  for (int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = gi / per;
    const int64_t e = gi - o * per;
    const int q = (int)(e / t.inner), ii = (int)(e - (int64_t)q * t.inner);
    const size_t tab = (size_t)(ii % t.n_tabs) * t.tab_stride + (size_t)q * t.n_taps;
    const cd* src = in + o * t.in_outer + (int64_t)ii * t.in_inner;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < t.n_taps; ++k) {
      const double wk = w[tab + k];
      const cd v = src[(int64_t)idx[tab + k] * t.in_j];
      a += wk * v.re;
      b += wk * v.im;
    }
    cd r(a, b);
    if (t.polar) {
      double sn, cs;
      sincos(a, &sn, &cs);
      r = cd(b * cs, b * sn);
    }
    out[o * t.out_outer + (int64_t)ii * t.out_inner + (int64_t)q * t.out_q] = r;
  }
}

// ---- timing estimate (grid.py:592-622): xc[d] = sum over (rx antenna, port) of | sum_n rx[r][n+d] conj(ref[p][n]) |,
// d = 0..N-1, with the reference waveform zero outside [n0, n0+M) (the CSI-RS symbols).  One workgroup per lag.
__global__ void __launch_bounds__(256)
xcorr_abs_kernel(const cd* __restrict__ rx, const cd* __restrict__ ref, int N, int n_ref, int n0, int M, int nr, int P,
                 double* __restrict__ xc) {
  __shared__ double red[2][4];
  const int d = blockIdx.x;
  double total = 0.0;
  for (int r = 0; r < nr; ++r)
    for (int p = 0; p < P; ++p) {
      const cd* a = rx + (size_t)r * N + d;
      const cd* b = ref + (size_t)p * n_ref;
      cd acc(0, 0);
      for (int n = n0 + threadIdx.x; n < n0 + M && n + d < N; n += blockDim.x) {
        const cd v = b[n];
        nrx::cmac(acc, a[n], cd(v.re, -v.im));
      }
      for (int off = 32; off > 0; off >>= 1) {
        acc.re += __shfl_down(acc.re, off);
        acc.im += __shfl_down(acc.im, off);
      }
      if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc.re; red[1][threadIdx.x >> 6] = acc.im; }
      __syncthreads();
      if (threadIdx.x == 0) {
        const double re = red[0][0] + red[0][1] + red[0][2] + red[0][3], im = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        total += hypot(re, im);
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) xc[d] = total;
}
}  // namespace

static int32_t chest_ls_mmse_impl(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                  const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                  int32_t L, int32_t K, int32_t nr, int32_t P, const double* noise_var,
                                  int32_t nv_stride, void* hk_ws, void* eq, double* scale, int32_t n_batch,
                                  uint32_t sym_mask, void* stream) {
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
  // the CDM-group means live behind the per-time-group estimates in the workspace (see the header for its size)
  const int n_j = n_k / k_cdm;
  const int64_t mrows = (int64_t)n_batch * (n_ds / l_cdm) * nr * P;
  double* means = (double*)((cd*)hk_ws + (size_t)n_batch * (n_ds / l_cdm) * K * nr * P);
#define NRX_CM_CASE(NR, NL)                                                                                              \
  if (nr == NR && P == NL) {                                                                                             \
    hipLaunchKernelGGL(chest_polar_prep_kernel<false>, dim3(nrx::stream_grid(mrows * n_j, 256)), dim3(256), 0, st,        \
                       (const cd*)rx, (const cd*)pilots, pil_set, port_ks, g, means, n_batch);                           \
    if ((int64_t)n_batch * K * P < (1ll << 31))                                                                          \
      hipLaunchKernelGGL(chest_ls_hk_kernel, dim3(nrx::stream_grid((long)n_batch * K * P, 256)), dim3(256), 0, st, port_ks, g, \
                         (cd*)hk_ws, n_batch, (const double*)means);                                                    \
    else                                                                                                                 \
      hipLaunchKernelGGL((chest_ls_kernel<double, false, true>), dim3(nrx::stream_grid((long)n_batch * K * nr * P, 256)), \
                         dim3(256), 0, st, (const cd*)rx, (const cd*)pilots, pil_set, port_ks, g, (cd*)hk_ws, n_batch, 1, \
                         (cd*)nullptr, (const double*)means);                                                           \
    hipLaunchKernelGGL((mmse_interp_kernel<NR, NL>), grid2, dim3(128), 0, st, (const cd*)rx, (const cd*)hk_ws, g,        \
                       noise_var, nv_stride, (cd*)eq, scale, n_batch, sym_mask);                                         \
    NRX_CHECK_LAUNCH("nrx_chest_ls_mmse");                                                                               \
    return NRX_OK;                                                                                                       \
  }
  NRX_CM_CASE(1, 1) NRX_CM_CASE(2, 1) NRX_CM_CASE(2, 2) NRX_CM_CASE(4, 1) NRX_CM_CASE(4, 2) NRX_CM_CASE(4, 3) NRX_CM_CASE(4, 4)
  NRX_CM_CASE(8, 1) NRX_CM_CASE(8, 2) NRX_CM_CASE(8, 4)
#undef NRX_CM_CASE
  NRX_REQUIRE(false, NRX_E_UNSUPPORTED, "nrx_chest_ls_mmse: (Nr=%d, layers=%d) not built", nr, P);
}

extern "C" int32_t nrx_chest_ls_mmse_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                         const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                         int32_t L, int32_t K, int32_t nr, int32_t P, const double* noise_var,
                                         int32_t nv_stride, void* hk_ws, void* eq, double* scale, int32_t n_batch,
                                         void* stream) {
  return chest_ls_mmse_impl(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, noise_var, nv_stride,
                            hk_ws, eq, scale, n_batch, 0xffffffffu, stream);
}

// ... for the OFDM symbols of `sym_mask` only (bit l = symbol l, L <= 32): the others -- symbols without data REs, whose
// equalised values nobody reads -- are left untouched in eq / scale.
extern "C" int32_t nrx_chest_ls_mmse_syms_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                              const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                              int32_t L, int32_t K, int32_t nr, int32_t P, const double* noise_var,
                                              int32_t nv_stride, void* hk_ws, void* eq, double* scale, int32_t n_batch,
                                              uint32_t sym_mask, void* stream) {
  NRX_REQUIRE(L <= 32, NRX_E_ARG, "nrx_chest_ls_mmse_syms: at most 32 symbols");
  return chest_ls_mmse_impl(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, noise_var, nv_stride,
                            hk_ws, eq, scale, n_batch, sym_mask, stream);
}


extern "C" int32_t nrx_chest_ls_f32(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L, int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream) { return chest_entry<float>(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, hest, n_batch, stream); }
extern "C" int32_t nrx_chest_ls_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L, int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream) { return chest_entry<double>(rx, pilots, pil_set, port_ks, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, hest, n_batch, stream); }


static int32_t chest_fill_geom(ChestGeom& g, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm,
                               int32_t n_k, int32_t L, int32_t K, int32_t nr, int32_t P, const char* who) {
  NRX_REQUIRE(dmrs_syms, NRX_E_ARG, "%s: NULL dmrs_syms", who);
  NRX_REQUIRE(n_ds >= 1 && n_ds <= 8 && l_cdm >= 1 && k_cdm >= 1 && n_k >= 1, NRX_E_ARG, "%s: bad DMRS geometry", who);
  NRX_REQUIRE(n_k % k_cdm == 0 && n_ds % l_cdm == 0, NRX_E_UNSUPPORTED, "%s: Partial CDMs are not supported in this version.", who);
  NRX_REQUIRE(n_ds / l_cdm <= 4, NRX_E_UNSUPPORTED, "%s: more than 4 DMRS time groups", who);
  NRX_REQUIRE(L >= 1 && K >= 1 && nr >= 1 && P >= 1, NRX_E_ARG, "%s: bad sizes", who);
  g.n_ds = n_ds;
  for (int i = 0; i < n_ds; ++i) {
    NRX_REQUIRE(dmrs_syms[i] >= 0 && dmrs_syms[i] < L, NRX_E_ARG, "%s: DMRS symbol index out of range", who);
    g.ds[i] = dmrs_syms[i];
  }
  g.l_cdm = l_cdm; g.k_cdm = k_cdm; g.n_k = n_k; g.L = L; g.K = K; g.nr = nr; g.P = P;
  return NRX_OK;
}

extern "C" int32_t nrx_chest_ls_ex_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                       const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                       int32_t L, int32_t K, int32_t nr, int32_t P, int32_t polar, void* pol_ws,
                                       void* hk_out, void* hest, int32_t n_batch, void* stream) {
  NRX_REQUIRE(rx && pilots && port_ks && hest, NRX_E_ARG, "nrx_chest_ls_ex: NULL buffer");
  NRX_REQUIRE(!polar || pol_ws, NRX_E_ARG, "nrx_chest_ls_ex: polar interpolation needs pol_ws");
  NRX_REQUIRE(n_batch >= 0, NRX_E_ARG, "nrx_chest_ls_ex: negative batch");
  ChestGeom g;
  const int32_t rc = chest_fill_geom(g, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, "nrx_chest_ls_ex");
  if (rc != NRX_OK) return rc;
  NRX_REQUIRE(!polar || n_k / k_cdm >= 2, NRX_E_ARG, "nrx_chest_ls_ex: polar interpolation needs two or more CDM groups");
  if (n_batch == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(nrx::stream_grid((long)n_batch * K * nr * P, 256));
  if (polar) {
    const int n_j = n_k / k_cdm;
    const int64_t rows = (int64_t)n_batch * (n_ds / l_cdm) * nr * P;
    hipLaunchKernelGGL(chest_polar_prep_kernel<true>, dim3(nrx::stream_grid(rows * n_j, 256)), dim3(256), 0, st, (const cd*)rx,
                       (const cd*)pilots, pil_set, port_ks, g, (double*)pol_ws, n_batch);
    hipLaunchKernelGGL(chest_unwrap_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, (double*)pol_ws, n_j, rows);
    hipLaunchKernelGGL((chest_ls_kernel<double, true>), grid, dim3(256), 0, st, (const cd*)rx, (const cd*)pilots, pil_set,
                       port_ks, g, (cd*)hest, n_batch, 0, (cd*)hk_out, (const double*)pol_ws);
  } else {
    hipLaunchKernelGGL((chest_ls_kernel<double, false>), grid, dim3(256), 0, st, (const cd*)rx, (const cd*)pilots, pil_set,
                       port_ks, g, (cd*)hest, n_batch, 0, (cd*)hk_out, (const double*)nullptr);
  }
  NRX_CHECK_LAUNCH("nrx_chest_ls_ex");
  return NRX_OK;
}

extern "C" int32_t nrx_chest_noise_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                       const int32_t* ks_sample, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                       int32_t L, int32_t K, int32_t nr, int32_t P, const void* hk, const void* tw,
                                       const double* win, int32_t rise, void* cir_ws, void* deltas, int32_t n_batch,
                                       void* stream) {
  NRX_REQUIRE(rx && pilots && port_ks && hk && tw && win && cir_ws && deltas, NRX_E_ARG, "nrx_chest_noise: NULL buffer");
  NRX_REQUIRE(rise >= 1 && 2 * rise <= K, NRX_E_ARG, "nrx_chest_noise: window of 2*%d taps does not fit %d subcarriers", rise, K);
  NRX_REQUIRE(n_batch >= 0, NRX_E_ARG, "nrx_chest_noise: negative batch");
  ChestGeom g;
  const int32_t rc = chest_fill_geom(g, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, "nrx_chest_noise");
  if (rc != NRX_OK) return rc;
  if (n_batch == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)n_batch * (n_ds / l_cdm) * nr * P;
  NRX_REQUIRE(rows < 0x7fffffff, NRX_E_UNSUPPORTED, "nrx_chest_noise: batch too large");
  hipLaunchKernelGGL(chest_cir_kernel, dim3((unsigned)rows, (unsigned)((2 * rise + 127) / 128)), dim3(128), 0, st,
                     (const cd*)hk, (const cd*)tw, win, rise, K, nr * P, (cd*)cir_ws);
  hipLaunchKernelGGL(chest_delta_kernel, dim3(nrx::stream_grid((long)n_batch * P * n_ds * n_k * nr, 256)), dim3(256), 0, st,
                     (const cd*)rx, (const cd*)pilots, pil_set, port_ks, ks_sample, g, (const cd*)cir_ws, (const cd*)tw, rise,
                     (cd*)deltas, n_batch);
  NRX_CHECK_LAUNCH("nrx_chest_noise");
  return NRX_OK;
}


extern "C" int32_t nrx_chest_pilot_means_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                             const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                             int32_t L, int32_t K, int32_t nr, int32_t P, int32_t polar, void* out,
                                             int32_t n_batch, void* stream) {
  NRX_REQUIRE(rx && pilots && port_ks && out, NRX_E_ARG, "nrx_chest_pilot_means: NULL buffer");
  NRX_REQUIRE(n_batch >= 0, NRX_E_ARG, "nrx_chest_pilot_means: negative batch");
  ChestGeom g;
  const int32_t rc = chest_fill_geom(g, dmrs_syms, n_ds, l_cdm, k_cdm, n_k, L, K, nr, P, "nrx_chest_pilot_means");
  if (rc != NRX_OK) return rc;
  if (n_batch == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int n_j = n_k / k_cdm;
  const int64_t rows = (int64_t)n_batch * (n_ds / l_cdm) * nr * P;
  if (polar) {
    hipLaunchKernelGGL(chest_polar_prep_kernel<true>, dim3(nrx::stream_grid(rows * n_j, 256)), dim3(256), 0, st, (const cd*)rx,
                       (const cd*)pilots, pil_set, port_ks, g, (double*)out, n_batch);
    hipLaunchKernelGGL(chest_unwrap_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, st, (double*)out, n_j, rows);
  } else {
    hipLaunchKernelGGL(chest_polar_prep_kernel<false>, dim3(nrx::stream_grid(rows * n_j, 256)), dim3(256), 0, st, (const cd*)rx,
                       (const cd*)pilots, pil_set, port_ks, g, (double*)out, n_batch);
  }
  NRX_CHECK_LAUNCH("nrx_chest_pilot_means");
  return NRX_OK;
}

extern "C" int32_t nrx_interp_taps_f64(const void* in, const int32_t* idx, const double* w, int32_t n_taps, int32_t n_tabs,
                                       int64_t tab_stride, int64_t n_outer, int32_t inner, int32_t n_out, int64_t in_outer,
                                       int64_t in_inner, int64_t in_j, int64_t out_outer, int64_t out_inner, int64_t out_q,
                                       int32_t polar, void* out, void* stream) {
  NRX_REQUIRE(in && idx && w && out, NRX_E_ARG, "nrx_interp_taps: NULL buffer");
  NRX_REQUIRE(n_taps >= 1 && n_tabs >= 1 && inner >= 1 && n_out >= 1 && n_outer >= 0 && tab_stride >= 0, NRX_E_ARG,
              "nrx_interp_taps: bad sizes");
  NRX_REQUIRE(inner % n_tabs == 0, NRX_E_ARG, "nrx_interp_taps: %d tables do not divide the inner extent %d", n_tabs, inner);
  if (n_outer == 0) return NRX_OK;
  TapGeom t;
  t.n_outer = n_outer; t.inner = inner; t.n_out = n_out; t.n_taps = n_taps; t.n_tabs = n_tabs; t.polar = polar;
  t.tab_stride = tab_stride; t.in_outer = in_outer; t.in_inner = in_inner; t.in_j = in_j;
  t.out_outer = out_outer; t.out_inner = out_inner; t.out_q = out_q;
  hipLaunchKernelGGL(interp_taps_kernel, dim3(nrx::stream_grid((long)(n_outer * n_out * inner), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const cd*)in, idx, w, t, (cd*)out);
  NRX_CHECK_LAUNCH("nrx_interp_taps");
  return NRX_OK;
}

extern "C" int32_t nrx_xcorr_abs_f64(const void* rx, const void* ref, int32_t n_samples, int32_t n_ref, int32_t ref_start,
                                     int32_t ref_len, int32_t nr, int32_t P, double* xc, void* stream) {
  NRX_REQUIRE(rx && ref && xc, NRX_E_ARG, "nrx_xcorr_abs: NULL buffer");
  NRX_REQUIRE(n_samples >= 1 && n_ref >= 1 && nr >= 1 && P >= 1, NRX_E_ARG, "nrx_xcorr_abs: bad sizes");
  NRX_REQUIRE(ref_start >= 0 && ref_len >= 0 && ref_start + ref_len <= n_ref && n_ref <= n_samples, NRX_E_ARG,
              "nrx_xcorr_abs: reference support [%d, %d) outside its %d samples (or longer than the received %d)", ref_start,
              ref_start + ref_len, n_ref, n_samples);
  hipLaunchKernelGGL(xcorr_abs_kernel, dim3((unsigned)n_samples), dim3(256), 0, (hipStream_t)stream, (const cd*)rx,
                     (const cd*)ref, n_samples, n_ref, ref_start, ref_len, nr, P, xc);
  NRX_CHECK_LAUNCH("nrx_xcorr_abs");
  return NRX_OK;
}
