// Tapped-delay-line channel (CDL/TDL): per-slot path gains, discrete CIR, frequency-domain channel matrix and
// the time-domain filtering of the transmit waveform (gfx950, float64).
//
// Replaces reference cdl.py:641-645,741-811,672-738,871-887 (CdlChannel.getPathGains: the time-varying part
// of TR 38.901 Eq 7.5-22/-29), channelmodel.py:321-354 (prepareForNextSlot: CIR + chanOffset),
// :362-400 (getChannelMatrix), :403-448 (applyToSignal).
#include "nrx_common.h"
#include "nrx_fft.h"

namespace {
using nrx::cx;
typedef cx<double> cd;

// ---------------------------------------------------------------------------------------------- CDL gains
// gains[b][t][rt][p]:  NLOS cluster n:  sum_m A[rt][n][m] * exp(j*2*pi*time[b][t]*nu[n][m])
//                      LOS (p = 0 when present):  Alos[rt] * exp(j*2*pi*time*nu_los)
// A already contains the field/polarisation/location terms, sqrt(P_n/M) and the output normalisations
// (channelmodel.py:451-469): everything that does not depend on time.
__global__ void __launch_bounds__(256)
cdl_gains_kernel(const cd* __restrict__ A, const double* __restrict__ nu, const cd* __restrict__ Alos, double nu_los,
                 const double* __restrict__ times, int n_t, int n_rt, int n_cl, int n_ray, cd* __restrict__ gains,
                 int64_t a_item_stride, int64_t nu_item_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* ph = (cd*)smem;  // [n_cl*n_ray] Doppler phasors of this (item, instant)
  const int bt = blockIdx.x;  // item * n_t + instant
  const double t = times[bt];
  // per-item ray coefficients (statistical models that redraw them for every slot, tdl.py:1043-1067); strides 0 = shared
  A += (int64_t)(bt / n_t) * a_item_stride;
  nu += (int64_t)(bt / n_t) * nu_item_stride;
  for (int i = threadIdx.x; i < n_cl * n_ray; i += blockDim.x) {
    double s, c;
    sincos((6.283185307179586 * t) * nu[i], &s, &c);  // cdl.py:887 exp(2j*pi*t*nu)
    ph[i] = cd(c, s);
  }
  __syncthreads();
  const int P = n_cl + (Alos ? 1 : 0);
  for (int o = threadIdx.x; o < n_rt * P; o += blockDim.x) {
    const int rt = o / P, p = o - rt * P;
    cd acc(0, 0);
    if (Alos && p == 0) {
      double s, c;
      sincos((6.283185307179586 * t) * nu_los, &s, &c);
      acc = Alos[rt] * cd(c, s);
    } else {
      const int n = p - (Alos ? 1 : 0);
      const cd* a = A + ((size_t)rt * n_cl + n) * n_ray;
      for (int m = 0; m < n_ray; ++m) nrx::cmac(acc, a[m], ph[n * n_ray + m]);
    }
    gains[((size_t)bt * n_rt + rt) * P + p] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------- CIR
// cir[b][t][rt][l] = sum_p gains[b][t][rt][p] * coeff[p][l]   (channelmodel.py:343)
__global__ void __launch_bounds__(256)
cir_kernel(const cd* __restrict__ gains, const double* __restrict__ coeff, int n_p, int cl, int64_t n_rows,
           cd* __restrict__ cir) {
  // one workgroup per (item, instant, rx, tx) row at a time: no 64-bit division per element, the row's gains are
  // wave-uniform loads
  for (int64_t row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const cd* gr = gains + (size_t)row * n_p;
    for (int l = threadIdx.x; l < cl; l += blockDim.x) {
      cd acc(0, 0);
#pragma unroll 8      // (eight (gain, coefficient) pairs in flight: the serial load-use chain was this kernel's time)
      for (int p = 0; p < n_p; ++p) {
        const double c = coeff[(size_t)p * cl + l];
        acc.re += gr[p].re * c;
        acc.im += gr[p].im * c;
      }
      cir[(size_t)row * cl + l] = acc;
    }
  }
}

// chanOffset[b] = argmax_l sum_r | sum_{c<nc, t} cir[b][c][r][t][l] |   (channelmodel.py:345-346; first max)
__global__ void __launch_bounds__(256)
chan_offset_kernel(const cd* __restrict__ cir, int n_t_total, int nc, int nr, int nt, int cl, int32_t* __restrict__ off) {
  __shared__ double bestv[256];
  __shared__ int besti[256];
  const int b = blockIdx.x;
  const cd* base = cir + (size_t)b * n_t_total * nr * nt * cl;
  double bv = -1.0;
  int bi = 0;
  for (int l = threadIdx.x; l < cl; l += blockDim.x) {
    double tot = 0;
    for (int r = 0; r < nr; ++r) {
      // (same left-to-right sum as before; the loads of a whole instant are issued together -- one dependent load per
      //  term made this one-workgroup-per-item kernel pure latency, 0.29 ms per 256 slots)
      cd s(0, 0);
#pragma unroll 2
      for (int c = 0; c < nc; ++c) {
        const cd* row = base + (((size_t)c * nr + r) * nt) * cl + l;
        cd v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = t < nt ? row[(size_t)t * cl] : cd(0, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (t < nt) s = s + v[t];
        for (int t = 8; t < nt; ++t) s = s + row[(size_t)t * cl];
      }
      tot += hypot(s.re, s.im);
    }
    if (tot > bv) { bv = tot; bi = l; }
  }
  bestv[threadIdx.x] = bv;
  besti[threadIdx.x] = bi;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = -1.0;
    int idx = 0;
    for (int i = 0; i < blockDim.x; ++i)
      if (bestv[i] > v || (bestv[i] == v && besti[i] < idx)) { v = bestv[i]; idx = besti[i]; }
    off[b] = idx;
  }
}

// ----------------------------------------------------------------------------------------- channel matrix
// H[b][c][k][rt] = FFT_nfft( cir[b][c][rt][.] circularly advanced by chanOffset )[(k - K/2) mod nfft]
// (channelmodel.py:381-399).  One LDS FFT per (b, c, rt); workgroups loop over tasks.
__global__ void __launch_bounds__(256)
chan_matrix_kernel(const cd* __restrict__ cir, int n_t_total, int nc, int n_rt, int cl, const int32_t* __restrict__ off,
                   int K, int nfft, int log2n, cd* __restrict__ H, int n_tasks, const cd* __restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* buf = (cd*)smem;
  for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const int rt = task % n_rt;
    const int c = (task / n_rt) % nc;
    const int b = task / (n_rt * nc);
    const cd* src = cir + (((size_t)b * n_t_total + c) * n_rt + rt) * cl;
    const int o = off[b];
    const int use = cl < nfft ? cl : nfft;  // longer responses are truncated (channelmodel.py:384-388)
    __syncthreads();
    for (int i = threadIdx.x; i < nfft; i += blockDim.x) buf[nrx::fft_idx(i)] = cd(0, 0);
    __syncthreads();
    for (int l = threadIdx.x; l < use; l += blockDim.x) buf[nrx::fft_idx((l - o + nfft) & (nfft - 1))] = src[l];
    __syncthreads();
    nrx::fft_dif_lds(buf, tw, nfft, log2n, false);
    cd* dst = H + (((size_t)b * nc + c) * K) * n_rt + rt;
    for (int k = threadIdx.x; k < K; k += blockDim.x)
      dst[(size_t)k * n_rt] = buf[nrx::fft_idx(nrx::fft_bitrev((k - K / 2 + nfft) & (nfft - 1), log2n))];
  }
}

// ------------------------------------------------------------------------------------------ applyToSignal
// y[b][r][n] = sum_t sum_l cir1[b][sym(n)][r][t][l] * x[b][t][n-l]   (zero state before the slot)
// which is what the reference's per-path lfilter + per-symbol gain mix evaluates (channelmodel.py:431-447):
// gains are piecewise constant per OUTPUT symbol (whole-symbol lengths, the extra (nc+1)-th set beyond the slot).
// Workgroup = one tile of TILE output samples inside ONE symbol, so the CIR is workgroup-uniform (scalar
// loads); the input tile (+cl-1 history) of all Nt antennas is staged in LDS.
struct TdGeom {
  int32_t n_sets;     // nc + 1
  int32_t start[17];  // first output sample of each gain set; start[n_sets] = ns
  int32_t tiles_per_set;
};
constexpr int TD_TILE = 256;

template <int NR>
__global__ void __launch_bounds__(TD_TILE)
apply_td_kernel(const cd* __restrict__ x, int nt, int64_t ns, const cd* __restrict__ cir1, int cl, TdGeom g,
                cd* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* xs = (cd*)smem;  // [nt][TD_TILE + cl - 1]
  const int b = blockIdx.y;
  const int set = blockIdx.x / g.tiles_per_set, tile = blockIdx.x % g.tiles_per_set;
  const int n0 = g.start[set] + tile * TD_TILE;
  const int n_end = g.start[set + 1];
  if (n0 >= n_end) return;
  const int span = TD_TILE + cl - 1;
  for (int i = threadIdx.x; i < nt * span; i += blockDim.x) {
    const int t = i / span, j = i - t * span;
    const int64_t n = (int64_t)n0 - (cl - 1) + j;
    xs[i] = (n >= 0 && n < ns) ? x[((size_t)b * nt + t) * ns + n] : cd(0, 0);
  }
  __syncthreads();
  const int n = n0 + threadIdx.x;
  if (n >= n_end) return;
  cd acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = cd(0, 0);
  const cd* cb = cir1 + ((size_t)b * g.n_sets + set) * NR * nt * cl;
  for (int t = 0; t < nt; ++t) {
    const cd* xr = xs + t * span + (cl - 1) + threadIdx.x;
    for (int l = 0; l < cl; ++l) {
      const cd xv = xr[-l];
#pragma unroll
      for (int r = 0; r < NR; ++r) nrx::cmac(acc[r], cb[((size_t)r * nt + t) * cl + l], xv);
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) y[((size_t)b * NR + r) * ns + n] = acc[r];
}

// Path form of the same filter (what the reference literally does, channelmodel.py:431-447): every path p is a
// flen-tap fractional-delay FIR at integer offset off_p, shared by all receive antennas; the per-symbol gains mix
// the filtered signals:  y[r][n] = sum_t sum_p g[sym(n)][r][t][p] * ( sum_k taps[p][k] * x[t][n - off_p - k] ).
// P*(2*flen + 4*Nr) real FMAs per (t, n) instead of 4*Nr*cl for the dense CIR form (4.6x fewer at CDL-C, 4x4).
template <int NR>
__global__ void __launch_bounds__(TD_TILE)
apply_td_paths_kernel(const cd* __restrict__ x, int nt, int64_t ns, const cd* __restrict__ gains1, int n_paths,
                      const double* __restrict__ taps, const int32_t* __restrict__ tap_off, int flen, int hist, TdGeom g,
                      cd* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* xs = (cd*)smem;  // [nt][TD_TILE + hist]
  const int b = blockIdx.y;
  const int set = blockIdx.x / g.tiles_per_set, tile = blockIdx.x % g.tiles_per_set;
  const int n0 = g.start[set] + tile * TD_TILE;
  const int n_end = g.start[set + 1];
  if (n0 >= n_end) return;
  const int span = TD_TILE + hist;
  for (int i = threadIdx.x; i < nt * span; i += blockDim.x) {
    const int t = i / span, j = i - t * span;
    const int64_t n = (int64_t)n0 - hist + j;
    xs[i] = (n >= 0 && n < ns) ? x[((size_t)b * nt + t) * ns + n] : cd(0, 0);
  }
  __syncthreads();
  const int n = n0 + threadIdx.x;
  if (n >= n_end) return;
  cd acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = cd(0, 0);
  const cd* gb = gains1 + ((size_t)b * g.n_sets + set) * NR * nt * n_paths;
  for (int t = 0; t < nt; ++t) {
    const cd* xr = xs + t * span + hist + threadIdx.x;
    for (int p = 0; p < n_paths; ++p) {
      const int off = tap_off[p];
      const double* c = taps + (size_t)p * flen;
      double fr = 0, fi = 0;
      for (int k = 0; k < flen; ++k) {
        const cd xv = xr[-(off + k)];
        fr += c[k] * xv.re;
        fi += c[k] * xv.im;
      }
      const cd f(fr, fi);
#pragma unroll
      for (int r = 0; r < NR; ++r) nrx::cmac(acc[r], gb[((size_t)r * nt + t) * n_paths + p], f);
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) y[((size_t)b * NR + r) * ns + n] = acc[r];
}

// Register-tiled path form: every thread produces R = 4 consecutive output samples, so one window of flen + 3 input
// samples feeds 4 FIR outputs (4.75 LDS reads per output sample and path instead of 16; the first version of this
// kernel was bound by LDS bandwidth).  The input tile is staged transposed, xs[t][s mod 4][s div 4], so that the
// 16-byte reads of neighbouring lanes (samples 4 apart) are contiguous.  Taps and gains are wave-uniform (scalar
// loads); arithmetic is explicit fma (the summation order already differs from SciPy's lfilter).
constexpr int TDP_R = 4;
constexpr int TDP_TILE = 128;  // threads per antenna group -> 512 output samples per workgroup
// Antenna groups per workgroup (more waves on the same staged tile).  The float64 FMA pipe issues at its full rate only
// with >= 3 waves per SIMD (profiles/r2_f64_issue_rates.txt: 4.4 cycles per wave instruction with two waves, 2.9-3.2
// with three or four), and the 54 KB tile allows two workgroups per CU: four groups = 8 waves per workgroup = 4 per SIMD.
// (6-wave workgroups do not work: the first lands 2,2,1,1 on the SIMDs and a second one is never co-scheduled.)
constexpr int TDP_GROUPS = 4;
constexpr int TDP_FLEN = 16;   // channelmodel.py:249-289: 16-tap fractional-delay filters

// One (tx antenna, path) term for the 4 samples of a thread.  q0..q3 are the thread's addresses of window elements 0..3 in
// the transposed tile: element m sits in sub-array (u + m) & 3, i.e. in the sub-array of element m & 3, (m >> 2) cells further
// on -- so the four alignments u & 3 of a window share ONE body whose LDS offsets are immediates, and the alignment only
// enters the four base addresses (4 VALU additions per term; the merged four-case body the compiler built before spent 109
// address additions per 192 FMAs, and this kernel is VALU-issue bound: one wave64 instruction per SIMD per 4 cycles).
template <int NR>
__device__ __forceinline__ void tdp4_term(const char* __restrict__ q0, const char* __restrict__ q1, const char* __restrict__ q2,
                                          const char* __restrict__ q3, const double* __restrict__ c,
                                          const cd* __restrict__ gv, int gstride, double (&ar)[NR][TDP_R],
                                          double (&ai)[NR][TDP_R]) {
  constexpr int R = TDP_R, W = TDP_FLEN + R - 1;
  double wr[W], wi[W];
#pragma unroll
  for (int m = 0; m < W; ++m) {
    const int mm = m;
    const char* qk = (mm & 3) == 0 ? q0 : ((mm & 3) == 1 ? q1 : ((mm & 3) == 2 ? q2 : q3));
    const cd v = *(const cd*)(qk + (mm >> 2) * (int)sizeof(cd));
    wr[m] = v.re;
    wi[m] = v.im;
  }
  double fr[R], fi[R];
#pragma unroll
  for (int j = 0; j < R; ++j) fr[j] = fi[j] = 0.0;
#pragma unroll
  for (int k = 0; k < TDP_FLEN; ++k) {
    const double ck = c[k];
#pragma unroll
    for (int j = 0; j < R; ++j) {   // x[n + j - off - k] = window element 15 + j - k
      fr[j] = fma(ck, wr[TDP_FLEN - 1 + j - k], fr[j]);
      fi[j] = fma(ck, wi[TDP_FLEN - 1 + j - k], fi[j]);
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const cd g = gv[(size_t)r * gstride];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      ar[r][j] = fma(g.re, fr[j], fma(-g.im, fi[j], ar[r][j]));
      ai[r][j] = fma(g.re, fi[j], fma(g.im, fr[j], ai[r][j]));
    }
  }
}

template <int NR, bool FLAT = false>      // FLAT: the groups share the (antenna, path) terms round robin (Nt not a multiple of GROUPS)
__global__ void __launch_bounds__(TDP_TILE * TDP_GROUPS, 4)   // four waves per SIMD (two workgroups per CU): <= 128 VGPRs
apply_td_paths4_kernel(const cd* __restrict__ x, int nt, int64_t ns, const cd* __restrict__ gains1, int n_paths,
                       const double* __restrict__ taps, const int32_t* __restrict__ tap_off, int hist, TdGeom g,
                       cd* __restrict__ y, double* __restrict__ pow_acc, int pow_nfft) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* xs = (cd*)smem;  // [nt][R][Q]
  constexpr int R = TDP_R;
  constexpr int PW = TDP_TILE / 64;      // waves of antenna group 0 = power-sum partials per workgroup
  const int b = blockIdx.y;
  const int set = blockIdx.x / g.tiles_per_set, tile = blockIdx.x % g.tiles_per_set;
  const int n0 = g.start[set] + tile * (TDP_TILE * R);
  const int n_end = g.start[set + 1];
  if (n0 >= n_end) {
    if (pow_acc && threadIdx.x < 3 * PW) pow_acc[((size_t)b * gridDim.x + blockIdx.x) * 3 * PW + threadIdx.x] = 0.0;
    return;
  }
  const int span = TDP_TILE * R + hist;   // hist is a multiple of R (host)
  const int Q = span / R;
  for (int i = threadIdx.x; i < nt * span; i += blockDim.x) {
    const int t = i / span, j = i - t * span;
    const int64_t n = (int64_t)n0 - hist + j;
    xs[(t * R + (j & (R - 1))) * Q + (j >> 2)] = (n >= 0 && n < ns) ? x[((size_t)b * nt + t) * ns + n] : cd(0, 0);
  }
  __syncthreads();
  // the workgroup's waves split the transmit antennas between them (same staged tile, twice the waves per CU)
  const int ts = (int)threadIdx.x % TDP_TILE;
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / TDP_TILE);      // whole waves: taps / gains addresses are scalar
  const int n = n0 + R * ts;
  double ar[NR][R], ai[NR][R];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j) ar[r][j] = ai[r][j] = 0.0;
  const cd* gb = gains1 + ((size_t)b * g.n_sets + set) * NR * nt * n_paths;
  if (n < n_end) {
    const char* xl = (const char*)(xs + ts);          // the thread's cell of sub-array 0 of antenna 0
    auto term = [&](int t, int p) __attribute__((always_inline)) {
      const int u = hist - tap_off[p] - (TDP_FLEN - 1);   // window element m is sample  R*ts + u + m  of the tile
      const double* c = taps + (size_t)p * TDP_FLEN;
      // byte offsets (wave-uniform) of window elements 0..3: sub-array (u + k) & 3, cell (u + k) >> 2
      const int o0 = ((t * R + ((u + 0) & 3)) * Q + ((u + 0) >> 2)) * (int)sizeof(cd);
      const int o1 = ((t * R + ((u + 1) & 3)) * Q + ((u + 1) >> 2)) * (int)sizeof(cd);
      const int o2 = ((t * R + ((u + 2) & 3)) * Q + ((u + 2) >> 2)) * (int)sizeof(cd);
      const int o3 = ((t * R + ((u + 3) & 3)) * Q + ((u + 3) >> 2)) * (int)sizeof(cd);
      const cd* gv = gb + (size_t)t * n_paths + p;
      tdp4_term<NR>(xl + o0, xl + o1, xl + o2, xl + o3, c, gv, nt * n_paths, ar, ai);
    };
    if constexpr (!FLAT) {           // a group takes whole transmit antennas
      for (int t = grp; t < nt; t += TDP_GROUPS)
        for (int p = 0; p < n_paths; ++p) term(t, p);
    } else {
      // fewer (or not a multiple of GROUPS) transmit antennas -- SISO, two layers: the groups share the (antenna, path) terms
      // round robin instead, or all but nt of them would idle (cfg1: 7.3 ms of a 13.8 ms step with three groups idle)
      const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)n_paths - 1) / (uint32_t)n_paths);   // q / n_paths = umulhi(q, magic) for q * n_paths < 2^32
      for (int q = grp; q < nt * n_paths; q += TDP_GROUPS) {
        const int t = (int)__umulhi((uint32_t)q, magic);
        term(t, q - t * n_paths);
      }
    }
  }
  // sum the groups' partial results through LDS (the staged tile is dead now): real parts, then imaginary parts, so that
  // the (GROUPS-1) x NR x R x TILE doubles of a phase fit the tile's space
  double* part = (double*)smem;   // [GROUPS - 1][NR * R][TDP_TILE]
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
    __syncthreads();
    if (grp > 0) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < R; ++j) part[((grp - 1) * NR * R + r * R + j) * TDP_TILE + ts] = ph == 0 ? ar[r][j] : ai[r][j];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          double a = ph == 0 ? ar[r][j] : ai[r][j];
#pragma unroll
          for (int q = 0; q < TDP_GROUPS - 1; ++q) a += part[(q * NR * R + r * R + j) * TDP_TILE + ts];
          if (ph == 0) ar[r][j] = a; else ai[r][j] = a;
        }
    }
  }
  if (grp == 0 && n < n_end) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (n + j < n_end) y[((size_t)b * NR + r) * ns + n + j] = cd(ar[r][j], ai[r][j]);
  }
  // Power sums of the received signal over the CP-stripped samples (Waveform.getRePower, waveform.py:107-117: the nfft samples
  // of every symbol from half its cyclic prefix on; np.var needs sum x and sum |x|^2) while the samples are in registers: one
  // (sum re, sum im, sum |x|^2) triple per WAVE of antenna group 0 -- no barrier, no atomics -- into pow_acc[item][workgroup]
  // [wave][3], summed in that order by var_finish_kernel (reproducible).  Saves the separate pass over the waveform.
  if (pow_acc && grp == 0) {
    double sr = 0, si = 0, s2 = 0;
    if (n < n_end && set < g.n_sets - 1) {                  // (the last gain set is the tail beyond the slot: no symbol)
      const int s0 = g.start[set];
      const int off = (int)rint((double)(n_end - s0 - pow_nfft) * 0.5);      // np.round(cpLen * 0.5)
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int pos = n + j - s0;
          if (n + j < n_end && pos >= off && pos < off + pow_nfft) {
            sr += ar[r][j];
            si += ai[r][j];
            s2 += ar[r][j] * ar[r][j] + ai[r][j] * ai[r][j];
          }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
      sr += __shfl_xor(sr, o, 64);
      si += __shfl_xor(si, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      double* o3 = pow_acc + (((size_t)b * gridDim.x + blockIdx.x) * PW + (threadIdx.x >> 6)) * 3;
      o3[0] = sr;
      o3[1] = si;
      o3[2] = s2;
    }
  }
}

// The same filter in float32 for the float32 waveform chain (the opt-in fast mode; the reference computes in complex128, so this
// is NOT the parity path).  A complex sample is one 64-bit register pair and every multiply-add a packed one (v_pk_fma_f32: both
// halves per instruction): tap x sample = 1 instruction, gain x filtered sample = 2 (g.re * (fr, fi) + g.im * (-fi, fr)) -- 96
// per (tx antenna, path) term and 4 outputs against 192 in float64 -- and the window reads are 8 bytes.  Same tiling, same
// transposed LDS tile, same group split; the groups' partial sums cross LDS in one phase (a sample is 8 bytes).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef nrx::cx<float> cf;

template <int NR>
__device__ __forceinline__ void tdp4f_term(const char* __restrict__ q0, const char* __restrict__ q1, const char* __restrict__ q2,
                                           const char* __restrict__ q3, const float* __restrict__ c, const f2* __restrict__ gv,
                                           int gstride, f2 (&acc)[NR][TDP_R]) {
  constexpr int R = TDP_R, W = TDP_FLEN + R - 1;
  f2 w[W];
#pragma unroll
  for (int m = 0; m < W; ++m) {
    const char* qk = (m & 3) == 0 ? q0 : ((m & 3) == 1 ? q1 : ((m & 3) == 2 ? q2 : q3));
    w[m] = *(const f2*)(qk + (m >> 2) * (int)sizeof(f2));
  }
  f2 f[R];
#pragma unroll
  for (int j = 0; j < R; ++j) f[j] = (f2)(0.0f);
#pragma unroll
  for (int k = 0; k < TDP_FLEN; ++k) {
    const float ck = c[k];
#pragma unroll
    for (int j = 0; j < R; ++j) f[j] = __builtin_elementwise_fma((f2)(ck), w[TDP_FLEN - 1 + j - k], f[j]);
  }
  f2 fs[R];
#pragma unroll
  for (int j = 0; j < R; ++j) fs[j] = (f2){-f[j].y, f[j].x};
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const f2 g = gv[(size_t)r * gstride];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      acc[r][j] = __builtin_elementwise_fma((f2)(g.x), f[j], acc[r][j]);
      acc[r][j] = __builtin_elementwise_fma((f2)(g.y), fs[j], acc[r][j]);
    }
  }
}

template <int NR, bool FLAT = false>
__global__ void __launch_bounds__(TDP_TILE * TDP_GROUPS, 4)
apply_td_paths4f_kernel(const f2* __restrict__ x, int nt, int64_t ns, const f2* __restrict__ gains1, int n_paths,
                        const float* __restrict__ taps, const int32_t* __restrict__ tap_off, int hist, TdGeom g,
                        f2* __restrict__ y, double* __restrict__ pow_acc, int pow_nfft) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f2* xs = (f2*)smem;  // [nt][R][Q]
  constexpr int R = TDP_R;
  constexpr int PW = TDP_TILE / 64;
  const int b = blockIdx.y;
  const int set = blockIdx.x / g.tiles_per_set, tile = blockIdx.x % g.tiles_per_set;
  const int n0 = g.start[set] + tile * (TDP_TILE * R);
  const int n_end = g.start[set + 1];
  if (n0 >= n_end) {
    if (pow_acc && threadIdx.x < 3 * PW) pow_acc[((size_t)b * gridDim.x + blockIdx.x) * 3 * PW + threadIdx.x] = 0.0;
    return;
  }
  const int span = TDP_TILE * R + hist;   // hist is a multiple of R (host)
  const int Q = span / R;
  for (int i = threadIdx.x; i < nt * span; i += blockDim.x) {
    const int t = i / span, j = i - t * span;
    const int64_t n = (int64_t)n0 - hist + j;
    xs[(t * R + (j & (R - 1))) * Q + (j >> 2)] = (n >= 0 && n < ns) ? x[((size_t)b * nt + t) * ns + n] : (f2)(0.0f);
  }
  __syncthreads();
  const int ts = (int)threadIdx.x % TDP_TILE;
  const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / TDP_TILE);
  const int n = n0 + R * ts;
  f2 acc[NR][R];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j) acc[r][j] = (f2)(0.0f);
  const f2* gb = gains1 + ((size_t)b * g.n_sets + set) * NR * nt * n_paths;
  if (n < n_end) {
    const char* xl = (const char*)(xs + ts);
    auto term = [&](int t, int p) __attribute__((always_inline)) {
      const int u = hist - tap_off[p] - (TDP_FLEN - 1);
      const float* c = taps + (size_t)p * TDP_FLEN;
      const int o0 = ((t * R + ((u + 0) & 3)) * Q + ((u + 0) >> 2)) * (int)sizeof(f2);
      const int o1 = ((t * R + ((u + 1) & 3)) * Q + ((u + 1) >> 2)) * (int)sizeof(f2);
      const int o2 = ((t * R + ((u + 2) & 3)) * Q + ((u + 2) >> 2)) * (int)sizeof(f2);
      const int o3 = ((t * R + ((u + 3) & 3)) * Q + ((u + 3) >> 2)) * (int)sizeof(f2);
      tdp4f_term<NR>(xl + o0, xl + o1, xl + o2, xl + o3, c, gb + (size_t)t * n_paths + p, nt * n_paths, acc);
    };
    if constexpr (!FLAT) {
      for (int t = grp; t < nt; t += TDP_GROUPS)
        for (int p = 0; p < n_paths; ++p) term(t, p);
    } else {      // (see the float64 kernel)
      const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)n_paths - 1) / (uint32_t)n_paths);   // q / n_paths = umulhi(q, magic) for q * n_paths < 2^32
      for (int q = grp; q < nt * n_paths; q += TDP_GROUPS) {
        const int t = (int)__umulhi((uint32_t)q, magic);
        term(t, q - t * n_paths);
      }
    }
  }
  f2* part = (f2*)smem;   // [GROUPS - 1][NR * R][TDP_TILE]: the staged tile is dead now
  __syncthreads();
  if (grp > 0) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < R; ++j) part[((grp - 1) * NR * R + r * R + j) * TDP_TILE + ts] = acc[r][j];
  }
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
      for (int q = 0; q < TDP_GROUPS - 1; ++q) acc[r][j] += part[(q * NR * R + r * R + j) * TDP_TILE + ts];
    }
  if (n < n_end) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (n + j < n_end) y[((size_t)b * NR + r) * ns + n + j] = acc[r][j];
  }
  if (pow_acc) {   // power sums over the CP-stripped samples, in double, one triple per wave (see the float64 kernel)
    double sr = 0, si = 0, s2 = 0;
    if (n < n_end && set < g.n_sets - 1) {
      const int s0 = g.start[set];
      const int off = (int)rint((double)(n_end - s0 - pow_nfft) * 0.5);
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int pos = n + j - s0;
          if (n + j < n_end && pos >= off && pos < off + pow_nfft) {
            const double vr = acc[r][j].x, vi = acc[r][j].y;
            sr += vr;
            si += vi;
            s2 += vr * vr + vi * vi;
          }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
      sr += __shfl_xor(sr, o, 64);
      si += __shfl_xor(si, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      double* o3 = pow_acc + (((size_t)b * gridDim.x + blockIdx.x) * PW + (threadIdx.x >> 6)) * 3;
      o3[0] = sr;
      o3[1] = si;
      o3[2] = s2;
    }
  }
}

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

}  // namespace

extern "C" int32_t nrx_cdl_gains_f64(const void* A, const double* nu, const void* A_los, double nu_los,
                                     const double* times, int32_t n_items, int32_t n_t, int32_t n_rx, int32_t n_tx,
                                     int32_t n_clusters, int32_t n_rays, void* gains, void* stream) {
  NRX_REQUIRE(A && nu && times && gains, NRX_E_ARG, "nrx_cdl_gains: NULL buffer");
  NRX_REQUIRE(n_t >= 1 && n_rx >= 1 && n_tx >= 1 && n_clusters >= 1 && n_rays >= 1 && n_items >= 0, NRX_E_ARG,
              "nrx_cdl_gains: bad sizes");
  if (n_items == 0) return NRX_OK;
  const size_t lds = sizeof(cd) * (size_t)n_clusters * n_rays;
  NRX_REQUIRE(lds <= 64 * 1024, NRX_E_UNSUPPORTED, "nrx_cdl_gains: too many rays (%d x %d)", n_clusters, n_rays);
  hipLaunchKernelGGL(cdl_gains_kernel, dim3(n_items * n_t), dim3(256), lds, (hipStream_t)stream, (const cd*)A, nu,
                     (const cd*)A_los, nu_los, times, n_t, n_rx * n_tx, n_clusters, n_rays, (cd*)gains, (int64_t)0, (int64_t)0);
  NRX_CHECK_LAUNCH("nrx_cdl_gains");
  return NRX_OK;
}

// The same with ray coefficients of their own for every item: A (n_items, Nr, Nt, N, M), nu (n_items, N, M).
extern "C" int32_t nrx_cdl_gains_items_f64(const void* A, const double* nu, const void* A_los, double nu_los,
                                           const double* times, int32_t n_items, int32_t n_t, int32_t n_rx, int32_t n_tx,
                                           int32_t n_clusters, int32_t n_rays, void* gains, void* stream) {
  NRX_REQUIRE(A && nu && times && gains, NRX_E_ARG, "nrx_cdl_gains_items: NULL buffer");
  NRX_REQUIRE(n_t >= 1 && n_rx >= 1 && n_tx >= 1 && n_clusters >= 1 && n_rays >= 1 && n_items >= 0, NRX_E_ARG,
              "nrx_cdl_gains_items: bad sizes");
  if (n_items == 0) return NRX_OK;
  const size_t lds = sizeof(cd) * (size_t)n_clusters * n_rays;
  NRX_REQUIRE(lds <= 64 * 1024, NRX_E_UNSUPPORTED, "nrx_cdl_gains_items: too many rays (%d x %d)", n_clusters, n_rays);
  hipLaunchKernelGGL(cdl_gains_kernel, dim3(n_items * n_t), dim3(256), lds, (hipStream_t)stream, (const cd*)A, nu,
                     (const cd*)A_los, nu_los, times, n_t, n_rx * n_tx, n_clusters, n_rays, (cd*)gains,
                     (int64_t)n_rx * n_tx * n_clusters * n_rays, (int64_t)n_clusters * n_rays);
  NRX_CHECK_LAUNCH("nrx_cdl_gains_items");
  return NRX_OK;
}

extern "C" int32_t nrx_cir_f64(const void* gains, const double* coeff, int32_t n_items, int32_t n_t, int32_t nc,
                               int32_t n_rx, int32_t n_tx, int32_t n_paths, int32_t cl, void* cir, int32_t* chan_offset,
                               void* stream) {
  NRX_REQUIRE(gains && coeff && cir, NRX_E_ARG, "nrx_cir: NULL buffer");
  NRX_REQUIRE(n_t >= 1 && nc >= 1 && nc <= n_t && n_rx >= 1 && n_tx >= 1 && n_paths >= 1 && cl >= 1 && n_items >= 0,
              NRX_E_ARG, "nrx_cir: bad sizes");
  if (n_items == 0) return NRX_OK;
  const int64_t rows = (int64_t)n_items * n_t * n_rx * n_tx;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cir_kernel, dim3((unsigned)(rows < (1 << 20) ? rows : (1 << 20))), dim3(256), 0, st, (const cd*)gains, coeff,
                     n_paths, cl, rows, (cd*)cir);
  if (chan_offset)
    hipLaunchKernelGGL(chan_offset_kernel, dim3(n_items), dim3(256), 0, st, (const cd*)cir, n_t, nc, n_rx, n_tx, cl,
                       chan_offset);
  NRX_CHECK_LAUNCH("nrx_cir");
  return NRX_OK;
}

extern "C" int32_t nrx_channel_matrix_f64(const void* cir, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                                          int32_t n_tx, int32_t cl, const int32_t* chan_offset, int32_t K, int32_t nfft,
                                          void* H, void* stream) {
  NRX_REQUIRE(cir && chan_offset && H, NRX_E_ARG, "nrx_channel_matrix: NULL buffer");
  NRX_REQUIRE(nfft >= 64 && nfft <= 8192 && (nfft & (nfft - 1)) == 0, NRX_E_ARG, "nrx_channel_matrix: nfft must be a power of two");
  NRX_REQUIRE(K > 0 && K <= nfft && nc >= 1 && nc <= n_t && cl >= 1, NRX_E_SHAPE, "nrx_channel_matrix: bad sizes");
  const int n_tasks = n_items * nc * n_rx * n_tx;
  if (n_tasks == 0) return NRX_OK;
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_channel_matrix: FFT twiddle table unavailable");
  const size_t lds = sizeof(cd) * nrx::fft_lds_elems((size_t)nfft);
  (void)hipFuncSetAttribute((const void*)chan_matrix_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(chan_matrix_kernel, dim3(n_tasks < 2048 ? n_tasks : 2048), dim3(256), lds, (hipStream_t)stream,
                     (const cd*)cir, n_t, nc, n_rx * n_tx, cl, chan_offset, K, nfft, ilog2(nfft), (cd*)H, n_tasks, tw);
  NRX_CHECK_LAUNCH("nrx_channel_matrix");
  return NRX_OK;
}

extern "C" int32_t nrx_apply_td_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* cir1,
                                    int32_t n_sets, int32_t n_rx, int32_t cl, const int32_t* set_lens, void* y,
                                    void* stream) {
  NRX_REQUIRE(x && cir1 && set_lens && y, NRX_E_ARG, "nrx_apply_td: NULL buffer");
  NRX_REQUIRE(n_sets >= 1 && n_sets <= 16 && n_tx >= 1 && cl >= 1 && ns > 0 && n_items >= 0, NRX_E_ARG, "nrx_apply_td: bad sizes");
  NRX_REQUIRE(n_rx == 1 || n_rx == 2 || n_rx == 4 || n_rx == 8, NRX_E_UNSUPPORTED, "nrx_apply_td: Nr must be 1, 2, 4 or 8 (got %d)", n_rx);
  if (n_items == 0) return NRX_OK;
  TdGeom g;
  g.n_sets = n_sets;
  int64_t s = 0;
  int maxlen = 0;
  for (int i = 0; i < n_sets; ++i) {
    g.start[i] = (int32_t)(s < ns ? s : ns);
    s += set_lens[i];
  }
  g.start[n_sets] = (int32_t)ns;  // samples past the listed symbols keep the last gain set (channelmodel.py:443-446)
  for (int i = 0; i < n_sets; ++i) {
    const int len = g.start[i + 1] - g.start[i];
    maxlen = len > maxlen ? len : maxlen;
  }
  // when ns is shorter than the listed symbols, trailing sets are empty: start[i] == ns
  for (int i = 1; i < n_sets; ++i)
    if (g.start[i] > g.start[n_sets]) g.start[i] = g.start[n_sets];
  g.tiles_per_set = (maxlen + TD_TILE - 1) / TD_TILE;
  if (g.tiles_per_set < 1) g.tiles_per_set = 1;
  const size_t lds = sizeof(cd) * (size_t)n_tx * (TD_TILE + cl - 1);
  NRX_REQUIRE(lds <= 160 * 1024, NRX_E_UNSUPPORTED, "nrx_apply_td: Nt*cl too large for LDS staging (%zu B)", lds);
  const dim3 grid(g.tiles_per_set * n_sets, n_items);
  hipStream_t st = (hipStream_t)stream;
#define NRX_TD_CASE(NR)                                                                                             \
  case NR:                                                                                                          \
    (void)hipFuncSetAttribute((const void*)apply_td_kernel<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(apply_td_kernel<NR>, grid, dim3(TD_TILE), lds, st, (const cd*)x, n_tx, ns, (const cd*)cir1, cl, g, \
                       (cd*)y);                                                                                     \
    break;
  switch (n_rx) {
    NRX_TD_CASE(1)
    NRX_TD_CASE(2)
    NRX_TD_CASE(4)
    NRX_TD_CASE(8)
  }
#undef NRX_TD_CASE
  NRX_CHECK_LAUNCH("nrx_apply_td");
  return NRX_OK;
}

// ------------------------------------------------------------------------------ channel at selected subcarriers
// Direct DFT of the CIR at n_k consecutive subcarriers starting at k0 (same result as nrx_channel_matrix at those
// bins; channelmodel.py:381-399) -- used for the wideband SVD precoder, which by the reference's grouping quirk
// only looks at the first PRB (pdsch.py:1142-1163), so the full K-point transform is not needed in the
// time-domain path.  H_sub: (n_items, nc, n_k, n_rx*n_tx).
namespace {
__global__ void __launch_bounds__(256)
chan_matrix_sub_kernel(const cd* __restrict__ cir, int n_t_total, int nc, int n_rt, int cl, const int32_t* __restrict__ off,
                       int K, int nfft, int k0, int n_k, cd* __restrict__ H, int64_t total, const cd* __restrict__ tw) {
  const int tws = nrx::FFT_TW_N / nfft;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int rt = (int)(g % n_rt);
    const int k = (int)((g / n_rt) % n_k);
    const int c = (int)((g / ((int64_t)n_rt * n_k)) % nc);
    const int b = (int)(g / ((int64_t)n_rt * n_k * nc));
    const cd* src = cir + (((size_t)b * n_t_total + c) * n_rt + rt) * cl;
    const int bin = (k0 + k - K / 2 + nfft) & (nfft - 1);
    const int o = off[b];
    const int use = cl < nfft ? cl : nfft;
    cd acc(0, 0);
#pragma unroll 4      // (four taps and their twiddles in flight; one wave per impulse response with a cross-lane sum was slower)
    for (int l = 0; l < use; ++l) {
      const int pos = (l - o + nfft) & (nfft - 1);
      const int ph = (int)(((int64_t)bin * pos) & (nfft - 1));
      // exp(-2 pi i ph / nfft) from the shared twiddle table (entries k < N/2; the other half by W[k + N/2] = -W[k])
      const int ti = ph * tws;
      cd w = tw[ti & (nrx::FFT_TW_N / 2 - 1)];
      if (ti >= nrx::FFT_TW_N / 2) w = cd(-w.re, -w.im);
      nrx::cmac(acc, src[l], w);
    }
    H[g] = acc;
  }
}

// SVD precoder (pdsch.py:1125-1131): mean of H over the given (symbols x subcarriers) block, right singular
// vectors of the Nr x Nt mean via a cyclic Jacobi eigen-decomposition of G = Hm Hm^H (Nr x Nr Hermitian),
// v_i = Hm^H u_i / sigma_i, F = V[:, :nl] / sqrt(nl).  One thread per batch item (the matrices are tiny).
// Column phases are implementation defined (LAPACK's are too); H*F -- all that the link sees -- is not.
constexpr int PMAX = 8;
__global__ void svd_precoder_kernel(const cd* __restrict__ Hblk, int n_avg, int nr, int nt, int nl, cd* __restrict__ F,
                                    int n_items) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_items) return;
  cd Hm[PMAX][32];
  const cd* src = Hblk + (size_t)b * n_avg * nr * nt;
  for (int r = 0; r < nr; ++r)
    for (int t = 0; t < nt; ++t) {
      cd s(0, 0);
      for (int a = 0; a < n_avg; ++a) s = s + src[((size_t)a * nr + r) * nt + t];
      Hm[r][t] = cd(s.re / n_avg, s.im / n_avg);
    }
  cd G[PMAX][PMAX], U[PMAX][PMAX];
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nr; ++j) {
      cd s(0, 0);
      for (int t = 0; t < nt; ++t) nrx::cmacc(s, Hm[j][t], Hm[i][t]);  // conj(Hm[j][t]) * Hm[i][t]
      G[i][j] = s;
      U[i][j] = cd(i == j ? 1.0 : 0.0, 0.0);
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double offn = 0, diagn = 0;
    for (int p = 0; p < nr; ++p) {
      diagn += G[p][p].re * G[p][p].re;
      for (int q = p + 1; q < nr; ++q) offn += nrx::norm2(G[p][q]);
    }
    if (offn <= 1e-34 * diagn || offn < 1e-300) break;   // off-diagonal mass below double-precision resolution
    for (int p = 0; p < nr; ++p)
      for (int q = p + 1; q < nr; ++q) {
        const double apq = sqrt(nrx::norm2(G[p][q]));
        if (apq < 1e-300) continue;
        // complex Jacobi rotation zeroing G[p][q]
        const cd ph(G[p][q].re / apq, G[p][q].im / apq);  // e^{j arg(G_pq)}
        const double tau = (G[q][q].re - G[p][p].re) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        // J = [[c, s*ph],[-s*conj(ph), c]] applied as G <- J^H G J, U <- U J
        for (int k = 0; k < nr; ++k) {  // columns p,q
          const cd gkp = G[k][p], gkq = G[k][q];
          G[k][p] = cd(c * gkp.re - s * (gkq.re * ph.re + gkq.im * ph.im), c * gkp.im - s * (gkq.im * ph.re - gkq.re * ph.im));
          G[k][q] = cd(s * (gkp.re * ph.re - gkp.im * ph.im) + c * gkq.re, s * (gkp.re * ph.im + gkp.im * ph.re) + c * gkq.im);
          const cd ukp = U[k][p], ukq = U[k][q];
          U[k][p] = cd(c * ukp.re - s * (ukq.re * ph.re + ukq.im * ph.im), c * ukp.im - s * (ukq.im * ph.re - ukq.re * ph.im));
          U[k][q] = cd(s * (ukp.re * ph.re - ukp.im * ph.im) + c * ukq.re, s * (ukp.re * ph.im + ukp.im * ph.re) + c * ukq.im);
        }
        for (int k = 0; k < nr; ++k) {  // rows p,q
          const cd gpk = G[p][k], gqk = G[q][k];
          G[p][k] = cd(c * gpk.re - s * (gqk.re * ph.re - gqk.im * ph.im), c * gpk.im - s * (gqk.re * ph.im + gqk.im * ph.re));
          G[q][k] = cd(s * (gpk.re * ph.re + gpk.im * ph.im) + c * gqk.re, s * (gpk.im * ph.re - gpk.re * ph.im) + c * gqk.im);
        }
      }
  }
  // order by descending eigenvalue, emit the first nl right singular vectors
  int order[PMAX];
  for (int i = 0; i < nr; ++i) order[i] = i;
  for (int i = 0; i < nr; ++i)
    for (int j = i + 1; j < nr; ++j)
      if (G[order[j]][order[j]].re > G[order[i]][order[i]].re) { const int tmp = order[i]; order[i] = order[j]; order[j] = tmp; }
  const double inv_nl = 1.0 / sqrt((double)nl);
  for (int i = 0; i < nl; ++i) {
    const int e = order[i];
    double nrm = 0;
    cd v[32];
    for (int t = 0; t < nt; ++t) {
      cd s(0, 0);
      for (int r = 0; r < nr; ++r) nrx::cmacc(s, Hm[r][t], U[r][e]);  // (Hm^H u)_t
      v[t] = s;
      nrm += nrx::norm2(s);
    }
    const double sc = nrm > 0 ? inv_nl / sqrt(nrm) : 0.0;
    for (int t = 0; t < nt; ++t) F[((size_t)b * nt + t) * nl + i] = cd(v[t].re * sc, v[t].im * sc);
  }
}

// Same computation with compile-time antenna counts: one wavefront per batch item, the lanes share the averaging
// (the only part with real memory traffic), every lane then runs the tiny Jacobi in registers (no scratch).
template <int NR, int NT>
__global__ void __launch_bounds__(64) svd_precoder_wave_kernel(const cd* __restrict__ Hblk, int n_avg, int nl,
                                                               cd* __restrict__ F, int n_items) {
  const int b = blockIdx.x, lane = threadIdx.x;
  constexpr int E = NR * NT;
  const cd* src = Hblk + (size_t)b * n_avg * E;
  // lane l sums entries l, l+64, ... of the (n_avg x E) block in order; E divides 64 or is a multiple of it
  cd Hm[NR][NT];
  {
    constexpr int GRP = E <= 64 ? 64 / E : 1;          // lanes per matrix entry
    const int e = E <= 64 ? lane % E : 0, gi = E <= 64 ? lane / E : 0;
    double sr[E > 64 ? E / 64 : 1], si[E > 64 ? E / 64 : 1];
    if constexpr (E <= 64) {
      double ar = 0, ai = 0;
      for (int a = gi; a < n_avg; a += GRP) {
        const cd v = src[(size_t)a * E + e];
        ar += v.re;
        ai += v.im;
      }
      // fold the GRP partial sums: lanes e, e+E, e+2E, ...
      for (int o = 32; o >= E; o >>= 1) {
        ar += __shfl_xor(ar, o, 64);
        ai += __shfl_xor(ai, o, 64);
      }
      sr[0] = ar;
      si[0] = ai;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          Hm[r][t] = cd(__shfl(sr[0], r * NT + t, 64) / n_avg, __shfl(si[0], r * NT + t, 64) / n_avg);
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          double ar = 0, ai = 0;
          for (int a = lane; a < n_avg; a += 64) {
            const cd v = src[(size_t)a * E + r * NT + t];
            ar += v.re;
            ai += v.im;
          }
          for (int o = 32; o > 0; o >>= 1) {
            ar += __shfl_xor(ar, o, 64);
            ai += __shfl_xor(ai, o, 64);
          }
          Hm[r][t] = cd(ar / n_avg, ai / n_avg);
        }
    }
  }
  cd G[NR][NR], U[NR][NR];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      cd s(0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) nrx::cmacc(s, Hm[j][t], Hm[i][t]);
      G[i][j] = s;
      U[i][j] = cd(i == j ? 1.0 : 0.0, 0.0);
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double offn = 0, diagn = 0;
#pragma unroll
    for (int p = 0; p < NR; ++p) {
      diagn += G[p][p].re * G[p][p].re;
#pragma unroll
      for (int q = p + 1; q < NR; ++q) offn += nrx::norm2(G[p][q]);
    }
    if (offn <= 1e-34 * diagn || offn < 1e-300) break;
#pragma unroll
    for (int p = 0; p < NR; ++p)
#pragma unroll
      for (int q = p + 1; q < NR; ++q) {
        const double apq = sqrt(nrx::norm2(G[p][q]));
        if (apq >= 1e-300) {
          const cd ph(G[p][q].re / apq, G[p][q].im / apq);
          const double tau = (G[q][q].re - G[p][p].re) / (2.0 * apq);
          const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const cd gkp = G[k][p], gkq = G[k][q];
            G[k][p] = cd(c * gkp.re - s * (gkq.re * ph.re + gkq.im * ph.im), c * gkp.im - s * (gkq.im * ph.re - gkq.re * ph.im));
            G[k][q] = cd(s * (gkp.re * ph.re - gkp.im * ph.im) + c * gkq.re, s * (gkp.re * ph.im + gkp.im * ph.re) + c * gkq.im);
            const cd ukp = U[k][p], ukq = U[k][q];
            U[k][p] = cd(c * ukp.re - s * (ukq.re * ph.re + ukq.im * ph.im), c * ukp.im - s * (ukq.im * ph.re - ukq.re * ph.im));
            U[k][q] = cd(s * (ukp.re * ph.re - ukp.im * ph.im) + c * ukq.re, s * (ukp.re * ph.im + ukp.im * ph.re) + c * ukq.im);
          }
#pragma unroll
          for (int k = 0; k < NR; ++k) {
            const cd gpk = G[p][k], gqk = G[q][k];
            G[p][k] = cd(c * gpk.re - s * (gqk.re * ph.re - gqk.im * ph.im), c * gpk.im - s * (gqk.re * ph.im + gqk.im * ph.re));
            G[q][k] = cd(s * (gpk.re * ph.re + gpk.im * ph.im) + c * gqk.re, s * (gpk.im * ph.re - gpk.re * ph.im) + c * gqk.im);
          }
        }
      }
  }
  // rank of each eigenvalue (descending, earlier index first on ties -- the selection order of the generic kernel)
  const double inv_nl = 1.0 / sqrt((double)nl);
#pragma unroll
  for (int e = 0; e < NR; ++e) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) rank += (G[j][j].re > G[e][e].re || (G[j][j].re == G[e][e].re && j < e)) ? 1 : 0;
    if (rank < nl) {
      cd v[NT];
      double nrm = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        cd s(0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) nrx::cmacc(s, Hm[r][t], U[r][e]);
        v[t] = s;
        nrm += nrx::norm2(s);
      }
      const double sc = nrm > 0 ? inv_nl / sqrt(nrm) : 0.0;
      if (lane == 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) F[((size_t)b * NT + t) * nl + rank] = cd(v[t].re * sc, v[t].im * sc);
    }
  }
}

// Hest[b][lk][r][p] = sum_t H[b][lk][r][t] * F[b][t][p]   ("perfect" CSI of the BLER notebook: H @ F)
__global__ void __launch_bounds__(256)
eff_channel_kernel(const cd* __restrict__ H, const cd* __restrict__ F, int64_t f_stride, int lk, int nr, int nt, int nl,
                   cd* __restrict__ out, int64_t total) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(g % nl);
    const int r = (int)((g / nl) % nr);
    const int64_t i = g / ((int64_t)nl * nr);  // b*lk + re
    const int b = (int)(i / lk);
    const cd* h = H + ((size_t)i * nr + r) * nt;
    const cd* f = F + (size_t)b * f_stride;
    cd acc(0, 0);
    for (int t = 0; t < nt; ++t) nrx::cmac(acc, h[t], f[t * nl + p]);
    out[g] = acc;
  }
}
}  // namespace

extern "C" int32_t nrx_channel_matrix_sub_f64(const void* cir, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                                              int32_t n_tx, int32_t cl, const int32_t* chan_offset, int32_t K, int32_t nfft,
                                              int32_t k0, int32_t n_k, void* H, void* stream) {
  NRX_REQUIRE(cir && chan_offset && H, NRX_E_ARG, "nrx_channel_matrix_sub: NULL buffer");
  NRX_REQUIRE(nfft >= 64 && (nfft & (nfft - 1)) == 0 && K > 0 && K <= nfft, NRX_E_ARG, "nrx_channel_matrix_sub: bad nfft/K");
  NRX_REQUIRE(k0 >= 0 && n_k >= 1 && k0 + n_k <= K && nc >= 1 && nc <= n_t, NRX_E_SHAPE, "nrx_channel_matrix_sub: bad range");
  const int64_t total = (int64_t)n_items * nc * n_k * n_rx * n_tx;
  if (total == 0) return NRX_OK;
  NRX_REQUIRE(nfft <= nrx::FFT_TW_N, NRX_E_UNSUPPORTED, "nrx_channel_matrix_sub: nfft > %d", nrx::FFT_TW_N);
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_channel_matrix_sub: FFT twiddle table unavailable");
  hipLaunchKernelGGL(chan_matrix_sub_kernel, dim3(nrx::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const cd*)cir, n_t, nc, n_rx * n_tx, cl, chan_offset, K, nfft, k0, n_k, (cd*)H, total, tw);
  NRX_CHECK_LAUNCH("nrx_channel_matrix_sub");
  return NRX_OK;
}

namespace {
// ------------------------------------------------------------------------ wideband precoder folded into the path gains
// gl[b][c][r][l][p] = sum_t gains[b][c][r][t][p] * F[b][t][l].  The wideband precoder (Grid.precode, grid.py:505-516) is one
// Nt x Nl matrix for every subcarrier, so it commutes with the IFFT, the cyclic prefix and the windowing, all of which act
// on each waveform alone: sum_t g[r][t][p] FIR_p(IFFT(sum_l F[t][l] X_l)) = sum_l (sum_t g[r][t][p] F[t][l]) FIR_p(IFFT(X_l)).
// The time-domain link therefore modulates the Nl LAYER grids (one read of each row, no mixing in the load) and filters
// them with these folded gains (channelmodel.py:431-447 with the layers as inputs).
__global__ void __launch_bounds__(256)
fold_precoder_kernel(const cd* __restrict__ gains, const cd* __restrict__ F, int64_t f_stride, int n_sets, int nr, int nt, int nl,
                     int P, int64_t total, cd* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    int64_t q = i / P;
    const int l = (int)(q % nl);
    q /= nl;                                   // (b * n_sets + c) * nr + r
    const int64_t b = q / ((int64_t)n_sets * nr);
    const cd* g = gains + q * nt * P + p;
    const cd* f = F + b * f_stride + l;
    cd acc(0, 0);
    for (int t = 0; t < nt; ++t) nrx::cmac(acc, g[(size_t)t * P], f[(size_t)t * nl]);
    out[i] = acc;
  }
}

// ------------------------------------------------------------------------------------ fused channel set-up (round 3)
// chanOffset and the channel matrix at n_k subcarriers straight from the path gains: what cir_kernel + chan_offset_kernel +
// chan_matrix_sub_kernel do in three launches through a (n, T, Nr, Nt, cl) CIR in HBM (1.3 MB per slot written and read twice;
// latency-bound kernels: 0.80 ms per 256 slots), here in one workgroup per item with the tap matrix and the item's gains in LDS
// and no CIR in memory.  The time-domain link needs the CIR for nothing else (the filter runs in path form).  Every value is
// produced by the SAME expression in the same order as in the three kernels -- cir = sum_p gain*coeff left to right, the
// offset's sums c outer / t inner, the matrix' sum over taps with nrx::cmac -- so offset and matrix are bit-identical.
constexpr int CS_THREADS = 512;
constexpr int CS_NK = 12;      // subcarriers (one PRB)
typedef const cd __attribute__((address_space(4))) * cgain_t;      // read-only in this kernel: wave-uniform rows come by scalar loads
template <int P>      // paths: a template parameter so that the per-tap / per-row register arrays are fully unrolled without guards
__global__ void __launch_bounds__(CS_THREADS)
chan_setup_kernel(const cd* __restrict__ gains, const double* __restrict__ coeff, int n_t_total, int nc, int nr, int nt,
                  int cl, int K, int nfft, int k0, int32_t* __restrict__ off_out, cd* __restrict__ H,
                  const cd* __restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* csT = (double*)smem;                              // [cl][P] tap matrix, transposed: a tap's coefficients are contiguous
  cd* W = (cd*)(csT + (size_t)P * cl);                      // [use][12] twiddles of (tap, subcarrier) once chanOffset is known
  __shared__ double bestv[CS_THREADS / 64];
  __shared__ int besti[CS_THREADS / 64];
  __shared__ int off_s;
  const int b = blockIdx.x, tid = threadIdx.x, n_rt = nr * nt;
  for (int i = tid; i < P * cl; i += CS_THREADS) {
    const int p = i / cl, l = i - p * cl;
    csT[(size_t)l * P + p] = coeff[i];
  }
  const cd* gb = gains + (size_t)b * n_t_total * n_rt * P;
  __syncthreads();
  // ---- chanOffset = argmax_l sum_r | sum_{c<nc, t} cir[c][r][t][l] |   (channelmodel.py:345-346; first max): one tap per thread,
  // its coefficients in registers, the gains of a (c, r, t) row wave-uniform
  double bv = -1.0;
  int bi = 0;
  for (int l = tid; l < cl; l += CS_THREADS) {
    double cf[P];
#pragma unroll
    for (int p = 0; p < P; ++p) cf[p] = csT[(size_t)l * P + p];
    double tot = 0;
    for (int r = 0; r < nr; ++r) {
      cd s(0, 0);
      for (int c = 0; c < nc; ++c)
        for (int t = 0; t < nt; ++t) {
          const cgain_t gr = (cgain_t)(gb + ((size_t)c * n_rt + r * nt + t) * P);
          cd acc(0, 0);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            acc.re += gr[p].re * cf[p];
            acc.im += gr[p].im * cf[p];
          }
          s = s + acc;
        }
      tot += hypot(s.re, s.im);
    }
    if (tot > bv) { bv = tot; bi = l; }
  }
  // first maximum over the threads (largest value, smallest tap among equals -- np.argmax): inside the wave by shuffles, then
  // over the waves.  (A serial scan of 512 LDS entries by one thread was 84 k cycles of the workgroup's 1.46 M.)
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
    for (int i = 0; i < CS_THREADS / 64; ++i)
      if (bestv[i] > v || (bestv[i] == v && besti[i] < idx)) { v = bestv[i]; idx = besti[i]; }
    off_out[b] = idx;
    off_s = idx;
  }
  __syncthreads();
  // ---- channel matrix at subcarriers k0 .. k0 + 11 (channelmodel.py:362-400 at those bins: direct DFT of the CIR placed
  // circularly shifted by chanOffset).  The twiddle of (tap, bin) is the same for every row: tabulated once.
  const int o = off_s;
  const int tws = nrx::FFT_TW_N / nfft;
  const int use = cl < nfft ? cl : nfft;
  for (int i = tid; i < use * CS_NK; i += CS_THREADS) {
    const int l = i / CS_NK, k = i - l * CS_NK;
    const int bin = (k0 + k - K / 2 + nfft) & (nfft - 1);
    const int pos = (l - o + nfft) & (nfft - 1);
    const int ph = (int)(((int64_t)bin * pos) & (nfft - 1));
    const int ti = ph * tws;
    cd w = tw[ti & (nrx::FFT_TW_N / 2 - 1)];
    if (ti >= nrx::FFT_TW_N / 2) w = cd(-w.re, -w.im);
    W[i] = w;
  }
  __syncthreads();
  // one (instant, rx, tx) row and one half of the 12 bins per thread: the row's gains in registers, the CIR value of a tap (computed
  // by both halves, same expression) feeds the thread's 6 bins.  (With all 12 bins per thread the 24 gains, 12 accumulators and a
  // tap's coefficients and twiddles did not fit in the 256 registers of a 512-thread workgroup: 9 gains were re-read from scratch
  // for every tap, 1.1 M of the workgroup's 1.46 M cycles; and only 224 of the 512 threads had a row.)
  constexpr int NKH = CS_NK / 2;
  for (int it = tid; it < 2 * nc * n_rt; it += CS_THREADS) {
    const int row = it >> 1, kh = (it & 1) * NKH;
    cd g[P];
#pragma unroll
    for (int p = 0; p < P; ++p) g[p] = gb[(size_t)row * P + p];
    cd acc[NKH];
#pragma unroll
    for (int k = 0; k < NKH; ++k) acc[k] = cd(0, 0);
    for (int l = 0; l < use; ++l) {
      // a tap's coefficients (same address in every lane: broadcast) and twiddles, all read before the first is used
      // (the scheduler otherwise sinks each LDS read to its multiply: a dozen exposed LDS latencies per tap)
      double cf[P];
      cd wv[NKH];
#pragma unroll
      for (int p = 0; p < P; ++p) cf[p] = csT[(size_t)l * P + p];
#pragma unroll
      for (int k = 0; k < NKH; ++k) wv[k] = W[l * CS_NK + kh + k];
      __builtin_amdgcn_sched_barrier(0);
      cd v(0, 0);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        v.re += g[p].re * cf[p];
        v.im += g[p].im * cf[p];
      }
#pragma unroll
      for (int k = 0; k < NKH; ++k) nrx::cmac(acc[k], v, wv[k]);
    }
    const int c = row / n_rt, rt = row - c * n_rt;
#pragma unroll
    for (int k = 0; k < NKH; ++k) H[(((size_t)b * nc + c) * CS_NK + kh + k) * n_rt + rt] = acc[k];
  }
}
}  // namespace

extern "C" int32_t nrx_chan_setup_f64(const void* gains, const double* coeff, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                                      int32_t n_tx, int32_t n_paths, int32_t cl, int32_t K, int32_t nfft, int32_t k0, int32_t n_k,
                                      int32_t* chan_offset, void* H, void* stream) {
  NRX_REQUIRE(gains && coeff && chan_offset && H, NRX_E_ARG, "nrx_chan_setup: NULL buffer");
  NRX_REQUIRE(n_t >= 1 && nc >= 1 && nc <= n_t && n_rx >= 1 && n_tx >= 1 && n_paths >= 1 && cl >= 1 && n_items >= 0,
              NRX_E_ARG, "nrx_chan_setup: bad sizes");
  NRX_REQUIRE(nfft >= 64 && (nfft & (nfft - 1)) == 0 && K > 0 && K <= nfft && k0 >= 0 && k0 + n_k <= K, NRX_E_ARG,
              "nrx_chan_setup: bad nfft / K / subcarrier range");
  const size_t lds = sizeof(double) * (size_t)n_paths * cl + sizeof(cd) * (size_t)(cl < nfft ? cl : nfft) * CS_NK;
  const bool built = n_paths == 13 || n_paths == 14 || n_paths == 15 || n_paths == 23 || n_paths == 24;   // the CDL / TDL profiles
  if (n_k != CS_NK || !built || nfft > nrx::FFT_TW_N || lds > 150 * 1024) {
    ::nrx::set_error("nrx_chan_setup: built for %d subcarriers, 13/14/15/23/24 paths and <= 150 KB of tap matrix + twiddles (got n_k %d, %d paths, %zu B)",
                     CS_NK, n_k, n_paths, lds);
    return NRX_E_UNSUPPORTED;      // the caller runs nrx_cir_f64 + nrx_channel_matrix_sub_f64
  }
  if (n_items == 0) return NRX_OK;
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_chan_setup: FFT twiddle table unavailable");
#define NRX_CS_CASE(PP)                                                                                                      \
  case PP:                                                                                                                   \
    (void)hipFuncSetAttribute((const void*)chan_setup_kernel<PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    hipLaunchKernelGGL(chan_setup_kernel<PP>, dim3(n_items), dim3(CS_THREADS), lds, (hipStream_t)stream, (const cd*)gains, coeff, \
                       n_t, nc, n_rx, n_tx, cl, K, nfft, k0, chan_offset, (cd*)H, tw);                                       \
    break;
  switch (n_paths) {
    NRX_CS_CASE(13)
    NRX_CS_CASE(14)
    NRX_CS_CASE(15)
    NRX_CS_CASE(23)
    NRX_CS_CASE(24)
  }
#undef NRX_CS_CASE
  NRX_CHECK_LAUNCH("nrx_chan_setup");
  return NRX_OK;
}

extern "C" int32_t nrx_svd_precoder_f64(const void* H_block, int32_t n_items, int32_t n_avg, int32_t n_rx, int32_t n_tx,
                                        int32_t n_layers, void* F, void* stream) {
  NRX_REQUIRE(H_block && F, NRX_E_ARG, "nrx_svd_precoder: NULL buffer");
  NRX_REQUIRE(n_rx >= 1 && n_rx <= PMAX && n_tx >= 1 && n_tx <= 32 && n_avg >= 1, NRX_E_UNSUPPORTED,
              "nrx_svd_precoder: Nr <= 8 and Nt <= 32 supported (got %dx%d)", n_rx, n_tx);
  NRX_REQUIRE(n_layers >= 1 && n_layers <= n_rx && n_layers <= n_tx, NRX_E_ARG, "nrx_svd_precoder: layers must be <= min(Nr,Nt)");
  if (n_items == 0) return NRX_OK;
#define NRX_SVD_CASE(NR, NT)                                                                                        \
  if (n_rx == NR && n_tx == NT) {                                                                                   \
    hipLaunchKernelGGL((svd_precoder_wave_kernel<NR, NT>), dim3(n_items), dim3(64), 0, (hipStream_t)stream,         \
                       (const cd*)H_block, n_avg, n_layers, (cd*)F, n_items);                                       \
    NRX_CHECK_LAUNCH("nrx_svd_precoder");                                                                           \
    return NRX_OK;                                                                                                  \
  }
  NRX_SVD_CASE(1, 1) NRX_SVD_CASE(1, 2) NRX_SVD_CASE(1, 4) NRX_SVD_CASE(2, 2) NRX_SVD_CASE(2, 4) NRX_SVD_CASE(4, 2)
  NRX_SVD_CASE(4, 4) NRX_SVD_CASE(2, 8) NRX_SVD_CASE(4, 8) NRX_SVD_CASE(2, 1) NRX_SVD_CASE(4, 1)
#undef NRX_SVD_CASE
  hipLaunchKernelGGL(svd_precoder_kernel, dim3((n_items + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const cd*)H_block,
                     n_avg, n_rx, n_tx, n_layers, (cd*)F, n_items);
  NRX_CHECK_LAUNCH("nrx_svd_precoder");
  return NRX_OK;
}

extern "C" int32_t nrx_effective_channel_f64(const void* H, const void* F, int64_t f_stride, int32_t n_items, int32_t lk,
                                             int32_t n_rx, int32_t n_tx, int32_t n_layers, void* out, void* stream) {
  NRX_REQUIRE(H && F && out, NRX_E_ARG, "nrx_effective_channel: NULL buffer");
  const int64_t total = (int64_t)n_items * lk * n_rx * n_layers;
  if (total == 0) return NRX_OK;
  hipLaunchKernelGGL(eff_channel_kernel, dim3(nrx::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const cd*)H, (const cd*)F, f_stride, lk, n_rx, n_tx, n_layers, (cd*)out, total);
  NRX_CHECK_LAUNCH("nrx_effective_channel");
  return NRX_OK;
}

static int32_t td_geom(const int32_t* set_lens, int32_t n_sets, int64_t ns, TdGeom* g) {
  g->n_sets = n_sets;
  int64_t s = 0;
  int maxlen = 0;
  for (int i = 0; i < n_sets; ++i) {
    g->start[i] = (int32_t)(s < ns ? s : ns);
    s += set_lens[i];
  }
  g->start[n_sets] = (int32_t)ns;  // samples past the listed symbols keep the last gain set (channelmodel.py:443-446)
  for (int i = 0; i < n_sets; ++i) {
    const int len = g->start[i + 1] - g->start[i];
    maxlen = len > maxlen ? len : maxlen;
  }
  g->tiles_per_set = (maxlen + TD_TILE - 1) / TD_TILE;
  if (g->tiles_per_set < 1) g->tiles_per_set = 1;
  return NRX_OK;
}

static int32_t apply_td_paths_impl(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                   int32_t n_sets, int32_t n_rx, int32_t n_paths, const double* taps,
                                   const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                   void* y, void* stream, double* pow_acc, int32_t pow_nfft, int64_t pow_capacity, int32_t* n_part) {
  NRX_REQUIRE(x && gains1 && taps && tap_off && set_lens && y, NRX_E_ARG, "nrx_apply_td_paths: NULL buffer");
  NRX_REQUIRE(n_sets >= 1 && n_sets <= 16 && n_tx >= 1 && n_paths >= 1 && flen >= 1 && hist >= 0 && ns > 0 && n_items >= 0,
              NRX_E_ARG, "nrx_apply_td_paths: bad sizes");
  NRX_REQUIRE(n_rx == 1 || n_rx == 2 || n_rx == 4 || n_rx == 8, NRX_E_UNSUPPORTED, "nrx_apply_td_paths: Nr must be 1, 2, 4 or 8 (got %d)", n_rx);
  if (n_items == 0) return NRX_OK;
  TdGeom g;
  td_geom(set_lens, n_sets, ns, &g);
  hipStream_t st = (hipStream_t)stream;
  {
    // register-tiled kernel: 4 output samples per thread
    const int hist4 = (hist + TDP_R - 1) / TDP_R * TDP_R;
    size_t lds4 = sizeof(cd) * (size_t)n_tx * (TDP_TILE * TDP_R + hist4);
    const size_t part4 = sizeof(double) * (TDP_GROUPS - 1) * (size_t)n_rx * TDP_R * TDP_TILE;   // group partial sums (one phase) reuse the tile
    if (lds4 < part4) lds4 = part4;
    if (flen == TDP_FLEN && lds4 <= 80 * 1024 && n_rx <= 4) {
      int maxlen = 0;
      for (int i = 0; i < g.n_sets; ++i) maxlen = g.start[i + 1] - g.start[i] > maxlen ? g.start[i + 1] - g.start[i] : maxlen;
      TdGeom g4 = g;
      g4.tiles_per_set = (maxlen + TDP_TILE * TDP_R - 1) / (TDP_TILE * TDP_R);
      if (g4.tiles_per_set < 1) g4.tiles_per_set = 1;
      const dim3 grid4(g4.tiles_per_set * n_sets, n_items);
      if (pow_acc) {
        const int64_t need = (int64_t)n_items * grid4.x * (TDP_TILE / 64) * 3;
        NRX_REQUIRE(pow_capacity >= need, NRX_E_SHAPE, "nrx_apply_td_paths_pow: pow_acc needs %lld doubles", (long long)need);
        *n_part = (int32_t)(grid4.x * (TDP_TILE / 64));
      }
#define NRX_TDP4_LAUNCH(NR, FL)                                                                                           \
    (void)hipFuncSetAttribute((const void*)apply_td_paths4_kernel<NR, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); \
    hipLaunchKernelGGL((apply_td_paths4_kernel<NR, FL>), grid4, dim3(TDP_TILE * TDP_GROUPS), lds4, st, (const cd*)x, n_tx, ns, \
                       (const cd*)gains1, n_paths, taps, tap_off, hist4, g4, (cd*)y, pow_acc, pow_nfft)
#define NRX_TDP4_CASE(NR)                                                                                                  \
  case NR:                                                                                                                 \
    if (n_tx % TDP_GROUPS == 0) { NRX_TDP4_LAUNCH(NR, false); } else { NRX_TDP4_LAUNCH(NR, true); }                        \
    break;
      switch (n_rx) {
        NRX_TDP4_CASE(1)
        NRX_TDP4_CASE(2)
        NRX_TDP4_CASE(4)
      }
#undef NRX_TDP4_CASE
#undef NRX_TDP4_LAUNCH
      NRX_CHECK_LAUNCH("nrx_apply_td_paths");
      return NRX_OK;
    }
  }
  if (pow_acc) {
    ::nrx::set_error("nrx_apply_td_paths_pow: power sums come with the register-tiled kernel only (16-tap filters, Nr <= 4)");
    return NRX_E_UNSUPPORTED;       // the caller runs nrx_apply_td_paths_f64 + nrx_noise_level_f64
  }
  const size_t lds = sizeof(cd) * (size_t)n_tx * (TD_TILE + hist);
  NRX_REQUIRE(lds <= 160 * 1024, NRX_E_UNSUPPORTED, "nrx_apply_td_paths: Nt*delay too large for LDS staging (%zu B)", lds);
  const dim3 grid(g.tiles_per_set * n_sets, n_items);
#define NRX_TDP_CASE(NR)                                                                                                  \
  case NR:                                                                                                                \
    (void)hipFuncSetAttribute((const void*)apply_td_paths_kernel<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(apply_td_paths_kernel<NR>, grid, dim3(TD_TILE), lds, st, (const cd*)x, n_tx, ns, (const cd*)gains1,  \
                       n_paths, taps, tap_off, flen, hist, g, (cd*)y);                                                   \
    break;
  switch (n_rx) {
    NRX_TDP_CASE(1)
    NRX_TDP_CASE(2)
    NRX_TDP_CASE(4)
    NRX_TDP_CASE(8)
  }
#undef NRX_TDP_CASE
  NRX_CHECK_LAUNCH("nrx_apply_td_paths");
  return NRX_OK;
}

extern "C" int32_t nrx_apply_td_paths_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                          int32_t n_sets, int32_t n_rx, int32_t n_paths, const double* taps,
                                          const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                          void* y, void* stream) {
  return apply_td_paths_impl(x, n_items, n_tx, ns, gains1, n_sets, n_rx, n_paths, taps, tap_off, flen, hist, set_lens, y, stream,
                             nullptr, 0, 0, nullptr);
}

// ... and the power sums of its output over the CP-stripped samples (see the kernel): pow_acc (n_items, *n_part, 3) float64,
// to be finished by nrx_noise_level_finish_f64.  NRX_E_UNSUPPORTED when the geometry has no register-tiled instantiation.
extern "C" int32_t nrx_apply_td_paths_pow_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                              int32_t n_sets, int32_t n_rx, int32_t n_paths, const double* taps,
                                              const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                              void* y, int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part,
                                              void* stream) {
  NRX_REQUIRE(pow_acc && n_part && nfft > 0, NRX_E_ARG, "nrx_apply_td_paths_pow: NULL pow_acc / n_part");
  for (int i = 0; i + 1 < n_sets; ++i)
    NRX_REQUIRE(set_lens && set_lens[i] > nfft, NRX_E_ARG, "nrx_apply_td_paths_pow: a symbol (%d samples) is not longer than nfft", set_lens[i]);
  return apply_td_paths_impl(x, n_items, n_tx, ns, gains1, n_sets, n_rx, n_paths, taps, tap_off, flen, hist, set_lens, y, stream,
                             pow_acc, nfft, pow_capacity, n_part);
}

// float32 waveform chain (fast mode): x (n_items, n_tx, ns) / y (n_items, n_rx, ns) complex64, gains1 complex64, taps float32.
// pow_acc may be NULL (no power sums).  Register-tiled kernel only: NRX_E_UNSUPPORTED for filter lengths other than 16 or
// Nr > 4 (the caller converts and takes the float64 entry).
extern "C" int32_t nrx_apply_td_paths_pow_f32(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                              int32_t n_sets, int32_t n_rx, int32_t n_paths, const float* taps,
                                              const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                              void* y, int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part,
                                              void* stream) {
  NRX_REQUIRE(x && gains1 && taps && tap_off && set_lens && y, NRX_E_ARG, "nrx_apply_td_paths_pow_f32: NULL buffer");
  NRX_REQUIRE(n_sets >= 1 && n_sets <= 16 && n_tx >= 1 && n_paths >= 1 && flen >= 1 && hist >= 0 && ns > 0 && n_items >= 0,
              NRX_E_ARG, "nrx_apply_td_paths_pow_f32: bad sizes");
  NRX_REQUIRE(!pow_acc || (n_part && nfft > 0), NRX_E_ARG, "nrx_apply_td_paths_pow_f32: power sums need n_part and nfft");
  if (pow_acc)
    for (int i = 0; i + 1 < n_sets; ++i)
      NRX_REQUIRE(set_lens[i] > nfft, NRX_E_ARG, "nrx_apply_td_paths_pow_f32: a symbol (%d samples) is not longer than nfft", set_lens[i]);
  if (n_items == 0) return NRX_OK;
  TdGeom g;
  td_geom(set_lens, n_sets, ns, &g);
  const int hist4 = (hist + TDP_R - 1) / TDP_R * TDP_R;
  size_t lds4 = sizeof(f2) * (size_t)n_tx * (TDP_TILE * TDP_R + hist4);
  const size_t part4 = sizeof(f2) * (TDP_GROUPS - 1) * (size_t)n_rx * TDP_R * TDP_TILE;
  if (lds4 < part4) lds4 = part4;
  if (!(flen == TDP_FLEN && lds4 <= 80 * 1024 && (n_rx == 1 || n_rx == 2 || n_rx == 4))) {
    ::nrx::set_error("nrx_apply_td_paths_pow_f32: built for 16-tap filters and Nr in {1, 2, 4} (flen %d, Nr %d)", flen, n_rx);
    return NRX_E_UNSUPPORTED;
  }
  int maxlen = 0;
  for (int i = 0; i < g.n_sets; ++i) maxlen = g.start[i + 1] - g.start[i] > maxlen ? g.start[i + 1] - g.start[i] : maxlen;
  g.tiles_per_set = (maxlen + TDP_TILE * TDP_R - 1) / (TDP_TILE * TDP_R);
  if (g.tiles_per_set < 1) g.tiles_per_set = 1;
  const dim3 grid4(g.tiles_per_set * n_sets, n_items);
  if (pow_acc) {
    const int64_t need = (int64_t)n_items * grid4.x * (TDP_TILE / 64) * 3;
    NRX_REQUIRE(pow_capacity >= need, NRX_E_SHAPE, "nrx_apply_td_paths_pow_f32: pow_acc needs %lld doubles", (long long)need);
    *n_part = (int32_t)(grid4.x * (TDP_TILE / 64));
  }
  hipStream_t st = (hipStream_t)stream;
#define NRX_TDP4F_LAUNCH(NR, FL)                                                                                          \
    (void)hipFuncSetAttribute((const void*)apply_td_paths4f_kernel<NR, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4); \
    hipLaunchKernelGGL((apply_td_paths4f_kernel<NR, FL>), grid4, dim3(TDP_TILE * TDP_GROUPS), lds4, st, (const f2*)x, n_tx, ns, \
                       (const f2*)gains1, n_paths, taps, tap_off, hist4, g, (f2*)y, pow_acc, nfft)
#define NRX_TDP4F_CASE(NR)                                                                                                 \
  case NR:                                                                                                                 \
    if (n_tx % TDP_GROUPS == 0) { NRX_TDP4F_LAUNCH(NR, false); } else { NRX_TDP4F_LAUNCH(NR, true); }                      \
    break;
  switch (n_rx) {
    NRX_TDP4F_CASE(1)
    NRX_TDP4F_CASE(2)
    NRX_TDP4F_CASE(4)
  }
#undef NRX_TDP4F_CASE
#undef NRX_TDP4F_LAUNCH
  NRX_CHECK_LAUNCH("nrx_apply_td_paths_pow_f32");
  return NRX_OK;
}

extern "C" int32_t nrx_fold_precoder_f64(const void* gains, const void* F, int64_t f_stride, int32_t n_items, int32_t n_sets,
                                         int32_t n_rx, int32_t n_tx, int32_t n_layers, int32_t n_paths, void* out, void* stream) {
  NRX_REQUIRE(gains && F && out, NRX_E_ARG, "nrx_fold_precoder: NULL buffer");
  NRX_REQUIRE(n_items >= 0 && n_sets >= 1 && n_rx >= 1 && n_tx >= 1 && n_layers >= 1 && n_layers <= n_tx && n_paths >= 1, NRX_E_ARG,
              "nrx_fold_precoder: bad sizes");
  NRX_REQUIRE(f_stride == 0 || f_stride >= (int64_t)n_tx * n_layers, NRX_E_SHAPE, "nrx_fold_precoder: f_stride too small");
  const int64_t total = (int64_t)n_items * n_sets * n_rx * n_layers * n_paths;
  if (total == 0) return NRX_OK;
  hipLaunchKernelGGL(fold_precoder_kernel, dim3(nrx::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const cd*)gains,
                     (const cd*)F, f_stride, n_sets, n_rx, n_tx, n_layers, n_paths, total, (cd*)out);
  NRX_CHECK_LAUNCH("nrx_fold_precoder");
  return NRX_OK;
}
