// Time-domain channel filtering by overlap-save (gfx950, float64): the same operator as nrx_apply_td_paths_f64
//
//     y[b][r][n] = sum_t sum_p g[b][set(n)][r][t][p] * sum_k taps[p][k] * x[b][t][n - off_p - k]        (channelmodel.py:403-448)
//
// evaluated per gain set (= OFDM symbol of the OUTPUT sample, channelmodel.py:431-447) as a circular convolution of 1024-sample
// blocks: Y_r = IFFT( sum_t H[r][t] . FFT(x_t block) ), H[r][t] = sum_p g[r][t][p] . C_p with C_p the 1024-point spectrum of path p's
// fractional-delay filter at its integer offset (a constant of the channel: nrx_td_path_spectra_f64, once per link).  A block
// yields 1024 - hist valid output samples (hist = longest path: max(off) + flen - 1), so the filter costs ~9x fewer flops than the
// path form at CDL-C 4x4 (24 paths x 16 taps), and -- the point of this kernel -- nothing but x and y touches HBM:
//
//   * one workgroup per (item, gain set); it walks the set's blocks.  H (Nr x Nt spectra) lives in REGISTERS: in the pointwise
//     phase a thread owns 8/NX positions of all NX x NX spectra (128 VGPRs at 4x4), so the 256 KB of spectra are never stored.
//   * a 1024-point transform = three radix-8 passes (stages 0-2, 3-5, 6-8) + stage 9; NX transforms run side by side, 128 threads
//     each, 8 points per thread and pass.  Forward = decimation in frequency (natural -> bit-reversed positions), inverse =
//     decimation in time with conjugated twiddles (bit-reversed -> natural): no permutation anywhere, the pointwise product is by
//     POSITION.  Stage 9 of both directions is twiddle-free and is done inside the pointwise phase (a thread owns position pairs).
//   * pass A of the forward transform reads x from global memory, pass A of the inverse writes y (and sums its power): the
//     waveforms never sit in LDS; LDS holds the NX transform buffers only (padded 1 element per 16: 68 KB at NX = 4).
//   * every twiddle a thread needs is the same for all its transforms: 9 complex values in registers, loaded once.
//   * 1/1024 is folded into C_p (exact).
//
// Accuracy: a float64 FFT convolution -- |error| ~ 1e-15 of the block's largest sample, like the path form's own rounding; LLRs
// agree with the path form within 1e-9 (tests/test_gpu_phy.py::test_overlap_save_filter_equals_the_path_form).
#include "nrx_common.h"
#include "nrx_fft.h"

namespace {
using nrx::cx;
typedef cx<double> cd;

constexpr int OSN = 1024;
constexpr int OS_ELEMS = OSN + OSN / 16;      // one element of padding per 16
__device__ __forceinline__ constexpr int osi(int e) { return e + (e >> 4); }

struct OsGeom {
  int32_t n_sets;
  int32_t start[17];      // first output sample of each gain set; start[n_sets] = ns
};

__device__ __forceinline__ cd cadd(cd a, cd b) { return cd(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cd csub(cd a, cd b) { return cd(a.re - b.re, a.im - b.im); }
// a * w  |  a * conj(w)
template <bool INV> __device__ __forceinline__ cd twmul(cd a, cd w) {
  if constexpr (!INV) return cd(fma(a.re, w.re, -(a.im * w.im)), fma(a.re, w.im, a.im * w.re));
  else return cd(fma(a.re, w.re, a.im * w.im), fma(a.im, w.re, -(a.re * w.im)));
}
// a * W8^M (forward) | a * conj(W8^M)
template <int M, bool INV> __device__ __forceinline__ cd rot8(cd a) {
  constexpr double R = 0.70710678118654752440;
  if constexpr (M == 0) return a;
  else if constexpr (M == 2) return INV ? cd(-a.im, a.re) : cd(a.im, -a.re);
  else if constexpr (M == 1) return INV ? cd((a.re - a.im) * R, (a.re + a.im) * R) : cd((a.re + a.im) * R, (a.im - a.re) * R);
  else return INV ? cd((-a.re - a.im) * R, (a.re - a.im) * R) : cd((a.im - a.re) * R, (-a.re - a.im) * R);
}

// Three decimation-in-frequency stages on 8 points at spacing q: stage a pairs (m, m+4) with twiddle w1*W8^m, stage b pairs
// (m, m+2) with w2*W4^(m&1), stage c pairs (m, m+1) with w4 (w1 = W^(lo*2^s), w2 = w1^2, w4 = w2^2 from the table).
#define OS_DIF_A(M)                                  \
  {                                                  \
    const cd u = x[M], v = x[M + 4];                 \
    x[M] = cadd(u, v);                               \
    x[M + 4] = twmul<false>(rot8<M, false>(csub(u, v)), w1); \
  }
#define OS_DIF_B(M)                                  \
  {                                                  \
    const cd u = x[M], v = x[M + 2];                 \
    x[M] = cadd(u, v);                               \
    x[M + 2] = twmul<false>(rot8<2 * ((M) & 1), false>(csub(u, v)), w2); \
  }
#define OS_DIF_C(M)                                  \
  {                                                  \
    const cd u = x[M], v = x[M + 1];                 \
    x[M] = cadd(u, v);                               \
    x[M + 1] = twmul<false>(csub(u, v), w4);         \
  }
__device__ __forceinline__ void dif3(cd (&x)[8], const cd w1, const cd w2, const cd w4) {
  OS_DIF_A(0) OS_DIF_A(1) OS_DIF_A(2) OS_DIF_A(3)
  OS_DIF_B(0) OS_DIF_B(1) OS_DIF_B(4) OS_DIF_B(5)
  OS_DIF_C(0) OS_DIF_C(2) OS_DIF_C(4) OS_DIF_C(6)
}
// ... and their inverse (decimation in time, conjugated twiddles, no scaling): the stages in reverse order
#define OS_DIT_C(M)                                  \
  {                                                  \
    const cd a = x[M], b = twmul<true>(x[M + 1], w4); \
    x[M] = cadd(a, b);                               \
    x[M + 1] = csub(a, b);                           \
  }
#define OS_DIT_B(M)                                  \
  {                                                  \
    const cd a = x[M], b = rot8<2 * ((M) & 1), true>(twmul<true>(x[M + 2], w2)); \
    x[M] = cadd(a, b);                               \
    x[M + 2] = csub(a, b);                           \
  }
#define OS_DIT_A(M)                                  \
  {                                                  \
    const cd a = x[M], b = rot8<M, true>(twmul<true>(x[M + 4], w1)); \
    x[M] = cadd(a, b);                               \
    x[M + 4] = csub(a, b);                           \
  }
__device__ __forceinline__ void dit3(cd (&x)[8], const cd w1, const cd w2, const cd w4) {
  OS_DIT_C(0) OS_DIT_C(2) OS_DIT_C(4) OS_DIT_C(6)
  OS_DIT_B(0) OS_DIT_B(1) OS_DIT_B(4) OS_DIT_B(5)
  OS_DIT_A(0) OS_DIT_A(1) OS_DIT_A(2) OS_DIT_A(3)
}

// The twiddles of thread j (0..127) of a transform: pass A (stages 0-2, spacing 128, lo = j), pass B (stages 3-5, spacing 16,
// lo = j & 15), pass C (stages 6-8, spacing 2, lo = j & 1); tw = W_8192^k (nrx_fft.h), W_1024^e = tw[8 e].
struct OsTw {
  cd a1, a2, a4, b1, b2, b4, c1, c2, c4;
};
__device__ __forceinline__ OsTw os_twiddles(const cd* __restrict__ tw, int j) {
  OsTw w;
  w.a1 = tw[8 * j];
  w.a2 = tw[16 * j];
  w.a4 = tw[32 * j];
  const int lb = j & 15;
  w.b1 = tw[64 * lb];
  w.b2 = tw[128 * lb];
  w.b4 = tw[256 * lb];
  const int lc = j & 1;
  w.c1 = tw[512 * lc];
  w.c2 = tw[1024 * lc];
  w.c4 = tw[2048 * lc];
  return w;
}
// LDS element index (unpadded base) of point 0 of thread j in passes B and C, and the padded stride between a thread's points
__device__ __forceinline__ int os_base_b(int j) { return osi((j >> 4) * 128 + (j & 15)); }      // points at +17 m
__device__ __forceinline__ int os_base_c(int j) { return osi((j >> 1) * 16 + (j & 1)); }        // points at +2 m (inside one 16-block)

// ------------------------------------------------------------------------------------------------- path spectra
// spec[p][pos] = (1/1024) * DIF-FFT_1024( taps[p][. - off_p] )[pos]      (positions = decimation-in-frequency output order)
__global__ void __launch_bounds__(128)
td_path_spectra_kernel(const double* __restrict__ taps, const int32_t* __restrict__ tap_off, int flen, const cd* __restrict__ tw,
                       cd* __restrict__ spec) {
  __shared__ __attribute__((aligned(16))) cd buf[OS_ELEMS];
  const int p = blockIdx.x, j = threadIdx.x;
  const OsTw w = os_twiddles(tw, j);
  const int off = tap_off[p];
  cd v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int k = j + 128 * m - off;
    v[m] = (k >= 0 && k < flen) ? cd(taps[(size_t)p * flen + k], 0.0) : cd(0.0, 0.0);
  }
  dif3(v, w.a1, w.a2, w.a4);
#pragma unroll
  for (int m = 0; m < 8; ++m) buf[osi(j) + 136 * m] = v[m];
  __syncthreads();
  const int bb = os_base_b(j);
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = buf[bb + 17 * m];
  dif3(v, w.b1, w.b2, w.b4);
#pragma unroll
  for (int m = 0; m < 8; ++m) buf[bb + 17 * m] = v[m];
  __syncthreads();
  const int bc = os_base_c(j);
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = buf[bc + 2 * m];
  dif3(v, w.c1, w.c2, w.c4);
#pragma unroll
  for (int m = 0; m < 8; ++m) buf[bc + 2 * m] = v[m];
  __syncthreads();
  const double sc = 1.0 / (double)OSN;
#pragma unroll
  for (int k = 0; k < 8; k += 2) {          // stage 9 on positions 8j + k, 8j + k + 1
    const cd u = buf[osi(8 * j) + k], q = buf[osi(8 * j) + k + 1];
    spec[(size_t)p * OSN + 8 * j + k] = cd((u.re + q.re) * sc, (u.im + q.im) * sc);
    spec[(size_t)p * OSN + 8 * j + k + 1] = cd((u.re - q.re) * sc, (u.im - q.im) * sc);
  }
}

// ------------------------------------------------------------------------------------------------- the filter
// LDS: NX transform buffers, then the twiddles of passes B (16 x 3) and C (2 x 3): only pass A's live in registers (the prefetched
// input block needs the others' 24 VGPRs).  During the H set-up the transform buffers hold the set's gains.
constexpr int OS_TWL = 64;

template <int NX>       // Nr = Nt = NX in {1, 2, 4}; 128 * NX threads
__global__ void __launch_bounds__(128 * NX, 2)
apply_td_os_kernel(const cd* __restrict__ x, int ns, const cd* __restrict__ gains1, int n_paths, const cd* __restrict__ spec,
                   const cd* __restrict__ tw, int hist, OsGeom g, cd* __restrict__ y, double* __restrict__ pow_acc, int pow_nfft) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cd* bufs = (cd*)smem;                     // [NX][OS_ELEMS]
  cd* twl = bufs + NX * OS_ELEMS;           // [OS_TWL]
  constexpr int PP = 8 / NX;                // positions a thread owns in the pointwise phase
  constexpr int WAVES = 2 * NX;
  const int tid = threadIdx.x;
  const int f = __builtin_amdgcn_readfirstlane(tid >> 7);      // the transform this thread works on in the passes (whole waves)
  const int j = tid & 127;
  const int set = blockIdx.x, b = blockIdx.y;
  const int s0 = g.start[set], n_end = g.start[set + 1];
  double sr = 0.0, si = 0.0, s2 = 0.0;
  if (s0 < n_end) {
    const cd* xr = x + ((size_t)b * NX + f) * (size_t)ns;
    cd* yr = y + ((size_t)b * NX + f) * (size_t)ns;
    const int V = OSN - hist;
    // input block of the outputs [n0, n0 + V): u[i] = x[n0 - hist + i], thread j takes i = j + 128 m
    auto load8 = [&](int n0, cd (&u)[8]) __attribute__((always_inline)) {
      const int i0 = n0 - hist + j;
      if (n0 - hist >= 0 && n0 - hist + OSN <= ns) {            // (workgroup-uniform) all in range: eight plain loads
#pragma unroll
        for (int m = 0; m < 8; ++m) u[m] = xr[i0 + 128 * m];
      } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int idx = i0 + 128 * m;
          u[m] = (idx >= 0 && idx < ns) ? xr[idx] : cd(0.0, 0.0);
        }
      }
    };
    cd pf[8];                               // the next block's input, in flight while this one is transformed
    load8(s0, pf);
    __builtin_amdgcn_sched_barrier(0);
    const cd wa1 = tw[8 * j], wa2 = tw[16 * j], wa4 = tw[32 * j];
    if (tid < 16) {
      twl[3 * tid] = tw[64 * tid];
      twl[3 * tid + 1] = tw[128 * tid];
      twl[3 * tid + 2] = tw[256 * tid];
    } else if (tid < 18) {
      const int lc = tid - 16;
      twl[48 + 3 * lc] = tw[512 * lc];
      twl[48 + 3 * lc + 1] = tw[1024 * lc];
      twl[48 + 3 * lc + 2] = tw[2048 * lc];
    }
    // ---- H[r][t] at this thread's PP positions:  sum_p g[r][t][p] * C_p.  The set's gains go through LDS (uniform reads), the
    // spectra of path p + 1 are fetched while path p is accumulated.
    cd H[NX][NX][PP];
#pragma unroll
    for (int r = 0; r < NX; ++r)
#pragma unroll
      for (int t = 0; t < NX; ++t)
#pragma unroll
        for (int k = 0; k < PP; ++k) H[r][t][k] = cd(0.0, 0.0);
    {
      const cd* gb = gains1 + ((size_t)b * g.n_sets + set) * NX * NX * n_paths;      // [r][t][p]
      cd* gl = bufs;
      for (int i = tid; i < NX * NX * n_paths; i += 128 * NX) gl[i] = gb[i];
      __syncthreads();
      const cd* sp = spec + PP * tid;
      cd c[PP];
#pragma unroll
      for (int k = 0; k < PP; ++k) c[k] = sp[k];
      const int np = n_paths;
      for (int p = 0; p < np; ++p) {
        cd cn[PP];
        const cd* spn = sp + (size_t)(p + 1 < np ? p + 1 : p) * OSN;
#pragma unroll
        for (int k = 0; k < PP; ++k) cn[k] = spn[k];
#pragma unroll
        for (int r = 0; r < NX; ++r)
#pragma unroll
          for (int t = 0; t < NX; ++t) {
            const cd gg = gl[(r * NX + t) * n_paths + p];
#pragma unroll
            for (int k = 0; k < PP; ++k) {
              H[r][t][k].re = fma(gg.re, c[k].re, fma(-gg.im, c[k].im, H[r][t][k].re));
              H[r][t][k].im = fma(gg.re, c[k].im, fma(gg.im, c[k].re, H[r][t][k].im));
            }
          }
#pragma unroll
        for (int k = 0; k < PP; ++k) c[k] = cn[k];
      }
    }
    cd* buf = bufs + (size_t)f * OS_ELEMS;
    const int ia = osi(j), ib = os_base_b(j), ic = os_base_c(j);
    const int ip = osi(PP * tid);           // pointwise phase: positions PP*tid .. +PP-1 (inside one 16-block) of every transform
    const cd* twb = twl + 3 * (j & 15);
    const cd* twc = twl + 48 + 3 * (j & 1);
    // power sums (Waveform.getRePower, waveform.py:107-117): the nfft samples of the symbol from round(cpLen / 2) on
    const bool want_pow = pow_acc && set < g.n_sets - 1;
    const int poff = (int)rint((double)(n_end - s0 - pow_nfft) * 0.5);
    __syncthreads();                        // (the H set-up has read the set's gains out of the transform buffers)
    for (int n0 = s0; n0 < n_end; n0 += V) {
      cd v[8];
      // ---- forward pass A on the prefetched input
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = pf[m];
      dif3(v, wa1, wa2, wa4);
      __syncthreads();       // (the previous block's inverse pass A has read this transform's buffer)
#pragma unroll
      for (int m = 0; m < 8; ++m) buf[ia + 136 * m] = v[m];
      if (n0 + V < n_end) load8(n0 + V, pf);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      {
        const cd w1 = twb[0], w2 = twb[1], w4 = twb[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[ib + 17 * m];
        dif3(v, w1, w2, w4);
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[ib + 17 * m] = v[m];
      }
      __syncthreads();
      {
        const cd w1 = twc[0], w2 = twc[1], w4 = twc[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[ic + 2 * m];
        dif3(v, w1, w2, w4);
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[ic + 2 * m] = v[m];
      }
      __syncthreads();
      // ---- pointwise: stage 9 forward, Y_r = sum_t H[r][t] X_t, stage 9 inverse; in place (a thread touches its own positions only)
      {
        cd X[NX][PP];
#pragma unroll
        for (int t = 0; t < NX; ++t)
#pragma unroll
          for (int k = 0; k < PP; k += 2) {
            const cd u = bufs[(size_t)t * OS_ELEMS + ip + k], q = bufs[(size_t)t * OS_ELEMS + ip + k + 1];
            X[t][k] = cadd(u, q);
            X[t][k + 1] = csub(u, q);
          }
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          cd Y[PP];
#pragma unroll
          for (int k = 0; k < PP; ++k) {
            double ar = 0.0, ai = 0.0;
#pragma unroll
            for (int t = 0; t < NX; ++t) {
              ar = fma(H[r][t][k].re, X[t][k].re, fma(-H[r][t][k].im, X[t][k].im, ar));
              ai = fma(H[r][t][k].re, X[t][k].im, fma(H[r][t][k].im, X[t][k].re, ai));
            }
            Y[k] = cd(ar, ai);
          }
#pragma unroll
          for (int k = 0; k < PP; k += 2) {
            bufs[(size_t)r * OS_ELEMS + ip + k] = cadd(Y[k], Y[k + 1]);
            bufs[(size_t)r * OS_ELEMS + ip + k + 1] = csub(Y[k], Y[k + 1]);
          }
        }
      }
      __syncthreads();
      // ---- inverse passes C, B (LDS) and A (to global memory)
      {
        const cd w1 = twc[0], w2 = twc[1], w4 = twc[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[ic + 2 * m];
        dit3(v, w1, w2, w4);
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[ic + 2 * m] = v[m];
      }
      __syncthreads();
      {
        const cd w1 = twb[0], w2 = twb[1], w4 = twb[2];
#pragma unroll
        for (int m = 0; m < 8; ++m) v[m] = buf[ib + 17 * m];
        dit3(v, w1, w2, w4);
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[ib + 17 * m] = v[m];
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 8; ++m) v[m] = buf[ia + 136 * m];
      dit3(v, wa1, wa2, wa4);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i = j + 128 * m;          // circular-convolution index: valid from hist on
        const int n = n0 - hist + i;
        if (i >= hist && n < n_end) {
          yr[n] = v[m];
          const int pos = n - s0;
          if (want_pow && pos >= poff && pos < poff + pow_nfft) {
            sr += v[m].re;
            si += v[m].im;
            s2 += v[m].re * v[m].re + v[m].im * v[m].im;
          }
        }
      }
    }
  }
  if (pow_acc) {        // one (sum re, sum im, sum |y|^2) triple per wave, in a fixed order: reproducible, no atomics
    for (int o = 32; o > 0; o >>= 1) {
      sr += __shfl_xor(sr, o, 64);
      si += __shfl_xor(si, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if ((tid & 63) == 0) {
      double* o3 = pow_acc + (((size_t)b * gridDim.x + blockIdx.x) * WAVES + (tid >> 6)) * 3;
      o3[0] = sr;
      o3[1] = si;
      o3[2] = s2;
    }
  }
}

}  // namespace

extern "C" int32_t nrx_td_path_spectra_f64(const double* taps, const int32_t* tap_off, int32_t n_paths, int32_t flen, void* spec,
                                           void* stream) {
  NRX_REQUIRE(taps && tap_off && spec, NRX_E_ARG, "nrx_td_path_spectra: NULL buffer");
  NRX_REQUIRE(n_paths >= 1 && flen >= 1 && flen <= OSN, NRX_E_ARG, "nrx_td_path_spectra: bad sizes");
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_td_path_spectra: FFT twiddle table unavailable");
  hipLaunchKernelGGL(td_path_spectra_kernel, dim3(n_paths), dim3(128), 0, (hipStream_t)stream, taps, tap_off, flen, tw, (cd*)spec);
  NRX_CHECK_LAUNCH("nrx_td_path_spectra");
  return NRX_OK;
}

extern "C" int32_t nrx_apply_td_os_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1, int32_t n_sets,
                                       int32_t n_rx, int32_t n_paths, const void* spec, int32_t hist, const int32_t* set_lens, void* y,
                                       int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part, void* stream) {
  NRX_REQUIRE(x && gains1 && spec && set_lens && y, NRX_E_ARG, "nrx_apply_td_os: NULL buffer");
  NRX_REQUIRE(n_sets >= 1 && n_sets <= 16 && n_paths >= 1 && hist >= 0 && ns > 0 && ns < (1ll << 30) && n_items >= 0, NRX_E_ARG, "nrx_apply_td_os: bad sizes");
  if (!(n_rx == n_tx && (n_rx == 1 || n_rx == 2 || n_rx == 4) && hist <= OSN - 384 && n_paths <= 1024 / n_rx)) {
    ::nrx::set_error("nrx_apply_td_os: built for Nr = Nt in {1, 2, 4} and paths no longer than %d samples (Nr %d, Nt %d, hist %d)", OSN - 384,
                     n_rx, n_tx, hist);
    return NRX_E_UNSUPPORTED;
  }
  NRX_REQUIRE(!pow_acc || (n_part && nfft > 0), NRX_E_ARG, "nrx_apply_td_os: power sums need n_part and nfft");
  if (pow_acc)
    for (int i = 0; i + 1 < n_sets; ++i)
      NRX_REQUIRE(set_lens[i] > nfft, NRX_E_ARG, "nrx_apply_td_os: a symbol (%d samples) is not longer than nfft", set_lens[i]);
  OsGeom g;
  g.n_sets = n_sets;
  int64_t s = 0;
  for (int i = 0; i < n_sets; ++i) {
    g.start[i] = (int32_t)(s < ns ? s : ns);
    s += set_lens[i];
  }
  g.start[n_sets] = (int32_t)ns;            // samples past the listed symbols keep the last gain set (channelmodel.py:443-446)
  const int waves = 2 * n_rx;
  if (pow_acc) {
    const int64_t need = (int64_t)n_items * n_sets * waves * 3;
    NRX_REQUIRE(pow_capacity >= need, NRX_E_SHAPE, "nrx_apply_td_os: pow_acc needs %lld doubles", (long long)need);
    *n_part = n_sets * waves;
  }
  if (n_items == 0) return NRX_OK;
  NRX_REQUIRE(n_items <= 65535, NRX_E_SHAPE, "nrx_apply_td_os: %d items exceed the grid's 65535 (split the batch)", n_items);
  const cd* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_apply_td_os: FFT twiddle table unavailable");
  const size_t lds = sizeof(cd) * ((size_t)n_rx * OS_ELEMS + OS_TWL);      // (the set's Nr x Nt x n_paths gains fit the transform buffers)
  const dim3 grid(n_sets, n_items);
#define NRX_OS_CASE(NX)                                                                                                         \
  case NX:                                                                                                                      \
    (void)hipFuncSetAttribute((const void*)apply_td_os_kernel<NX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    hipLaunchKernelGGL(apply_td_os_kernel<NX>, grid, dim3(128 * NX), lds, (hipStream_t)stream, (const cd*)x, (int)ns, (const cd*)gains1, \
                       n_paths, (const cd*)spec, tw, hist, g, (cd*)y, pow_acc, nfft);                                           \
    break;
  switch (n_rx) {
    NRX_OS_CASE(1)
    NRX_OS_CASE(2)
    NRX_OS_CASE(4)
  }
#undef NRX_OS_CASE
  NRX_CHECK_LAUNCH("nrx_apply_td_os");
  return NRX_OK;
}
