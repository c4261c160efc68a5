// OFDM modulation (IFFT + cyclic prefix + raised-cosine overlap windowing) and demodulation (CP strip + FFT)
// as batched LDS FFTs (gfx950).
//
// Replaces reference grid.py:521-582 (Grid.ofdmModulate), waveform.py:380-470 (applyWindowing "STD"),
// waveform.py:473-527 (Waveform.ofdmDemodulate).  HBM-bound: one read of the grid, one write of the waveform.
#include "nrx_common.h"
#include "nrx_fft.h"
#include "nrx_rng.h"

// Threads of a symbol-parallel modulator workgroup: the 256 the radix-16 passes of a 4096-point transform occupy.  (512 were
// used while the fill kept one load per thread in flight; with the loads of a chunk issued together 256 threads cover the
// latency, and at two waves per SIMD each has the 256 registers that keeping a pass's twiddle loads together needs.)
#define MOD_THREADS 256
// elements per thread whose global loads are issued together in the fill phases
constexpr int FILL_U = 8;


namespace {
using nrx::cx;

// up to 8 consecutive slots of one subframe per call (Grid.ofdmModulate takes several slots, grid.py:546-549)
constexpr int NRX_MAX_SYMS = 112;
struct SymGeom {
  int32_t n_sym;
  int32_t cp[NRX_MAX_SYMS];     // CP length of each symbol of the slot(s) (samples)
  int32_t start[NRX_MAX_SYMS];  // first sample of each symbol (CP included) inside the slot(s)
};

// One workgroup per (batch item, antenna port): the 14 symbols are produced in order because the windowed tail
// of symbol l-1 is overlap-added onto the head of symbol l (waveform.py:436-465).
template <typename T>
__global__ void __launch_bounds__(256)
ofdm_mod_kernel(const cx<T>* __restrict__ grid, int K, int nfft, int log2n, SymGeom g, int w, int slot_len,
                cx<T>* __restrict__ wave, int64_t wave_stride /* samples per (item, port) row */,
                const cx<T>* __restrict__ f, int64_t f_stride, int nl, int ports, const cx<double>* __restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<T>* buf = (cx<T>*)smem;
  cx<T>* tail = buf + nrx::fft_lds_elems((size_t)nfft);   // [w] windowed tail of the previous symbol (behind the padded FFT buffer)
  cx<T>* head0 = tail + w;      // [w] windowed head of symbol 0 (completed by the last symbol's tail)
  const int row = blockIdx.x;   // item * ports + port
  // f != null: the grid holds `nl` layers per item and this row is antenna port `row % ports` of the wideband
  // precoder F (Grid.precode, grid.py:505-516, fused into the load): x_port = sum_n F[port][n] * layer_n
  const int item = f ? row / ports : 0, port = f ? row % ports : 0;
  const cx<T>* src = f ? grid + (size_t)item * nl * g.n_sym * K : grid + (size_t)row * g.n_sym * K;
  cx<double> fw[8];
  if (f)
    for (int n = 0; n < nl; ++n) fw[n] = cx<double>(f[(size_t)item * f_stride + (size_t)port * nl + n]);
  cx<T>* dst = wave + (size_t)row * wave_stride;
  const int pad_lo = (nfft - K + 1) / 2;  // grid.py:543
  const double inv_n = 1.0 / (double)nfft;
  for (int l = 0; l < g.n_sym; ++l) {
    __syncthreads();
    // zero-pad to nfft and ifftshift: buf[i] = padded[(i + nfft/2) mod nfft]
    for (int i = threadIdx.x; i < nfft; i += blockDim.x) {
      const int j = (i + nfft / 2) & (nfft - 1);
      const int k = j - pad_lo;
      cx<T> v(0, 0);
      if (k >= 0 && k < K) {
        if (f) {
          cx<double> acc(0, 0);
          for (int n = 0; n < nl; ++n) nrx::cmac(acc, fw[n], cx<double>(src[((size_t)n * g.n_sym + l) * K + k]));
          v = cx<T>(acc);
        } else {
          v = src[(size_t)l * K + k];
        }
      }
      buf[nrx::fft_idx(i)] = v;
    }
    __syncthreads();
    nrx::fft_dif_lds(buf, tw, nfft, log2n, true);
    const int cp = g.cp[l], n_l = cp + nfft;
    // extended symbol ex[i], i in [0, n_l + w): sample time t = i - w - cp of the periodic extension
    for (int i = threadIdx.x; i < n_l + w; i += blockDim.x) {
      const int t = (i - w - cp + 2 * nfft) & (nfft - 1);
      const cx<T> x = buf[nrx::fft_idx(nrx::fft_bitrev(t, log2n))];
      double win = 1.0;
      if (i < w) win = 0.5 * (1.0 - sinpi((double)(w - 1 - 2 * i) / (double)(2 * w)));
      else if (i >= n_l) win = 0.5 * (1.0 - sinpi((double)(w - 1 - 2 * (n_l + w - 1 - i)) / (double)(2 * w)));
      cx<T> v((T)((double)x.re * inv_n * win), (T)((double)x.im * inv_n * win));
      if (i >= n_l) continue;  // tails are handled below (needs the barrier)
      if (i < w) {
        if (l == 0) { head0[i] = v; continue; }
        v = v + tail[i];
      }
      int pos = g.start[l] + i - w;  // roll(-w) of waveform.py:467
      if (pos < 0) pos += slot_len;
      dst[pos] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < w; q += blockDim.x) {
      const int i = n_l + q;
      const int t = (i - w - cp + 2 * nfft) & (nfft - 1);
      const cx<T> x = buf[nrx::fft_idx(nrx::fft_bitrev(t, log2n))];
      const double win = 0.5 * (1.0 - sinpi((double)(w - 1 - 2 * (w - 1 - q)) / (double)(2 * w)));
      tail[q] = cx<T>((T)((double)x.re * inv_n * win), (T)((double)x.im * inv_n * win));
    }
  }
  __syncthreads();
  // the last symbol's tail wraps onto the slot start (waveform.py:462-465)
  for (int q = threadIdx.x; q < w; q += blockDim.x) {
    int pos = g.start[0] + q - w;
    if (pos < 0) pos += slot_len;
    dst[pos] = head0[q] + tail[q];
  }
}

// No-window variant is the same kernel with w = 0 (tail/head loops vanish).

// One FFT per (item, antenna, symbol); workgroups loop over tasks so the twiddle table is built once.
template <typename T, typename TO = T, bool F64N = false>       // TO: element type of the grid written (float32 transform, float64 grid: _f32o64); F64N: nrx_rng.h
__global__ void __launch_bounds__(256, 2)   // two 68 KB workgroups per CU = two waves per SIMD: 256 registers each
ofdm_demod_kernel(const cx<T>* __restrict__ wave, int64_t wave_stride, int64_t wave_len,
                  const int32_t* __restrict__ t_off, int t_off_stride, int n_ant, int K, int nfft, int log2n, SymGeom g,
                  cx<TO>* __restrict__ grid, int n_tasks, const cx<double>* __restrict__ tw,
                  const T* __restrict__ sigma, int sigma_stride, uint64_t seed, uint64_t stream_id, int64_t batch_offset,
                  const int64_t* __restrict__ item_ids, double cp_offset_ratio) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<T>* buf = (cx<T>*)smem;
  for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const int l = task % g.n_sym;
    const int row = task / g.n_sym;  // item * n_ant + antenna
    const int item = row / n_ant;
    const int64_t ts = t_off ? (int64_t)t_off[(size_t)item * t_off_stride] : 0;  // Waveform.sync (waveform.py:317-341)
    const cx<T>* src = wave + (size_t)row * wave_stride;
    const int cp = g.cp[l];
    const int off = (int)rint((double)cp * cp_offset_ratio);  // np.round(cpLens * cpOffsetRatio), waveform.py:507
    __syncthreads();
    // Fill in two phases per chunk of FILL_U elements per thread: all loads first (in flight together), then the noise and
    // the LDS stores.  (As one loop the compiler kept one load in flight: load, wait, noise, store -- the branch around
    // the noise ends the block the loads could have been hoisted in.)
    const double sg = sigma ? (double)sigma[(size_t)item * sigma_stride] : 0.0;
    const uint64_t nid = sigma ? (uint64_t)(item_ids ? item_ids[item] : batch_offset + item) : 0;
    for (int i0 = threadIdx.x; i0 < nfft; i0 += FILL_U * blockDim.x) {
      cx<T> xs[FILL_U];
#pragma unroll
      for (int u = 0; u < FILL_U; ++u) {
        const int i = i0 + u * blockDim.x;
        const int64_t s = ts + g.start[l] + off + ((cp - off + i) & (nfft - 1));  // waveform.py:509
        xs[u] = (i < nfft && s < wave_len) ? src[s] : cx<T>(0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // sigma != null: the received waveform is noiseless and the AWGN of nrx_awgn_* (same generator, same element
      // numbering: element = antenna * wave_len + sample of the item) is added while loading -- computed here, behind the
      // issue of the loads and in front of their first use
      cx<double> nz[FILL_U];
      if (sigma) {
#pragma unroll
        for (int u = 0; u < FILL_U; ++u) {
          const int i = i0 + u * blockDim.x;
          const int64_t s = ts + g.start[l] + off + ((cp - off + i) & (nfft - 1));
          nz[u] = nrx::awgn_noise<F64N>(sg, seed, stream_id, nid, (int64_t)(row - item * n_ant) * wave_len + s);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < FILL_U; ++u) {
        const int i = i0 + u * blockDim.x;
        const int64_t s = ts + g.start[l] + off + ((cp - off + i) & (nfft - 1));
        cx<T> v = xs[u];
        if (sigma && s < wave_len) v = cx<T>((T)((double)v.re + nz[u].re), (T)((double)v.im + nz[u].im));
        if (i < nfft) buf[nrx::fft_idx(i)] = v;
      }
    }
    __syncthreads();
    nrx::fft_dif_lds(buf, tw, nfft, log2n, false);
    cx<TO>* dst = grid + ((size_t)row * g.n_sym + l) * K;
#pragma unroll 4
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
      const int q = (k - K / 2 + nfft) & (nfft - 1);  // fftshift + centre K bins (waveform.py:514-520)
      const cx<T> v = buf[nrx::fft_idx(nrx::fft_bitrev(q, log2n))];
      dst[k] = cx<TO>((TO)v.re, (TO)v.im);
    }
  }
}

// Symbol-parallel modulator (used by the precoded entry): one workgroup per (item, port, symbol).  The raised-cosine
// overlap couples neighbouring symbols only through w samples: a workgroup writes its own windowed head to its final
// place and parks its windowed tail in `tails`; ofdm_tail_add_kernel then adds tail l onto head l+1 (head + tail, the
// order of the sequential kernel, so both produce identical samples).
template <typename T>
__global__ void __launch_bounds__(MOD_THREADS, 2)
ofdm_mod_sym_kernel(const cx<T>* __restrict__ grid, int K, int nfft, int log2n, SymGeom g, int w, int slot_len,
                    cx<T>* __restrict__ wave, int64_t wave_stride, const cx<T>* __restrict__ f, int64_t f_stride, int nl,
                    int ports, const cx<double>* __restrict__ tw, cx<T>* __restrict__ tails, int xcd_pairs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cx<T>* buf = (cx<T>*)smem;
  // Workgroup -> (item, port, symbol).  With a precoder every port of an (item, symbol) pair reads the same nl layer rows:
  // consecutive workgroups go to the 8 XCDs round-robin (each XCD has its own L2), so the ports of a pair are placed 8
  // workgroups apart -- same XCD, dispatched together -- and the layer rows come from HBM once instead of once per port.
  int l, row;
  if (f && xcd_pairs) {
    const int b = blockIdx.x, x = b & 7, pt = (b >> 3) % ports, q = (b >> 3) / ports;
    const int pair = q * 8 + x;              // (item, symbol) pair; the host pads the grid to whole groups of 8 pairs
    if (pair >= xcd_pairs) return;
    l = pair % g.n_sym;
    row = (pair / g.n_sym) * ports + pt;
  } else {
    l = blockIdx.x % g.n_sym;
    row = blockIdx.x / g.n_sym;             // item * ports + port
  }
  const int item = f ? row / ports : 0, port = f ? row % ports : 0;
  const cx<T>* src = f ? grid + (size_t)item * nl * g.n_sym * K : grid + (size_t)row * g.n_sym * K;
  cx<double> fw[8];
  if (f)
    for (int n = 0; n < nl; ++n) fw[n] = cx<double>(f[(size_t)item * f_stride + (size_t)port * nl + n]);
  cx<T>* dst = wave + (size_t)row * wave_stride;
  const int pad_lo = (nfft - K + 1) / 2;  // grid.py:543
  const double inv_n = 1.0 / (double)nfft;
  // (unrolled by 4: the loads of four iterations are in flight together -- with two 4-wave workgroups per CU a
  //  load-use-load chain of 16 iterations was most of this kernel's time)
  if (f) {
#pragma unroll 4
    for (int i = threadIdx.x; i < nfft; i += blockDim.x) {
      const int j = (i + nfft / 2) & (nfft - 1);
      const int k = j - pad_lo;
      cx<T> v(0, 0);
      if (k >= 0 && k < K) {
        cx<double> acc(0, 0);
        for (int n = 0; n < nl; ++n) nrx::cmac(acc, fw[n], cx<double>(src[((size_t)n * g.n_sym + l) * K + k]));
        v = cx<T>(acc);
      }
      buf[nrx::fft_idx(i)] = v;
    }
  } else {
    // all loads of a chunk first, then the LDS stores (see the demodulator's fill)
    for (int i0 = threadIdx.x; i0 < nfft; i0 += FILL_U * blockDim.x) {
      cx<T> xs[FILL_U];
#pragma unroll
      for (int u = 0; u < FILL_U; ++u) {
        const int i = i0 + u * blockDim.x;
        const int k = ((i + nfft / 2) & (nfft - 1)) - pad_lo;
        xs[u] = (i < nfft && k >= 0 && k < K) ? src[(size_t)l * K + k] : cx<T>(0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < FILL_U; ++u) {
        const int i = i0 + u * blockDim.x;
        if (i < nfft) buf[nrx::fft_idx(i)] = xs[u];
      }
    }
  }
  __syncthreads();
  nrx::fft_dif_lds(buf, tw, nfft, log2n, true);
  const int cp = g.cp[l], n_l = cp + nfft;
#pragma unroll 4
  for (int i = threadIdx.x; i < n_l + w; i += blockDim.x) {
    const int t = (i - w - cp + 2 * nfft) & (nfft - 1);
    const cx<T> x = buf[nrx::fft_idx(nrx::fft_bitrev(t, log2n))];
    double win = 1.0;
    if (i < w) win = 0.5 * (1.0 - sinpi((double)(w - 1 - 2 * i) / (double)(2 * w)));
    else if (i >= n_l) win = 0.5 * (1.0 - sinpi((double)(w - 1 - 2 * (n_l + w - 1 - i)) / (double)(2 * w)));
    const cx<T> v((T)((double)x.re * inv_n * win), (T)((double)x.im * inv_n * win));
    if (i >= n_l) {
      tails[((size_t)row * g.n_sym + l) * w + (i - n_l)] = v;
    } else {
      int pos = g.start[l] + i - w;  // roll(-w) of waveform.py:467
      if (pos < 0) pos += slot_len;
      dst[pos] = v;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
ofdm_tail_add_kernel(cx<T>* __restrict__ wave, int64_t wave_stride, SymGeom g, int w, int slot_len,
                     const cx<T>* __restrict__ tails, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % w);
    const int64_t rl = i / w;
    const int l = (int)(rl % g.n_sym);
    const int64_t row = rl / g.n_sym;
    const int ln = (l + 1) % g.n_sym;      // the last symbol's tail wraps onto the slot start (waveform.py:462-465)
    int pos = g.start[ln] + q - w;
    if (pos < 0) pos += slot_len;
    cx<T>* d = wave + (size_t)row * wave_stride + pos;
    *d = *d + tails[i];
  }
}

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

int32_t fill_geom(const int32_t* cp_lens, int32_t n_sym, int32_t nfft, SymGeom* g, int* slot_len) {
  NRX_REQUIRE(cp_lens && n_sym >= 1 && n_sym <= NRX_MAX_SYMS, NRX_E_ARG, "nrx_ofdm: need 1..%d CP lengths", NRX_MAX_SYMS);
  NRX_REQUIRE(nfft >= 64 && nfft <= 8192 && (nfft & (nfft - 1)) == 0, NRX_E_ARG, "nrx_ofdm: nfft must be a power of two in [64, 8192]");
  g->n_sym = n_sym;
  int s = 0;
  for (int l = 0; l < n_sym; ++l) {
    NRX_REQUIRE(cp_lens[l] >= 0 && cp_lens[l] < nfft, NRX_E_ARG, "nrx_ofdm: CP length out of range");
    g->cp[l] = cp_lens[l];
    g->start[l] = s;
    s += cp_lens[l] + nfft;
  }
  *slot_len = s;
  return NRX_OK;
}

template <typename T>
int32_t mod_entry(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym,
                  int32_t window_len, void* wave, int64_t wave_stride, void* stream, const void* f = nullptr,
                  int64_t f_stride = 0, int32_t nl = 0, int32_t ports = 1, void* tails = nullptr, bool force_sym = false) {
  NRX_REQUIRE(grid && wave, NRX_E_ARG, "nrx_ofdm_modulate: NULL buffer");
  NRX_REQUIRE(!f || (nl >= 1 && nl <= 8 && ports >= 1 && n_rows % ports == 0), NRX_E_ARG,
              "nrx_ofdm_modulate_precoded: bad layer / port counts");
  SymGeom g;
  int slot_len;
  int32_t rc = fill_geom(cp_lens, n_sym, nfft, &g, &slot_len);
  if (rc) return rc;
  NRX_REQUIRE(K > 0 && K <= nfft, NRX_E_SHAPE, "nrx_ofdm_modulate: K (%d) exceeds nfft (%d)", K, nfft);
  NRX_REQUIRE(wave_stride >= slot_len, NRX_E_SHAPE, "nrx_ofdm_modulate: wave_stride < slot length %d", slot_len);
  int wmin = nfft;
  for (int l = 0; l < n_sym; ++l) wmin = g.cp[l] < wmin ? g.cp[l] : wmin;
  NRX_REQUIRE(window_len >= 0 && (window_len == 0 || window_len < wmin), NRX_E_ARG,
              "nrx_ofdm_modulate: The windowing size must be smaller than CP size");
  if (n_rows == 0) return NRX_OK;
  const cx<double>* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_ofdm_modulate: FFT twiddle table unavailable");
  if (tails || window_len == 0 || force_sym) {   // symbol-parallel form
    NRX_REQUIRE(tails || window_len == 0, NRX_E_ARG, "nrx_ofdm_modulate_sym: windowing needs the tails workspace");
    const size_t lds = sizeof(cx<T>) * nrx::fft_lds_elems((size_t)nfft);
    auto kern = ofdm_mod_sym_kernel<T>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // precoded: XCD-aware placement of the ports of an (item, symbol) pair (see the kernel)
    const int pairs = f ? (n_rows / ports) * n_sym : 0;
    const int n_wg = f ? ((pairs + 7) / 8) * 8 * ports : n_rows * n_sym;
    hipLaunchKernelGGL(kern, dim3(n_wg), dim3(MOD_THREADS), lds, (hipStream_t)stream, (const cx<T>*)grid, K, nfft,
                       ilog2(nfft), g, window_len, slot_len, (cx<T>*)wave, wave_stride, (const cx<T>*)f, f_stride, nl,
                       ports, tw, (cx<T>*)tails, pairs);
    if (window_len > 0) {
      const int64_t total = (int64_t)n_rows * n_sym * window_len;
      hipLaunchKernelGGL(ofdm_tail_add_kernel<T>, dim3(nrx::stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                         (cx<T>*)wave, wave_stride, g, window_len, slot_len, (const cx<T>*)tails, total);
    }
    NRX_CHECK_LAUNCH("nrx_ofdm_modulate");
    return NRX_OK;
  }
  const size_t lds = sizeof(cx<T>) * (nrx::fft_lds_elems((size_t)nfft) + 2 * (size_t)window_len);
  auto kern = ofdm_mod_kernel<T>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(n_rows), dim3(256), lds, (hipStream_t)stream, (const cx<T>*)grid, K, nfft, ilog2(nfft), g,
                     window_len, slot_len, (cx<T>*)wave, wave_stride, (const cx<T>*)f, f_stride, nl, ports, tw);
  NRX_CHECK_LAUNCH("nrx_ofdm_modulate");
  return NRX_OK;
}

template <typename T, typename TO = T>
int32_t demod_entry(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride,
                    int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym,
                    void* grid, void* stream, const void* sigma = nullptr, int32_t sigma_stride = 0, uint64_t seed = 0,
                    uint64_t stream_id = 0, int64_t batch_offset = 0, const int64_t* item_ids = nullptr,
                    double cp_offset_ratio = 0.5) {
  NRX_REQUIRE(cp_offset_ratio >= 0.0 && cp_offset_ratio <= 1.0, NRX_E_ARG, "nrx_ofdm_demodulate: cpOffsetRatio must be in [0, 1]");
  NRX_REQUIRE(!sigma || wave_stride == wave_len, NRX_E_SHAPE,
              "nrx_ofdm_demodulate_awgn: rows must be contiguous (element numbering of nrx_awgn)");
  NRX_REQUIRE(wave && grid, NRX_E_ARG, "nrx_ofdm_demodulate: NULL buffer");
  SymGeom g;
  int slot_len;
  int32_t rc = fill_geom(cp_lens, n_sym, nfft, &g, &slot_len);
  if (rc) return rc;
  NRX_REQUIRE(K > 0 && K <= nfft, NRX_E_SHAPE, "nrx_ofdm_demodulate: K (%d) exceeds nfft (%d)", K, nfft);
  NRX_REQUIRE(wave_len >= slot_len && wave_stride >= wave_len, NRX_E_SHAPE,
              "nrx_ofdm_demodulate: waveform shorter than one slot (%d samples)", slot_len);
  const int n_tasks = n_items * n_ant * n_sym;
  if (n_tasks == 0) return NRX_OK;
  const cx<double>* tw = nrx::fft_twiddle_table((hipStream_t)stream);
  NRX_REQUIRE(tw, NRX_E_HIP, "nrx_ofdm_demodulate: FFT twiddle table unavailable");
  const size_t lds = sizeof(cx<T>) * nrx::fft_lds_elems((size_t)nfft);
  auto kern = (sigma && nrx::noise_f64()) ? ofdm_demod_kernel<T, TO, true> : ofdm_demod_kernel<T, TO, false>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid_dim = n_tasks < 4096 ? n_tasks : 4096;
  hipLaunchKernelGGL(kern, dim3(grid_dim), dim3(256), lds, (hipStream_t)stream, (const cx<T>*)wave, wave_stride, wave_len,
                     t_off, t_off_stride, n_ant, K, nfft, ilog2(nfft), g, (cx<TO>*)grid, n_tasks, tw, (const T*)sigma,
                     sigma_stride, seed, stream_id, batch_offset, item_ids, cp_offset_ratio);
  NRX_CHECK_LAUNCH("nrx_ofdm_demodulate");
  return NRX_OK;
}

}  // namespace

extern "C" int32_t nrx_ofdm_demodulate_awgn_f32(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* grid, void* stream) { NRX_REQUIRE(sigma, NRX_E_ARG, "nrx_ofdm_demodulate_awgn: NULL sigma"); return demod_entry<float>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, sigma, sigma_stride, seed, stream_id, batch_offset, item_ids); }
// float32 waveform and transform, complex128 grid out (the float32 waveform chain hands over to the float64 estimator here)
extern "C" int32_t nrx_ofdm_demodulate_awgn_f32o64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* grid, void* stream) { NRX_REQUIRE(sigma, NRX_E_ARG, "nrx_ofdm_demodulate_awgn: NULL sigma"); return demod_entry<float, double>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, sigma, sigma_stride, seed, stream_id, batch_offset, item_ids); }
extern "C" int32_t nrx_ofdm_demodulate_f32o64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream) { return demod_entry<float, double>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, nullptr, 0, 0, 0, 0, nullptr, cp_offset_ratio); }
extern "C" int32_t nrx_ofdm_demodulate_awgn_f64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* grid, void* stream) { NRX_REQUIRE(sigma, NRX_E_ARG, "nrx_ofdm_demodulate_awgn: NULL sigma"); return demod_entry<double>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, sigma, sigma_stride, seed, stream_id, batch_offset, item_ids); }
extern "C" int32_t nrx_ofdm_modulate_f32(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* stream) { return mod_entry<float>(grid, n_rows, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream); }
extern "C" int32_t nrx_ofdm_modulate_f64(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* stream) { return mod_entry<double>(grid, n_rows, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream); }
extern "C" int32_t nrx_ofdm_modulate_sym_f32(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws, void* stream) { return mod_entry<float>(grid, n_rows, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream, nullptr, 0, 0, 1, tails_ws, true); }
extern "C" int32_t nrx_ofdm_modulate_sym_f64(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws, void* stream) { return mod_entry<double>(grid, n_rows, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream, nullptr, 0, 0, 1, tails_ws, true); }
extern "C" int32_t nrx_ofdm_modulate_precoded_f32(const void* layers, int32_t n_items, int32_t n_layers, int32_t n_ports, const void* f, int64_t f_stride, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws, void* stream) { NRX_REQUIRE(f, NRX_E_ARG, "nrx_ofdm_modulate_precoded: NULL precoder"); NRX_REQUIRE(tails_ws || window_len == 0, NRX_E_ARG, "nrx_ofdm_modulate_precoded: windowing needs the tails workspace"); return mod_entry<float>(layers, n_items * n_ports, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream, f, f_stride, n_layers, n_ports, tails_ws); }
extern "C" int32_t nrx_ofdm_modulate_precoded_f64(const void* layers, int32_t n_items, int32_t n_layers, int32_t n_ports, const void* f, int64_t f_stride, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws, void* stream) { NRX_REQUIRE(f, NRX_E_ARG, "nrx_ofdm_modulate_precoded: NULL precoder"); NRX_REQUIRE(tails_ws || window_len == 0, NRX_E_ARG, "nrx_ofdm_modulate_precoded: windowing needs the tails workspace"); return mod_entry<double>(layers, n_items * n_ports, K, nfft, cp_lens, n_sym, window_len, wave, wave_stride, stream, f, f_stride, n_layers, n_ports, tails_ws); }
extern "C" int32_t nrx_ofdm_demodulate_f32(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream) { return demod_entry<float>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, nullptr, 0, 0, 0, 0, nullptr, cp_offset_ratio); }
extern "C" int32_t nrx_ofdm_demodulate_f64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off, int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft, const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream) { return demod_entry<double>(wave, wave_stride, wave_len, t_off, t_off_stride, n_items, n_ant, K, nfft, cp_lens, n_sym, grid, stream, nullptr, 0, 0, 0, 0, nullptr, cp_offset_ratio); }

