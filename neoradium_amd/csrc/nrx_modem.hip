// QAM mapping / max-log LLR demapping with PDSCH scrambling and layer/RE (de)mapping fused (gfx950).
//
// Replaces reference modulation.py:127-156 (Modem.modulate), :159-204 (getLLRsFromSymbols),
// pdsch.py:603-616 (scrambleBits/scrambleLLRs), :619-639 + :855-932 (layer map + populateGrid scatter),
// :935-1005 (getLLRsFromGrid gather).  Streaming kernels: one thread per modulation symbol; arithmetic is
// float64 internally whatever the storage type (an LLR is a difference of near-equal squared distances).
#include "nrx_common.h"
#include "nrx_cplx.h"

namespace {
using nrx::cx;

// TS 38.211 5.1.2-5.1.7 amplitude of one axis from its bits a[0] (sign), a[1..h-1] (MSB-first amplitude bits),
// un-normalised odd integer; the recursion of reference modulation.py:60-74.
__device__ __forceinline__ int pam_level(uint32_t axis_bits, int h) {
  // axis_bits: bit (h-1-i) holds a[i]
  int v = 1;
  for (int i = h - 1; i >= 1; --i) {  // innermost bit first
    const int b = (axis_bits >> (h - 1 - i)) & 1;
    v = (1 << (h - i)) - (1 - 2 * b) * v;
  }
  const int s = (axis_bits >> (h - 1)) & 1;
  return (1 - 2 * s) * v;
}

// split the qm bits of a symbol (MSB first: b0 b1 b2 ...) into real-axis bits (even positions) and imag-axis bits
__device__ __forceinline__ void split_axes(uint32_t v, int qm, uint32_t& re_bits, uint32_t& im_bits) {
  re_bits = im_bits = 0;
  const int h = qm / 2;
  for (int i = 0; i < h; ++i) {
    re_bits = (re_bits << 1) | ((v >> (qm - 1 - 2 * i)) & 1);
    im_bits = (im_bits << 1) | ((v >> (qm - 2 - 2 * i)) & 1);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
qam_map_kernel(const uint8_t* __restrict__ bits, int64_t bits_stride, const uint8_t* __restrict__ scr, int qm,
               double scale, const int32_t* __restrict__ re_index, int n_sym, cx<T>* __restrict__ out,
               int64_t out_stride, int n_batch) {
  // grid: (x-blocks over the symbols, batch items): 32-bit indices, no 64-bit division per element
  for (int b = blockIdx.y; b < n_batch; b += gridDim.y)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sym; i += gridDim.x * blockDim.x) {
    const uint8_t* src = bits + (size_t)b * bits_stride + (size_t)i * qm;
    uint32_t v = 0;
    for (int q = 0; q < qm; ++q) {
      uint32_t bit = src[q] & 1;
      if (scr) bit ^= scr[(size_t)i * qm + q] & 1;  // pdsch.py:603-608
      v = (v << 1) | bit;
    }
    double re, im;
    if (qm == 1) {
      re = im = (double)(1 - 2 * (int)v) * scale;
    } else {
      uint32_t rb, ib;
      split_axes(v, qm, rb, ib);
      re = (double)pam_level(rb, qm / 2) * scale;
      im = (double)pam_level(ib, qm / 2) * scale;
    }
    const int64_t dst = re_index ? (int64_t)re_index[i] : (int64_t)i;
    out[(size_t)b * out_stride + dst] = cx<T>((T)re, (T)im);
  }
}

// PDSCH.getGrid + populateGrid in one pass (pdsch.py:670-695, 855-932): every grid element is written exactly once --
// a data RE gets its scrambled + modulated symbol (gather through the INVERSE layer/RE map: element -> symbol number),
// everything else (DMRS, empty REs) is copied from the slot-number-in-frame template selected per batch item.  Replaces
// "copy the template, then scatter the symbols" (one 16-byte write per element instead of write + read-modify-write).
template <typename T>
__global__ void __launch_bounds__(256)
pdsch_populate_kernel(const uint8_t* __restrict__ bits, int64_t bits_stride, const uint8_t* __restrict__ scr, int qm,
                      double scale, const int32_t* __restrict__ re_inv, const cx<T>* __restrict__ templ,
                      const int64_t* __restrict__ templ_sel, int64_t elems, cx<T>* __restrict__ out, int n_batch) {
  // grid: (x-blocks, n_batch): no 64-bit division per element
  const int b = blockIdx.y;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)elems; e += gridDim.x * blockDim.x) {
    const int i = re_inv[e];
    cx<T> o;
    if (i < 0) {
      o = templ[(size_t)templ_sel[b] * elems + e];
    } else {
      const uint8_t* src = bits + (size_t)b * bits_stride + (size_t)i * qm;
      uint32_t v = 0;
      for (int q = 0; q < qm; ++q) {
        uint32_t bit = src[q] & 1;
        if (scr) bit ^= scr[(size_t)i * qm + q] & 1;  // pdsch.py:603-608
        v = (v << 1) | bit;
      }
      double re, im;
      if (qm == 1) {
        re = im = (double)(1 - 2 * (int)v) * scale;
      } else {
        uint32_t rb, ib;
        split_axes(v, qm, rb, ib);
        re = (double)pam_level(rb, qm / 2) * scale;
        im = (double)pam_level(ib, qm / 2) * scale;
      }
      o = cx<T>((T)re, (T)im);
    }
    out[(size_t)b * elems + e] = o;
  }
}

// The same with the modulation order as a template parameter (even orders): the qm byte-per-bit values of a symbol and of
// its scrambling bits are fetched as qm/2 16-bit words (bit 2k = real-axis bit in the low byte, bit 2k+1 = imaginary-axis
// bit in the high byte), which halves the number of loads per element; needs 2-byte aligned rows (checked by the host).
template <typename T, int QM>
__global__ void __launch_bounds__(256)
pdsch_populate_q_kernel(const uint8_t* __restrict__ bits, int64_t bits_stride, const uint8_t* __restrict__ scr, double scale,
                        const int32_t* __restrict__ re_inv, const cx<T>* __restrict__ templ,
                        const int64_t* __restrict__ templ_sel, int64_t elems, cx<T>* __restrict__ out, int n_batch) {
  constexpr int h = QM / 2;
  const int b = blockIdx.y;
  const cx<T>* tb = templ + (size_t)templ_sel[b] * elems;
  const uint8_t* bb = bits + (size_t)b * bits_stride;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)elems; e += gridDim.x * blockDim.x) {
    const int i = re_inv[e];
    cx<T> o;
    if (i < 0) {
      o = tb[e];
    } else {
      const uint16_t* src = (const uint16_t*)(bb + (size_t)i * QM);
      uint32_t rb = 0, ib = 0;
#pragma unroll
      for (int k = 0; k < h; ++k) {
        uint32_t w = src[k];
        if (scr) w ^= ((const uint16_t*)(scr + (size_t)i * QM))[k];   // pdsch.py:603-608
        rb = (rb << 1) | (w & 1u);
        ib = (ib << 1) | ((w >> 8) & 1u);
      }
      o = cx<T>((T)((double)pam_level(rb, h) * scale), (T)((double)pam_level(ib, h) * scale));
    }
    out[(size_t)b * elems + e] = o;
  }
}

// ... and with the layer structure of the map known (TS 38.211 7.3.1.3: symbol i goes to layer i mod P, so the P planes of
// the grid hold symbols i, i+1, .., i+P-1 at the same (symbol, subcarrier)): one thread per RE position fetches the
// P*QM contiguous bit bytes once (aligned 16-bit words, neighbouring threads read neighbouring bytes) and writes its P
// planes -- the per-element form gathers QM bytes out of every P*QM.  `planes` is checked against the map by the caller.
template <typename T, int QM, int PL>
__global__ void __launch_bounds__(256)
pdsch_populate_qp_kernel(const uint8_t* __restrict__ bits, int64_t bits_stride, const uint8_t* __restrict__ scr, double scale,
                         const int32_t* __restrict__ re_inv, const cx<T>* __restrict__ templ,
                         const int64_t* __restrict__ templ_sel, int64_t elems, cx<T>* __restrict__ out, int n_batch) {
  constexpr int h = QM / 2;
  const int b = blockIdx.y;
  const int lk = (int)(elems / PL);                      // elements of one plane
  const cx<T>* tb = templ + (size_t)templ_sel[b] * elems;
  const uint8_t* bb = bits + (size_t)b * bits_stride;
  cx<T>* ob = out + (size_t)b * elems;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < lk; e += gridDim.x * blockDim.x) {
    const int i = re_inv[e];
    if (i < 0) {
#pragma unroll
      for (int p = 0; p < PL; ++p) ob[(size_t)p * lk + e] = tb[(size_t)p * lk + e];
      continue;
    }
    const uint16_t* src = (const uint16_t*)(bb + (size_t)i * QM);
    const uint16_t* sc = scr ? (const uint16_t*)(scr + (size_t)i * QM) : nullptr;
    uint32_t w[PL * h];
#pragma unroll
    for (int k = 0; k < PL * h; ++k) w[k] = src[k];
    if (sc) {
#pragma unroll
      for (int k = 0; k < PL * h; ++k) w[k] ^= sc[k];     // pdsch.py:603-608
    }
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      uint32_t rb = 0, ib = 0;
#pragma unroll
      for (int k = 0; k < h; ++k) {
        rb = (rb << 1) | (w[p * h + k] & 1u);
        ib = (ib << 1) | ((w[p * h + k] >> 8) & 1u);
      }
      ob[(size_t)p * lk + e] = cx<T>((T)((double)pam_level(rb, h) * scale), (T)((double)pam_level(ib, h) * scale));
    }
  }
}

// Max-log LLRs (useMax=True, the reference default).  The exhaustive max over the 2^qm points of
// -|y-s|^2/s2 separates per axis for square QAM: bits on the real axis only see (Re y - a)^2 because the
// imaginary-axis minimum is common to both hypotheses and cancels in the difference.
// QM is a template parameter: the per-axis minima live in registers and the level loop unrolls.  The float-output
// (throughput) instantiation multiplies by 1/s2 instead of dividing twice per bit; the double-output one keeps the
// reference's divisions so that it stays bit-identical to NumPy.
// Code-block de-interleaving done by the demapper's stores (nrx_qam_demap_cb_*): E_r = e_small for the first n_small blocks,
// e_small + f for the others (ldpc.py:846-856); e_small = 0: plain symbol-major output.
// pitch > 0 (nrx_qam_demap_rr_*): block r starts at r * pitch instead of offset_r and positions >= sys_len move up by the F
// filler positions -- the rate-recovered layout of a first transmission (rv 0, no repetition: ldpc.py:1330-1418 with k0 = 0).
struct DeintGeom {
  int e_small, n_small, f;
  int pitch, sys_len, F;
};

template <typename T, typename TL, int QM, bool RR = false>
__global__ void __launch_bounds__(256)
qam_demap_kernel(const cx<T>* __restrict__ syms, int64_t sym_stride, const T* __restrict__ scales,
                 const T* __restrict__ noise_var, int nv_stride, const uint8_t* __restrict__ scr, double scale,
                 const int32_t* __restrict__ re_index, int n_sym, TL* __restrict__ llr, int64_t llr_stride,
                 int n_batch, double nv_floor, DeintGeom dg) {
  constexpr int h = QM / 2;
  constexpr bool RECIP = sizeof(TL) == 4;
  double lev[h > 0 ? (1 << h) : 1];
#pragma unroll
  for (int a = 0; a < (1 << h); ++a) lev[a] = (double)pam_level(a, h) * scale;
  // grid: (x-blocks over the symbols, batch items): 32-bit indices, no 64-bit division per element.
  // Every load of a symbol is issued before its arithmetic starts, the RE index of the thread's NEXT symbol with them: left
  // as one expression per use the compiler sank each load to its use -- index, wait, symbol, wait, and one scrambling byte
  // with a wait behind each of the QM division chains: QM + 2 dependent round trips per symbol.
  const int stride = gridDim.x * blockDim.x;
  for (int b = blockIdx.y; b < n_batch; b += gridDim.y) {
    double nv = (double)noise_var[(size_t)b * nv_stride];
    nv = nv > nv_floor ? nv : nv_floor;  // pdsch.py:966 max(noiseVar, 1e-10)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int64_t src = i < n_sym ? (re_index ? (int64_t)re_index[i] : (int64_t)i) : 0;
    for (; i < n_sym; i += stride) {
      const cx<T> y = syms[(size_t)b * sym_stride + src];
      const double sc = scales ? (double)scales[(size_t)b * sym_stride + src] : 1.0;
      uint32_t sbits = 0;                 // bit `pos` = scrambling bit of LLR `pos` of this symbol (pdsch.py:611-616)
      if (scr) {
        if constexpr (QM == 1) {
          sbits = scr[i] & 1u;
        } else {
          const uint16_t* sp = (const uint16_t*)(scr + (size_t)i * QM);     // QM even: 16-bit aligned (checked by the host)
#pragma unroll
          for (int k = 0; k < h; ++k) {
            const uint32_t w = sp[k];
            sbits |= ((w & 1u) << (2 * k)) | (((w >> 8) & 1u) << (2 * k + 1));
          }
        }
      }
      const int inext = i + stride;
      const int64_t src_next = inext < n_sym ? (re_index ? (int64_t)re_index[inext] : (int64_t)inext) : 0;
      __builtin_amdgcn_sched_barrier(0);
      const double rn = RECIP ? sc / nv : 0.0;
      // LLR q of symbol i goes to i*QM + q (the reference's order), or -- code-block option -- to its de-interleaved place
      // inside its code block: position q*(E_r/QM) + s of block r (ldpc.py:1390-1397 done by the store; stride dq between the
      // QM values of a symbol, consecutive symbols = consecutive lanes = consecutive addresses)
      TL* dst = llr + (size_t)b * llr_stride + (size_t)i * QM;
      int dq = 1;
      int fill_from = 0x7fffffff;
      if (dg.e_small > 0) {
        const int ss = dg.e_small / QM, split = dg.n_small * ss;
        int r, sidx, off;
        if (i < split) { r = i / ss; sidx = i - r * ss; dq = ss; off = r * dg.e_small; }
        else { dq = (dg.e_small + dg.f) / QM; r = (i - split) / dq; sidx = (i - split) - r * dq; off = dg.n_small * dg.e_small + r * (dg.e_small + dg.f); }
        dst = llr + (size_t)b * llr_stride + off + sidx;
        if constexpr (RR) {
          const int rg = i < split ? r : dg.n_small + r;
          dst = llr + (size_t)b * llr_stride + (size_t)rg * dg.pitch + sidx;
          fill_from = dg.sys_len - sidx;      // value q sits at position sidx + q*dq: behind the fillers iff q*dq >= fill_from
        }
      }
      if constexpr (QM == 1) {
        const double d0 = ((double)y.re - scale) * ((double)y.re - scale) + ((double)y.im - scale) * ((double)y.im - scale);
        const double d1 = ((double)y.re + scale) * ((double)y.re + scale) + ((double)y.im + scale) * ((double)y.im + scale);
        double l = (-d0 / nv) - (-d1 / nv);
        if (scr) l *= (double)(1 - 2 * (int)(sbits & 1u));
        dst[RR && 0 >= fill_from ? dg.F : 0] = (TL)(l * sc);
      } else {
        TL out[QM];
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
          const double yv = axis == 0 ? (double)y.re : (double)y.im;
          double m0[h], m1[h];
#pragma unroll
          for (int q = 0; q < h; ++q) m0[q] = m1[q] = 1e300;
#pragma unroll
          for (int a = 0; a < (1 << h); ++a) {
            const double d = yv - lev[a];
            const double d2 = d * d;
#pragma unroll
            for (int q = 0; q < h; ++q) {
              if ((a >> (h - 1 - q)) & 1) m1[q] = d2 < m1[q] ? d2 : m1[q];
              else m0[q] = d2 < m0[q] ? d2 : m0[q];
            }
          }
#pragma unroll
          for (int q = 0; q < h; ++q) {
            const int pos = 2 * q + axis;  // bit index inside the symbol
            double l;
            if constexpr (RECIP) {
              l = (m1[q] - m0[q]) * rn;                        // = ((-m0/nv) - (-m1/nv)) * sc up to rounding
              if (scr) l = ((sbits >> pos) & 1u) ? -l : l;
            } else {
              l = (-m0[q] / nv) - (-m1[q] / nv);               // modulation.py:200-202, positive = bit 0
              if (scr) l *= (double)(1 - 2 * (int)((sbits >> pos) & 1u));  // pdsch.py:611-616
              l = l * sc;                                                    // pdsch.py:1002-1003
            }
            out[pos] = (TL)l;
          }
        }
#pragma unroll
        for (int q = 0; q < QM; ++q) dst[(size_t)q * dq + (RR && q * dq >= fill_from ? dg.F : 0)] = out[q];
      }
      src = src_next;
    }
  }
}

// Exact log-likelihood ratios (useMax=False, modulation.py:198-201): exhaustive log-sum-exp with the
// reference's +-700 exponent clip.
template <typename T, typename TL>
__global__ void __launch_bounds__(256)
qam_demap_exact_kernel(const cx<T>* __restrict__ syms, int64_t sym_stride, const T* __restrict__ scales,
                       const T* __restrict__ noise_var, int nv_stride, const uint8_t* __restrict__ scr, int qm,
                       double scale, const int32_t* __restrict__ re_index, int n_sym, TL* __restrict__ llr,
                       int64_t llr_stride, int n_batch, double nv_floor) {
  // grid: (x-blocks over the symbols, batch items): 32-bit indices, no 64-bit division per element
  for (int b = blockIdx.y; b < n_batch; b += gridDim.y)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_sym; i += gridDim.x * blockDim.x) {
    const int64_t src = re_index ? (int64_t)re_index[i] : (int64_t)i;
    const cx<T> y = syms[(size_t)b * sym_stride + src];
    double nv = (double)noise_var[(size_t)b * nv_stride];
    nv = nv > nv_floor ? nv : nv_floor;
    const double sc = scales ? (double)scales[(size_t)b * sym_stride + src] : 1.0;
    double s0[10], s1[10];
    for (int q = 0; q < qm; ++q) s0[q] = s1[q] = 0.0;
    for (uint32_t v = 0; v < (1u << qm); ++v) {
      double re, im;
      if (qm == 1) re = im = (double)(1 - 2 * (int)v) * scale;
      else {
        uint32_t rb, ib;
        split_axes(v, qm, rb, ib);
        re = (double)pam_level(rb, qm / 2) * scale;
        im = (double)pam_level(ib, qm / 2) * scale;
      }
      const double dr = (double)y.re - re, di = (double)y.im - im;
      double e = -(dr * dr + di * di) / nv;
      e = e < -700.0 ? -700.0 : (e > 700.0 ? 700.0 : e);
      const double w = exp(e);
      for (int q = 0; q < qm; ++q) {
        if ((v >> (qm - 1 - q)) & 1) s1[q] += w;
        else s0[q] += w;
      }
    }
    TL* dst = llr + (size_t)b * llr_stride + (size_t)i * qm;
    for (int q = 0; q < qm; ++q) {
      double l = log(s0[q]) - log(s1[q]);
      if (scr) l *= (double)(1 - 2 * (int)(scr[(size_t)i * qm + q] & 1));
      dst[q] = (TL)(l * sc);
    }
  }
}

double qam_scale(int qm) {
  static const int norm[11] = {0, 2, 2, 0, 10, 0, 42, 0, 170, 0, 682};  // modulation.py:58
  return 1.0 / sqrt((double)norm[qm]);
}
// launch shape of the per-symbol kernels: x = blocks over the symbols of an item, y = batch items
dim3 sym_batch_grid(int n_sym, int n_batch) {
  int gx = (n_sym + 255) / 256;
  if (n_batch >= 64 && gx > 64) gx = 64;
  return dim3(gx < 1 ? 1 : gx, n_batch < 65535 ? (n_batch < 1 ? 1 : n_batch) : 65535);
}
bool qm_ok(int qm) { return qm == 1 || qm == 2 || qm == 4 || qm == 6 || qm == 8 || qm == 10; }

template <typename T>
int32_t map_entry(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index,
                  int32_t n_sym, void* out, int64_t out_stride, int32_t n_batch, void* stream) {
  NRX_REQUIRE(bits && out, NRX_E_ARG, "nrx_qam_map: NULL buffer");
  NRX_REQUIRE(qm_ok(qm), NRX_E_ARG, "nrx_qam_map: unsupported modulation order %d", qm);
  NRX_REQUIRE(n_sym >= 0 && n_batch >= 0 && bits_stride >= (int64_t)n_sym * qm, NRX_E_SHAPE, "nrx_qam_map: bad sizes");
  if (n_sym == 0 || n_batch == 0) return NRX_OK;
  hipLaunchKernelGGL(qam_map_kernel<T>, sym_batch_grid(n_sym, n_batch), dim3(256), 0,
                     (hipStream_t)stream, bits, bits_stride, scr, qm, qam_scale(qm), re_index, n_sym, (cx<T>*)out,
                     out_stride, n_batch);
  NRX_CHECK_LAUNCH("nrx_qam_map");
  return NRX_OK;
}

template <typename T>
int32_t populate_entry(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_inv,
                       const void* templ, const int64_t* templ_sel, int64_t elems, void* out, int32_t n_batch, int32_t planes,
                       void* stream) {
  NRX_REQUIRE(bits && re_inv && templ && templ_sel && out, NRX_E_ARG, "nrx_pdsch_populate: NULL buffer");
  NRX_REQUIRE(qm_ok(qm), NRX_E_ARG, "nrx_pdsch_populate: unsupported modulation order %d", qm);
  NRX_REQUIRE(elems >= 0 && elems < (1ll << 31) && n_batch >= 0 && n_batch < 65536 && bits_stride >= 0, NRX_E_SHAPE, "nrx_pdsch_populate: bad sizes");
  if (elems == 0 || n_batch == 0) return NRX_OK;
  int gx = (int)((elems + 255) / 256);
  if (gx > 64) gx = 64;
  const bool aligned = (((uintptr_t)bits | (uintptr_t)scr | (uintptr_t)bits_stride) & 1) == 0;
  NRX_REQUIRE(planes >= 0 && (planes <= 1 || elems % planes == 0), NRX_E_ARG, "nrx_pdsch_populate: %d planes do not divide the grid", planes);
  if (aligned && qm == 6 && (planes == 2 || planes == 4)) {     // the layer-vector form (built where it is used: 64-QAM)
    const int lk = (int)(elems / planes);
    int gp = (lk + 255) / 256;
    if (gp > 64) gp = 64;
    if (planes == 4)
      hipLaunchKernelGGL((pdsch_populate_qp_kernel<T, 6, 4>), dim3(gp, n_batch), dim3(256), 0, (hipStream_t)stream, bits,
                         bits_stride, scr, qam_scale(6), re_inv, (const cx<T>*)templ, templ_sel, elems, (cx<T>*)out, n_batch);
    else
      hipLaunchKernelGGL((pdsch_populate_qp_kernel<T, 6, 2>), dim3(gp, n_batch), dim3(256), 0, (hipStream_t)stream, bits,
                         bits_stride, scr, qam_scale(6), re_inv, (const cx<T>*)templ, templ_sel, elems, (cx<T>*)out, n_batch);
    NRX_CHECK_LAUNCH("nrx_pdsch_populate");
    return NRX_OK;
  }
#define NRX_POP_CASE(Q)                                                                                                    \
  case Q:                                                                                                                  \
    hipLaunchKernelGGL((pdsch_populate_q_kernel<T, Q>), dim3(gx, n_batch), dim3(256), 0, (hipStream_t)stream, bits,        \
                       bits_stride, scr, qam_scale(Q), re_inv, (const cx<T>*)templ, templ_sel, elems, (cx<T>*)out, n_batch); \
    break;
  switch (aligned ? qm : 0) {
    NRX_POP_CASE(2)
    NRX_POP_CASE(4)
    NRX_POP_CASE(6)
    NRX_POP_CASE(8)
    NRX_POP_CASE(10)
    default:
      hipLaunchKernelGGL(pdsch_populate_kernel<T>, dim3(gx, n_batch), dim3(256), 0, (hipStream_t)stream, bits, bits_stride,
                         scr, qm, qam_scale(qm), re_inv, (const cx<T>*)templ, templ_sel, elems, (cx<T>*)out, n_batch);
  }
#undef NRX_POP_CASE
  NRX_CHECK_LAUNCH("nrx_pdsch_populate");
  return NRX_OK;
}

// nrx_qam_demap_rr_*: what the demapper's stores did not reach of the first n_fill code-word positions of every block --
// zeros behind the E_r transmitted positions, LARGE_LLR on the fillers (ldpc.py:1401-1418).
template <typename TL>
__global__ void __launch_bounds__(256)
rr_tail_kernel(TL* __restrict__ llr, int64_t llr_stride, DeintGeom dg, int C, int n_fill) {
  const int b = blockIdx.x / C, r = blockIdx.x - b * C;
  const int E = r < dg.n_small ? dg.e_small : dg.e_small + dg.f;
  const int nE = E < dg.sys_len ? E : E + dg.F;
  TL* dst = llr + (size_t)b * llr_stride + (size_t)r * dg.pitch;
  for (int n = nE + (int)threadIdx.x; n < n_fill; n += blockDim.x) dst[n] = (n >= dg.sys_len && n < dg.sys_len + dg.F) ? (TL)1e20 : (TL)0;
  if (nE > dg.sys_len)
    for (int n = dg.sys_len + (int)threadIdx.x; n < dg.sys_len + dg.F && n < n_fill; n += blockDim.x) dst[n] = (TL)1e20;
}

template <typename T, typename TL>
int32_t demap_entry(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var, int32_t nv_stride,
                    const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym, void* llr,
                    int64_t llr_stride, int32_t n_batch, int32_t exact, double nv_floor, void* stream,
                    int32_t n_code_blocks = 0, int32_t n_layers = 0, const nrx_ldpc_cfg* rr = nullptr, int32_t rr_cols = 0) {
  NRX_REQUIRE(syms && noise_var && llr, NRX_E_ARG, "nrx_qam_demap: NULL buffer");
  DeintGeom dg{0, 0, 0, 0, 0, 0};
  if (rr) {
    dg.pitch = rr->N;
    dg.sys_len = rr->K - 2 * rr->Zc - rr->F;
    dg.F = rr->F;
  }
  if (n_code_blocks > 0) {      // per-code-block de-interleaved output (E_r split as nrx_ldpc_cb_lens, ldpc.py:846-856)
    NRX_REQUIRE(!exact && n_layers >= 1, NRX_E_UNSUPPORTED, "nrx_qam_demap_cb: max-log LLRs only, n_layers >= 1");
    const int f = n_layers * qm, G = n_sym * qm, gb = (G + f - 1) / f;
    NRX_REQUIRE(G % f == 0 && gb / n_code_blocks > 0, NRX_E_SHAPE, "nrx_qam_demap_cb: G=%d is not %d code blocks of multiples of %d bits", G, n_code_blocks, f);
    dg.e_small = (gb / n_code_blocks) * f;
    dg.n_small = n_code_blocks - gb % n_code_blocks;
    dg.f = f;
  }
  NRX_REQUIRE(qm_ok(qm), NRX_E_ARG, "nrx_qam_demap: unsupported modulation order %d", qm);
  NRX_REQUIRE(n_sym >= 0 && n_batch >= 0 && llr_stride >= (rr ? (int64_t)rr->C * rr->N : (int64_t)n_sym * qm), NRX_E_SHAPE, "nrx_qam_demap: bad sizes");
  NRX_REQUIRE(!scr || qm == 1 || ((uintptr_t)scr & 1u) == 0, NRX_E_ARG, "nrx_qam_demap: the scrambling sequence must be 2-byte aligned");
  if (n_sym == 0 || n_batch == 0) return NRX_OK;
  const dim3 grid = sym_batch_grid(n_sym, n_batch);
  if (exact)
    hipLaunchKernelGGL((qam_demap_exact_kernel<T, TL>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)syms,
                       sym_stride, (const T*)scales, (const T*)noise_var, nv_stride, scr, qm, qam_scale(qm), re_index,
                       n_sym, (TL*)llr, llr_stride, n_batch, nv_floor);
  else {
#define NRX_DEMAP_CASE(Q)                                                                                              \
  case Q:                                                                                                              \
    if (rr)                                                                                                            \
      hipLaunchKernelGGL((qam_demap_kernel<T, TL, Q, true>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)syms, \
                         sym_stride, (const T*)scales, (const T*)noise_var, nv_stride, scr, qam_scale(Q), re_index,    \
                         n_sym, (TL*)llr, llr_stride, n_batch, nv_floor, dg);                                          \
    else                                                                                                               \
      hipLaunchKernelGGL((qam_demap_kernel<T, TL, Q>), grid, dim3(256), 0, (hipStream_t)stream, (const cx<T>*)syms,    \
                         sym_stride, (const T*)scales, (const T*)noise_var, nv_stride, scr, qam_scale(Q), re_index,    \
                         n_sym, (TL*)llr, llr_stride, n_batch, nv_floor, dg);                                          \
    break;
    switch (qm) {
      NRX_DEMAP_CASE(1)
      NRX_DEMAP_CASE(2)
      NRX_DEMAP_CASE(4)
      NRX_DEMAP_CASE(6)
      NRX_DEMAP_CASE(8)
      NRX_DEMAP_CASE(10)
      default: NRX_REQUIRE(false, NRX_E_ARG, "nrx_qam_demap: unsupported qm=%d", qm);
    }
#undef NRX_DEMAP_CASE
  }
  NRX_CHECK_LAUNCH("nrx_qam_demap");
  if (rr) {
    hipLaunchKernelGGL(rr_tail_kernel<TL>, dim3(n_batch * rr->C), dim3(256), 0, (hipStream_t)stream, (TL*)llr, llr_stride, dg, rr->C,
                       rr_cols * rr->Zc);
    NRX_CHECK_LAUNCH("nrx_qam_demap_rr");
  }
  return NRX_OK;
}

}  // namespace

extern "C" int32_t nrx_qam_map_f32(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm,
                                   const int32_t* re_index, int32_t n_sym, void* out, int64_t out_stride,
                                   int32_t n_batch, void* stream) {
  return map_entry<float>(bits, bits_stride, scr, qm, re_index, n_sym, out, out_stride, n_batch, stream);
}
extern "C" int32_t nrx_qam_map_f64(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm,
                                   const int32_t* re_index, int32_t n_sym, void* out, int64_t out_stride,
                                   int32_t n_batch, void* stream) {
  return map_entry<double>(bits, bits_stride, scr, qm, re_index, n_sym, out, out_stride, n_batch, stream);
}
#define NRX_DEMAP(NAME, T, TL)                                                                                     \
  extern "C" int32_t NAME(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,         \
                          int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym, \
                          void* llr, int64_t llr_stride, int32_t n_batch, int32_t exact, double nv_floor,          \
                          void* stream) {                                                                          \
    return demap_entry<T, TL>(syms, sym_stride, scales, noise_var, nv_stride, scr, qm, re_index, n_sym, llr,       \
                              llr_stride, n_batch, exact, nv_floor, stream);                                       \
  }
NRX_DEMAP(nrx_qam_demap_f32, float, float)
NRX_DEMAP(nrx_qam_demap_f64, double, double)
NRX_DEMAP(nrx_qam_demap_f64o32, double, float)
#define NRX_DEMAP_CB(NAME, T, TL)                                                                                  \
  extern "C" int32_t NAME(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,         \
                          int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym, \
                          int32_t n_code_blocks, int32_t n_layers, void* llr, int64_t llr_stride, int32_t n_batch, \
                          double nv_floor, void* stream) {                                                         \
    NRX_REQUIRE(n_code_blocks >= 1, NRX_E_ARG, "nrx_qam_demap_cb: n_code_blocks must be >= 1");                    \
    return demap_entry<T, TL>(syms, sym_stride, scales, noise_var, nv_stride, scr, qm, re_index, n_sym, llr,       \
                              llr_stride, n_batch, 0, nv_floor, stream, n_code_blocks, n_layers);                  \
  }
NRX_DEMAP_CB(nrx_qam_demap_cb_f32, float, float)
NRX_DEMAP_CB(nrx_qam_demap_cb_f64, double, double)
NRX_DEMAP_CB(nrx_qam_demap_cb_f64o32, double, float)
// The demapper whose stores do the whole rate recovery of a FIRST transmission (rv 0, no wrap-around repetition: E_r <= N - F
// for every block): llr = the (n_batch * C, N) buffer nrx_ldpc_rate_recover_* would write, columns [0, n_cols) of the punctured
// code word only (n_cols * Zc positions; behind them only transmitted positions are written -- pass the columns the decoder will read).
#define NRX_DEMAP_RR(NAME, T, TL)                                                                                  \
  extern "C" int32_t NAME(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,         \
                          int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym, \
                          const nrx_ldpc_cfg* cfg, int32_t n_layers, int32_t n_cols, void* llr, int32_t n_batch,   \
                          double nv_floor, void* stream) {                                                         \
    NRX_REQUIRE(cfg && cfg->C >= 1 && n_layers >= 1 && qm >= 1, NRX_E_ARG, "nrx_qam_demap_rr: bad configuration"); \
    NRX_REQUIRE(n_cols >= 1 && n_cols * cfg->Zc <= cfg->N, NRX_E_ARG, "nrx_qam_demap_rr: n_cols outside the code word"); \
    const int f_ = n_layers * qm, gb_ = (n_sym * qm + f_ - 1) / f_;                                                \
    const int e_max = ((gb_ + cfg->C - 1) / cfg->C) * f_;                                                          \
    if (e_max > cfg->N - cfg->F) {                                                                                 \
      ::nrx::set_error("nrx_qam_demap_rr: E_r = %d wraps around the %d-position buffer (repetition): use nrx_ldpc_rate_recover", e_max, cfg->N - cfg->F); \
      return NRX_E_UNSUPPORTED;                                                                                    \
    }                                                                                                              \
    return demap_entry<T, TL>(syms, sym_stride, scales, noise_var, nv_stride, scr, qm, re_index, n_sym, llr,       \
                              (int64_t)cfg->C * cfg->N, n_batch, 0, nv_floor, stream, cfg->C, n_layers, cfg, n_cols); \
  }
NRX_DEMAP_RR(nrx_qam_demap_rr_f32, float, float)
NRX_DEMAP_RR(nrx_qam_demap_rr_f64, double, double)
NRX_DEMAP_RR(nrx_qam_demap_rr_f64o32, double, float)

extern "C" int32_t nrx_pdsch_populate_f32(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_inv, const void* templ, const int64_t* templ_sel, int64_t elems, void* out, int32_t n_batch, int32_t planes, void* stream) { return populate_entry<float>(bits, bits_stride, scr, qm, re_inv, templ, templ_sel, elems, out, n_batch, planes, stream); }
extern "C" int32_t nrx_pdsch_populate_f64(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_inv, const void* templ, const int64_t* templ_sel, int64_t elems, void* out, int32_t n_batch, int32_t planes, void* stream) { return populate_entry<double>(bits, bits_stride, scr, qm, re_inv, templ, templ_sel, elems, out, n_batch, planes, stream); }
