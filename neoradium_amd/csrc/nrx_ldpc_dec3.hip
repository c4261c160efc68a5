// Float64 layered min-sum LDPC decoder with the WHOLE working set on chip (hard decisions of the K information bits,
// bit-identical to the reference's float64 arithmetic, ldpc.py:1495-1581).
//
// nrx_ldpc_dec.hip keeps the float64 check-node state of all 46 (42) rows in an L2/MALL-resident workspace, because
// 527 KB per code block cannot live on chip.  NR LDPC is raptor-like, though: at the code rates the link actually
// runs, only the first RA rows have their extension parity transmitted and the others are exact no-ops for the
// information bits (see nrx_ldpc_decode_rows_*).  For RA <= 16 everything fits:
//   * LDS: the 26 core columns of TWO code blocks, float64, one buffer per column, stored UNROTATED:
//     2 x 26 x 384 x 8 B = 156 KB.  Lane z (= check row z of every layer) reads element (z + shift) mod Zc and writes
//     its result back to the same element, so nothing crosses lanes inside a layer and a workgroup barrier is only
//     needed between layers that share a column (Lay::plan_in);
//   * VGPRs (<= 168, three waves per SIMD): 0.75*min1 and 0.75*min2 of every layer (4 registers per layer), the packed
//     sign/argmin words, and the posterior of every layer's degree-1 extension column (2 registers per layer; in
//     float64 (r - m) + m' is not the channel LLR again, so it has to be carried);
//   * the lifting size is a template parameter: every (column, shift) pair is a DS immediate, the wrap-around of
//     (z + shift) is one v_cndmask between two base registers, selected by a wave-uniform 64-bit mask that a scalar
//     load fetches from a constant table.
// No HBM traffic between the initial fill and the hard decisions.  Arithmetic and its order are those of
// ldpc_dec_kernel<double, BG, true> (which stays the path for every other lifting size / row count and for soft output).
#include <stdlib.h>
#include <mutex>
#include "nrx_ldpc_graph.h"
#include "nrx_common.h"

namespace nrx_dec3 {
using namespace nrx_ldpc;

// m[w][e]: lanes of wave w (of a code block) whose element (z + shift_e mod Zc) runs past the end of the column
struct WrapTab {
  uint64_t m[ZMAX / 64][ESTRIDE];
};
template <int BG, int ZI, int RA> constexpr WrapTab make_wrap() {
  WrapTab t{};
  const int zc = kZ.z[ZI], ils = kZ.ils[ZI];
  for (int w = 0; w < ZMAX / 64; ++w)
    for (int e = 0; e < GR<BG, RA>::EDGES; ++e) {
      const int s = G<BG>::shift(ils, e) % zc;
      uint64_t m = 0;
      for (int l = 0; l < 64; ++l)
        if (64 * w + l + s >= zc) m |= 1ull << l;
      t.m[w][e] = m;
    }
  return t;
}
typedef const uint64_t __attribute__((address_space(4))) * mtab_t;

__device__ __forceinline__ double clip10(double x) {
  const double c = 1e10;
  return x < -c ? -c : (x > c ? c : x);
}
__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ double sign_from(uint32_t signsrc, double mag) {  // mag >= 0; sign taken from bit 31 of signsrc
  return __hiloint2double((int)((signsrc & 0x80000000u) | hi32(mag)), __double2loint(mag));
}
__device__ __forceinline__ uint32_t dbl(uint32_t w) {   // w + w as an add the optimiser cannot turn into a shift
  uint32_t r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(w));
  return r;
}

template <int BG> constexpr bool ext_shifts_are_zero() {
  using B = G<BG>;
  for (int ils = 0; ils < 8; ++ils)
    for (int e = 0; e < B::EDGES; ++e)
      if (B::col(e) >= B::CORE && B::shift(ils, e) != 0) return false;
  return true;
}

// NS = 2 code blocks per workgroup (2 x Zc/64 waves).  Lane z of a code block's waves = check row z of every layer.
template <int BG, int ZI, int RA>
__global__ void __launch_bounds__(2 * kZ.z[ZI], 3)
ldpc_dec_chip64_kernel(const double* __restrict__ llr, int n_cb, int n_iter, uint8_t* __restrict__ hard, mtab_t wtab) {
  static_assert(ext_shifts_are_zero<BG>(), "extension columns are expected to be unshifted");
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  constexpr int NS = 2;
  constexpr int ZC = kZ.z[ZI];
  constexpr int ILS = kZ.ils[ZI];
  static_assert(ZC % 64 == 0, "whole waves only: a lane beyond Zc would write into a live element");
  constexpr int ZS = ZC;                                   // column stride (doubles)
  constexpr int BUF = B::CORE * ZS;                        // one code block, doubles
  // Static allocation: the LDS base is a compile-time constant, so (column, shift) offsets fold into the DS immediates.
  // ZS doubles of padding in front keep the "base - Zc" addresses non-negative.
  __shared__ double Praw[ZS + NS * BUF];
  static_assert(sizeof(double) * (ZS + NS * BUF) <= 160 * 1024, "LDS budget");
  const int slot = __builtin_amdgcn_readfirstlane((int)threadIdx.x / ZS);     // wave-uniform: slots are whole waves
  const int z = (int)threadIdx.x - slot * ZS;
  const uint32_t sb = (uint32_t)(ZS + slot * BUF) * 8u;    // byte offset of the slot's columns inside Praw
  const mtab_t wm = wtab + __builtin_amdgcn_readfirstlane(z >> 6) * ESTRIDE;   // this wave's row of wrap masks
  const uint32_t zb0 = 8u * (uint32_t)z + sb;
  double* const Ps = Praw + ZS + slot * BUF;
  constexpr uint32_t zc8 = 8u * (uint32_t)ZC;
  constexpr int N = (B::COLS - 2) * ZC, K = B::KB * ZC;
  constexpr uint32_t HI = 40960;                           // second DS base: immediates are 16 bit
  static_assert(8 * (BUF - 1) - (int)HI < 65536, "DS immediates out of range");
  static_assert(Y::plan_in.ok, "barrier placement leaves a column hazard");
  constexpr int NEXT = Y::n_ext() > 0 ? Y::n_ext() : 1;

  double m1[B::ROWS], m2[B::ROWS];
  double rext[NEXT];                                       // posterior of each layer's extension column, element z
  uint32_t sgw[Y::n_wide() > 0 ? Y::n_wide() : 1];
  uint32_t sgn[(Y::n_narrow() + 1) / 2];                   // two 16-bit fields per word

  for (int cb0 = blockIdx.x * NS; cb0 < n_cb; cb0 += gridDim.x * NS) {
    const int cb = cb0 + slot;
    int one = 1;
    asm volatile("" : "+s"(one));                          // keeps the per-layer `if (live)` a real branch (see dec2)
    const bool live = cb < n_cb && one != 0;
    const double* in = llr + (size_t)(live ? cb : n_cb - 1) * N;
    // ---- load: clip, prepend the two punctured columns as zeros (ldpc.py:1536-1538); + 0.0 turns -0.0 into +0.0
    static_for<B::CORE>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      Ps[c * ZS + z] = (c < 2) ? 0.0 : clip10(in[(c - 2) * ZC + z]) + 0.0;
    });
    static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
      constexpr int L = decltype(lc)::value;
      m1[L] = 0.0;
      m2[L] = 0.0;
      if constexpr (Y::has_ext(L)) rext[Y::ext_idx(L)] = clip10(in[(Y::ext_col(L) - 2) * ZC + z]) + 0.0;
    });
    static_for<(Y::n_wide() > 0 ? Y::n_wide() : 1)>([&](auto i) __attribute__((always_inline)) { sgw[decltype(i)::value] = 0u; });
    static_for<(Y::n_narrow() + 1) / 2>([&](auto i) __attribute__((always_inline)) { sgn[decltype(i)::value] = 0u; });
    __syncthreads();

    // wrap masks of the layer about to run (SGPR pairs)
    uint64_t wcur[19];
    static_for<(Y::has_ext(0) ? Y::deg(0) - 1 : Y::deg(0))>([&](auto jc) __attribute__((always_inline)) {
      wcur[decltype(jc)::value] = wm[B::row_start(0) + decltype(jc)::value];
    });

    for (int it = 0; it < n_iter; ++it) {
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = B::row_start(L);
        constexpr int D = Y::deg(L);
        constexpr bool EXT = Y::has_ext(L);
        constexpr int DC = EXT ? D - 1 : D;  // core edges
        constexpr bool WIDE = Y::wide(L);
        uint32_t zbo = zb0;
        uint32_t wo = 0;   // opaque zero: keeps the layer's mask loads inside the layer
        asm volatile("" : "+v"(zbo), "+s"(wo));
        const mtab_t wml = (mtab_t)((const char __attribute__((address_space(4)))*)wm + __builtin_amdgcn_readfirstlane(wo));
        // byte addresses of element z of column 0 of this slot: plain, wrapped (- Zc), and both + HI
        const uint32_t zb = zbo, zbw = zbo - zc8, zbh = zbo + HI, zbwh = zbo - zc8 + HI;
        if (__builtin_expect(live, 1)) {
          double t[D];
          // ---- pass 1a: issue every LDS read of the layer (last edge first: the order pass 1b consumes them in)
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = DC - 1 - decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            constexpr uint32_t off = 8u * (uint32_t)(col * ZS + B::shift(ILS, E0 + j) % ZC);
            const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);            // z + shift >= Zc
            if constexpr (off < 65536) t[j] = *(const double*)((const char*)Praw + (wraps ? zbw : zb) + off);
            else t[j] = *(const double*)((const char*)Praw + (wraps ? zbwh : zbh) + (off - HI));
          });
          __builtin_amdgcn_sched_barrier(0);
          // ---- old state (sign/argmin word: argmin in the low bits of its field, sign of edge j above it)
          const double om1 = m1[L], om2 = m2[L];
          uint32_t word, oidx;
          int top;   // left shift that brings the sign of edge 0 to bit 31
          if constexpr (WIDE) {
            word = sgw[Y::wide_idx(L)];
            oidx = word & 31u;
            top = 31 - 5;
          } else {
            constexpr int ni = Y::narrow_idx(L);
            word = sgn[ni / 2];
            oidx = (ni & 1) ? ((word >> 16) & 15u) : (word & 15u);
            top = (ni & 1) ? (31 - 20) : (31 - 4);
          }
          // ---- pass 1b: t_j = r_j - msg_old_j  (ldpc.py:1550-1553); the extension column's r comes from its register
          if constexpr (EXT) t[D - 1] = rext[Y::ext_idx(L)];
          bool was_min[D];
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            was_min[j] = oidx == (uint32_t)j;
          });
          uint32_t wrun = word << (top - (D - 1));           // sign of edge D-1 at bit 31; doubled per edge
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            uint32_t wnext = 0;
            if constexpr (j > 0) wnext = dbl(wrun);
            const double mag = was_min[j] ? om2 : om1;
            t[j] = t[j] - sign_from(wrun, mag);
            wrun = wnext;
          });
          // ---- min-sum (ldpc.py:1556-1564): two smallest magnitudes by min/max, sign parity by XOR of the sign words
          double a1 = __builtin_fabs(t[0]);
          double a2 = 3.0e38;
          uint32_t px = hi32(t[0]);
          static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value + 1;
            const double a = __builtin_fabs(t[j]);
            a2 = __builtin_fmin(a2, __builtin_fmax(a1, a));
            a1 = __builtin_fmin(a1, a);
            px ^= hi32(t[j]);
          });
          // QUIRK ldpc.py:1563: min2 = min(min2, |v_argmin + 1e5|) with the SIGNED argmin entry; it can only win where
          // min2 > 5e4 (filler / saturated LLRs): wave-uniform cold path.
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(a2 > 5.0e4) != 0, 0)) {
            double v = t[D - 1];
            static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 2 - decltype(jc)::value;
              v = __builtin_fabs(t[j]) == a1 ? t[j] : v;     // ends at the first index holding the minimum
            });
            const double q = __builtin_fabs(v + 100000.0);
            a2 = q < a2 ? q : a2;
          }
          const double nm1 = a1 * 0.75, nm2 = a2 * 0.75;     // ldpc.py:1573 (the scale commutes with the sign)
          m1[L] = nm1;
          m2[L] = nm2;
          // ---- pass 2 (last edge first): r_j = t_j + msg_new_j, written back to the element it was read from.  An entry
          // equal to min1 gets min2 (with ties min2 == min1, so every tied entry may take it); first such index = argmin.
          bool is_min[D];
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            is_min[j] = __builtin_fabs(t[j]) == a1;
          });
          uint32_t nsg = 0, idx = 0;
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            const uint32_t sx = px ^ hi32(t[j]);              // bit 31 = parity ^ sign(t_j)
            nsg = __builtin_amdgcn_alignbit(nsg, sx, 31);     // (nsg << 1) | (sx >> 31): edge j ends at bit j
            idx = is_min[j] ? (uint32_t)j : idx;
            const double mag = is_min[j] ? nm2 : nm1;
            const double r = t[j] + sign_from(sx, mag);
            if constexpr (col < B::CORE) {
              constexpr uint32_t off = 8u * (uint32_t)(col * ZS + B::shift(ILS, E0 + j) % ZC);
              const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);
              if constexpr (off < 65536) *(double*)((char*)Praw + (wraps ? zbw : zb) + off) = r;
              else *(double*)((char*)Praw + (wraps ? zbwh : zbh) + (off - HI)) = r;
            } else {
              rext[Y::ext_idx(L)] = r;
            }
          });
          if constexpr (WIDE) {
            sgw[Y::wide_idx(L)] = idx | (nsg << 5);           // argmin [4:0], signs [5+D-1:5]
          } else {
            constexpr int ni = Y::narrow_idx(L);
            const uint32_t f = idx | (nsg << 4);              // 16-bit field: argmin [3:0], signs [4+D-1:4]
            if constexpr (ni & 1) sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x05040100u);   // f.lo16 : word.lo16
            else sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x03020504u);                    // word.hi16 : f.lo16
          }
          // the masks of the next layer: their last use in this layer is behind us
          __builtin_amdgcn_sched_barrier(0);
          constexpr int Ln = (L + 1) % B::ROWS;
          constexpr int DCn = Y::has_ext(Ln) ? Y::deg(Ln) - 1 : Y::deg(Ln);
          static_for<DCn>([&](auto jc) __attribute__((always_inline)) {
            wcur[decltype(jc)::value] = wml[B::row_start(Ln) + decltype(jc)::value];
          });
        }
        if constexpr (Y::barrier_in_before((L + 1) % B::ROWS)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);   // nothing migrates between layers (register pressure)
      });
    }
    __syncthreads();

    // ---- hard decisions of the information columns (ldpc.py:1578-1581)
    if (live) {
      static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        hard[(size_t)cb * K + c * ZC + z] = Ps[c * ZS + z] < 0.0 ? 1 : 0;
      });
    }
    __syncthreads();
  }
}

__constant__ WrapTab kWrap1_384_r13 = make_wrap<1, zindex_c(384), 13>();
__constant__ WrapTab kWrap1_384_r15 = make_wrap<1, zindex_c(384), 15>();

struct DevTab { const uint64_t* p[2]; bool ok; };

}  // namespace nrx_dec3

// Called by nrx_ldpc_decode_rows_f64 (nrx_ldpc_dec.hip) for hard decisions of the K information bits.
// Returns 1 when no on-chip instantiation covers (bg, Zc, n_rows): the caller then runs the workspace kernel.
int32_t nrx_ldpc_decode_chip64_launch(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                      int32_t n_rows, uint8_t* hard, hipStream_t st) {
  using namespace nrx_dec3;
  static const bool off = getenv("NRX_LDPC_NOCHIP64") != nullptr;      // developer switch: always the workspace kernel
  if (off || cfg->bg != 1 || cfg->Zc != 384 || cfg->iLS != 1 || n_rows > 15) return 1;
  // device addresses of the wrap-mask tables: per device (a __constant__ symbol has one address per device)
  static DevTab tabs[16] = {};
  static std::mutex mu;
  int dev = 0;
  NRX_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, NRX_E_HIP, "nrx_ldpc_decode_f64: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(mu);
  DevTab& dt = tabs[dev];
  if (!dt.ok) {
    void* p[2] = {};
    const hipError_t e[2] = {hipGetSymbolAddress(&p[0], HIP_SYMBOL(kWrap1_384_r13)), hipGetSymbolAddress(&p[1], HIP_SYMBOL(kWrap1_384_r15))};
    for (int i = 0; i < 2; ++i)
      NRX_REQUIRE(e[i] == hipSuccess && p[i], NRX_E_HIP, "nrx_ldpc_decode_f64: hipGetSymbolAddress(wrap masks) failed");
    for (int i = 0; i < 2; ++i) dt.p[i] = (const uint64_t*)p[i];
    dt.ok = true;
  }
  const int n_wg = (n_cb + 1) / 2;
  const int grid = n_wg < 1024 ? n_wg : 1024;
  constexpr int ZI384 = zindex_c(384);
  if (n_rows <= 13)
    hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 13>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard, (mtab_t)dt.p[0]);
  else
    hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 15>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard, (mtab_t)dt.p[1]);
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_f64(on-chip)");
  return NRX_OK;
}
