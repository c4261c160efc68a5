// Layered normalised min-sum LDPC decoder for gfx950 (MI355X).
//
// Replaces reference ldpc.py:1495-1581 (LdpcDecoder.decode).  Mapping (DESIGN.md "LDPC decoder"):
//   * one workgroup = one code block, lane z = check row z of EVERY layer (Zc <= 384 lanes = up to 6 wave64s);
//   * posterior LLRs of the "core" columns (22 info + 4 core parity for BG1, 10 + 4 for BG2 -- the only
//     columns shared between layers) live in LDS, read/written at (z + shift) mod Zc: consecutive lanes hit
//     consecutive banks, so every ds_read/ds_write is conflict free;
//   * the check-node state is kept compressed per (layer, lane): 0.75*min1, 0.75*min2, argmin and the sign
//     bits.  f32 variant: 3 VGPRs x 46 layers, never leaves the register file.  f64 variant (bit-exact with the
//     reference's float64 arithmetic): state + the degree-1 extension-column posteriors live in an L2/MALL
//     resident workspace, because 209 KB of float64 posteriors do not fit the 160 KB LDS;
//   * layers are fully unrolled (the base-graph structure is constexpr, shifts come from the kernarg segment
//     via scalar loads), one s_barrier per layer.
// Reference quirks reproduced literally: clip to +-1e10, punctured columns start at 0, sign(0)=+1 through
// (v < 0), first-index argmin, second minimum = min(|v_argmin + 1e5|, other |v|), 0.75 scaling, no early stop.
#include <stdlib.h>
#include <utility>
#include "gen_ldpc_bg.h"
#include "nrx_common.h"

namespace {

constexpr int ZMAX = 384;

// All 51 lifting sizes (TS 38.212 Table 5.3.2-1), ascending, and for each the set index iLS.
struct ZList {
  int16_t z[51];
  int8_t ils[51];
};
constexpr ZList make_zlist() {
  ZList l{};
  int n = 0;
  for (int z = 2; z <= 384; ++z) {
    const int base[8] = {2, 3, 5, 7, 9, 11, 13, 15};
    for (int i = 0; i < 8; ++i)
      for (int v = base[i]; v <= 384; v *= 2)
        if (v == z) {
          l.z[n] = (int16_t)z;
          l.ils[n] = (int8_t)i;
          ++n;
        }
  }
  return l;
}
constexpr ZList kZList = make_zlist();

// (shift mod Zc) for every lifting size and edge, resident in the constant address space so that the per-edge
// shifts are scalar loads (s_load_dword) with compile-time offsets.
constexpr int SHIFT_STRIDE = 320;
struct ModTab {
  int32_t v[51 * SHIFT_STRIDE];
};
template <int BG> constexpr ModTab make_modtab() {
  ModTab t{};
  for (int zi = 0; zi < 51; ++zi) {
    const int z = kZList.z[zi], ils = kZList.ils[zi];
    if constexpr (BG == 1) {
      for (int e = 0; e < NRX_BG1_EDGES; ++e) t.v[zi * SHIFT_STRIDE + e] = kBg1Shift[ils][e] % z;
    } else {
      for (int e = 0; e < NRX_BG2_EDGES; ++e) t.v[zi * SHIFT_STRIDE + e] = kBg2Shift[ils][e] % z;
    }
  }
  return t;
}
__constant__ ModTab kModTab1 = make_modtab<1>();
__constant__ ModTab kModTab2 = make_modtab<2>();

template <int BG> struct BgT;
template <> struct BgT<1> {
  static constexpr int ROWS = NRX_BG1_ROWS, COLS = NRX_BG1_COLS, EDGES = NRX_BG1_EDGES, KB = 22, CORE = 26;
  static constexpr int row_start(int r) { return kBg1RowStart[r]; }
  static constexpr int col(int e) { return kBg1Col[e]; }
  static __device__ __forceinline__ int shift(int i) { return kModTab1.v[i]; }
};
template <> struct BgT<2> {
  static constexpr int ROWS = NRX_BG2_ROWS, COLS = NRX_BG2_COLS, EDGES = NRX_BG2_EDGES, KB = 10, CORE = 14;
  static constexpr int row_start(int r) { return kBg2RowStart[r]; }
  static constexpr int col(int e) { return kBg2Col[e]; }
  static __device__ __forceinline__ int shift(int i) { return kModTab2.v[i]; }
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

template <typename T> struct FpBits;
template <> struct FpBits<float> {
  static __device__ __forceinline__ float with_sign(float mag, uint32_t bit) {  // mag >= 0; bit in {0,1}
    return __uint_as_float(__float_as_uint(mag) | (bit << 31));
  }
};
template <> struct FpBits<double> {
  static __device__ __forceinline__ double with_sign(double mag, uint32_t bit) {
    return __longlong_as_double(__double_as_longlong(mag) | ((long long)bit << 63));
  }
};

template <typename T> __device__ __forceinline__ T clip10(T x) {
  const T c = (T)1e10;
  return x < -c ? -c : (x > c ? c : x);
}

__device__ __forceinline__ float absT(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double absT(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float minT(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double minT(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float maxT(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double maxT(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ uint32_t hiword(float x) { return __float_as_uint(x); }        // the word holding the sign
__device__ __forceinline__ uint32_t hiword(double x) { return (uint32_t)__double2hiint(x); }

__device__ __forceinline__ int wrap(int z, int s, int zc) {
  unsigned a = (unsigned)(z + s);
  unsigned b = a - (unsigned)zc;
  return (int)(a < b ? a : b);  // unsigned min: picks a-zc when a >= zc
}

// Per-workgroup workspace of the f64 variant, [row][ZMAX] each.
// Addressed as (uniform base) + (compile-time offset of the array and layer) + (32-bit lane offset): one SGPR pair and
// one VGPR serve every access of a layer.  (With per-array lane pointers the compiler hoisted ~180 loop-invariant
// 64-bit addresses out of the iteration loop and spilled 324 VGPRs to scratch.)
template <typename T> struct ExactWs {
  static constexpr size_t bytes(int rows) { return (size_t)rows * ZMAX * (3 * sizeof(T) + sizeof(uint32_t)); }
  static constexpr size_t off_m1(int rows, int L) { return (size_t)L * ZMAX * sizeof(T); }
  static constexpr size_t off_m2(int rows, int L) { return ((size_t)rows + L) * ZMAX * sizeof(T); }
  static constexpr size_t off_rext(int rows, int L) { return ((size_t)2 * rows + L) * ZMAX * sizeof(T); }
  static constexpr size_t off_sg(int rows, int L) { return (size_t)3 * rows * ZMAX * sizeof(T) + (size_t)L * ZMAX * sizeof(uint32_t); }
};

// f64 (EXACT): the state lives in the workspace and TWO code blocks share a workgroup (LDS 2 x 80 KB, 12 waves = 3 per
// SIMD).  As two 6-wave workgroups the second one is never co-scheduled -- the first lands 2,2,1,1 on the SIMDs and at
// 168 VGPRs the dispatcher finds no room for another 2,2,1,1 (tools/ubench/occ_test.hip) -- so the kernel ran with half
// the waves it was written for.
template <typename T, int BG, bool EXACT>
__global__ void __launch_bounds__((EXACT ? 2 : 1) * ZMAX, EXACT ? 3 : 1)
ldpc_dec_kernel(const T* __restrict__ llr, int n_cb, int zc, int n_iter, int out_cols, int n_cols_in,
                uint8_t* __restrict__ hard, T* __restrict__ belief, char* __restrict__ ws, int tab_off, int n_rows) {
  using G = BgT<BG>;
  constexpr int NS = EXACT ? 2 : 1;                       // code blocks per workgroup (slots of whole waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tslot = (int)blockDim.x / NS;
  const int slot = NS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / tslot);
  T* P = (T*)smem + (size_t)slot * G::CORE * ZMAX;  // [CORE][ZMAX] per slot
  const int z = (int)threadIdx.x - slot * tslot;
  const int N = n_cols_in * zc;  // (COLS-2)*zc

  // register-resident check-node state (f32 variant)
  T m1[EXACT ? 1 : G::ROWS];
  T m2[EXACT ? 1 : G::ROWS];
  uint32_t sg[G::ROWS];   // (f64 too: 46 registers fit beside the 3-waves-per-SIMD budget and save 8 B per lane-layer)
  T ech[EXACT ? 1 : G::ROWS];  // channel LLR of the layer's degree-1 extension column, at (z+shift) mod Zc
  using W = ExactWs<T>;
  char* const wsb = EXACT ? ws + ((size_t)blockIdx.x * NS + slot) * W::bytes(G::ROWS) : nullptr;   // this slot's workspace
  constexpr int R = G::ROWS;
  // explicit global address space: after the opaque copy of the base the compiler no longer infers it and would emit
  // FLAT accesses (both counters, waits of zero)
  typedef T __attribute__((address_space(1))) * gT;
  typedef uint32_t __attribute__((address_space(1))) * gU;
  auto wsT = [](char* base, size_t off, uint32_t z) __attribute__((always_inline)) -> gT {
    return (gT)(base + off + (size_t)(z * (uint32_t)sizeof(T)));
  };
  // hot-loop form: (uniform base + compile-time offset) is made an opaque SGPR pair first, so that the access is
  // `global_load/store v, v_lane_offset, s[base]` (without this the compiler adds the lane offset first and then
  // needs two 64-bit VALU additions per access for the large constant)
  auto wsL = [](char* base, size_t off, uint32_t lane_off) __attribute__((always_inline)) -> gT {
    char* b = base + off;
    asm volatile("" : "+s"(b));
    return (gT)(b + (size_t)lane_off);
  };

  for (int cb0 = blockIdx.x * NS; cb0 < n_cb; cb0 += gridDim.x * NS) {
    const int cb = cb0 + slot;
    const bool active = z < zc && cb < n_cb;                // (a slot without a code block only keeps the barriers company)
    const T* in = llr + (size_t)(cb < n_cb ? cb : 0) * N;
    // ---- load: clip, prepend the two punctured columns as zeros (ldpc.py:1536-1538)
    if (active) {
      static_for<G::CORE>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        P[c * ZMAX + z] = (c < 2) ? (T)0 : clip10<T>(in[(c - 2) * zc + z]) + (T)0;   // + 0: -0.0 -> +0.0 (sign test is v < 0)
      });
      static_for<G::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        if constexpr (EXACT) {
          *wsT(wsb, W::off_m1(R, L), z) = (T)0;
          *wsT(wsb, W::off_m2(R, L), z) = (T)0;
          sg[L] = 0u;
          constexpr int e_last = G::row_start(L + 1) - 1;
          constexpr int col = G::col(e_last);
          if constexpr (col >= G::CORE) {
            *wsT(wsb, W::off_rext(R, L), z) = clip10<T>(in[(col - 2) * zc + wrap(z, G::shift(tab_off + e_last), zc)]) + (T)0;
          }
        } else {
          m1[L] = (T)0;
          m2[L] = (T)0;
          sg[L] = 0u;
          constexpr int e_last = G::row_start(L + 1) - 1;
          constexpr int col = G::col(e_last);
          if constexpr (col >= G::CORE) ech[L] = clip10<T>(in[(col - 2) * zc + wrap(z, G::shift(tab_off + e_last), zc)]) + (T)0;
          else ech[L] = (T)0;
        }
      });
    }
    __syncthreads();

    // f64: the state of the NEXT layer (0.75*min1, 0.75*min2, extension posterior) is fetched from the workspace
    // while the current layer computes; every layer used to start by waiting ~1-2 us for its own three loads.
    T pf_m1 = (T)0, pf_m2 = (T)0, pf_rx = (T)0;
    if constexpr (EXACT) {
      pf_m1 = *wsT(wsb, W::off_m1(R, 0), z);
      pf_m2 = *wsT(wsb, W::off_m2(R, 0), z);
      if constexpr (G::col(G::row_start(1) - 1) >= G::CORE) pf_rx = *wsT(wsb, W::off_rext(R, 0), z);
    }
    for (int it = 0; it < n_iter; ++it) {
      static_for<G::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = G::row_start(L);
        constexpr int D = G::row_start(L + 1) - E0;
        constexpr int Ln = (L + 1) % G::ROWS;
        constexpr bool NEXT_EXT = G::col(G::row_start(Ln + 1) - 1) >= G::CORE;
        // rows >= n_rows are skipped (kernel argument, uniform): the caller guarantees that their extension columns carry
        // all-zero LLRs (punctured parity), and such a row sends +-0 to every core column -- see nrx_ldpc_decode_rows_*
        if (L >= n_rows) return;
        const bool last_row = L + 1 == n_rows;   // the next active layer is layer 0 of the next iteration
        // Opaque copies: the (z+shift) mod Zc addresses and the kernarg shift loads are invariant over the
        // iteration loop; without this the compiler hoists all ~300 of them and spills.
        int zz = z;
        int so = tab_off;
        char* wb = wsb;
        asm volatile("" : "+v"(zz), "+s"(so), "+s"(wb));
        // state of this layer (fetched while the previous one ran) and the fetch for the next one: OUTSIDE the
        // `active` branch -- inside it the values become phis at the join and the register allocator copies them at
        // the end of the layer, which waits for the loads in the layer that issued them.  (Lanes z >= Zc read their own
        // unused workspace entries: ZMAX entries per row are allocated.)
        const uint32_t zo = (uint32_t)zz * (uint32_t)sizeof(T);
        const T cm1 = pf_m1, cm2 = pf_m2, cur_rx = pf_rx;
        if constexpr (EXACT) {
          pf_m1 = *wsL(wb, last_row ? W::off_m1(R, 0) : W::off_m1(R, Ln), zo);
          pf_m2 = *wsL(wb, last_row ? W::off_m2(R, 0) : W::off_m2(R, Ln), zo);
          if constexpr (NEXT_EXT) pf_rx = *wsL(wb, W::off_rext(R, Ln), zo);   // (unused by layer 0 when wrapping early)
        }
        if (active) {
          T om1, om2;
          uint32_t osg;
          if constexpr (EXACT) {
            om1 = cm1;
            om2 = cm2;
            osg = sg[L];
          } else {
            om1 = m1[L];
            om2 = m2[L];
            osg = sg[L];
          }
          const uint32_t oidx = osg >> 24;
          T t[D];
          int ad[D];
          // ---- pass 1: extrinsic values t_j = r_j - msg_old_j  (ldpc.py:1550-1553)
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int col = G::col(E0 + j);
            T p;
            if constexpr (col < G::CORE) {
              ad[j] = col * ZMAX + wrap(zz, G::shift(so + E0 + j), zc);
              p = P[ad[j]];
            } else if constexpr (EXACT) {
              ad[j] = 0;
              p = cur_rx;
            } else {
              // degree-1 extension column: r - msg_old is the channel LLR itself (up to fp32 rounding)
              ad[j] = 0;
              p = ech[L];
            }
            if constexpr (col < G::CORE || EXACT) {
              const T mag = (oidx == (uint32_t)j) ? om2 : om1;
              const T old = FpBits<T>::with_sign(mag, (osg >> j) & 1u);
              t[j] = p - old;
            } else {
              t[j] = p;
            }
          });
          // ---- min-sum (ldpc.py:1556-1564).  The two smallest magnitudes by min/max (no compare + select chains), the
          // sign parity by XOR of the sign words; the argmin falls out of an equality test: first index holding min1.
          T a1 = absT(t[0]);
          T a2 = (T)3.0e38;
          uint32_t px = hiword(t[0]);
          static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value + 1;
            const T a = absT(t[j]);
            a2 = minT(a2, maxT(a1, a));
            a1 = minT(a1, a);
            px ^= hiword(t[j]);
          });
          uint32_t idx = 0;
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            idx = absT(t[j]) == a1 ? (uint32_t)j : idx;
          });
          // QUIRK ldpc.py:1563: the argmin entry is bumped by +100000 (signed) before the 2nd min is taken, i.e.
          // min2 = min(min2, |v_argmin + 1e5|).  |v_argmin + 1e5| >= 1e5 - min1 >= 1e5 - min2, so the bump can only win
          // where min2 > 5e4 (filler / saturated LLRs): wave-uniform cold path, same expression for every lane in it.
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(a2 > (T)5.0e4) != 0, 0)) {
            T v = t[D - 1];
            static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 2 - decltype(jc)::value;
              v = idx == (uint32_t)j ? t[j] : v;
            });
            const T q = absT(v + (T)100000);
            a2 = q < a2 ? q : a2;
          }
          const T nm1 = a1 * (T)0.75, nm2 = a2 * (T)0.75;  // ldpc.py:1573 (scale commutes with the sign)
          // sign of the new message on edge j = parity ^ sign(t_j)   (no -0.0 can occur: inputs are loaded with + 0.0)
          uint32_t nsg = 0;
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            nsg = __builtin_amdgcn_alignbit(nsg, px ^ hiword(t[j]), 31);   // (nsg << 1) | sign bit
          });
          nsg |= idx << 24;
          if constexpr (EXACT) {
            *wsL(wb, W::off_m1(R, L), zo) = nm1;
            *wsL(wb, W::off_m2(R, L), zo) = nm2;
            sg[L] = nsg;
          } else {
            m1[L] = nm1;
            m2[L] = nm2;
            sg[L] = nsg;
          }
          // ---- pass 2: r_j = t_j + msg_new_j  (ldpc.py:1567-1576)
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int col = G::col(E0 + j);
            if constexpr (col < G::CORE || EXACT) {
              const T mag = (idx == (uint32_t)j) ? nm2 : nm1;
              const T nw = FpBits<T>::with_sign(mag, (nsg >> j) & 1u);
              const T r = t[j] + nw;
              if constexpr (col < G::CORE) P[ad[j]] = r;
              else *wsL(wb, W::off_rext(R, L), zo) = r;
            }
          });
        }
        __syncthreads();
      });
    }

    // ---- outputs (ldpc.py:1578-1581)
    if (active) {
      const size_t ob = (size_t)cb * out_cols;
      static_for<G::CORE>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        if (c * zc < out_cols) {
          const T r = P[c * ZMAX + z];
          if (hard) hard[ob + c * zc + z] = r < (T)0 ? 1 : 0;
          if (belief) belief[ob + c * zc + z] = r;
        }
      });
      if (out_cols > G::CORE * zc) {
        static_for<G::ROWS>([&](auto lc) __attribute__((always_inline)) {
          constexpr int L = decltype(lc)::value;
          constexpr int e = G::row_start(L + 1) - 1;
          constexpr int col = G::col(e);
          if constexpr (col >= G::CORE) {
            const int pos = wrap(z, G::shift(tab_off + e), zc);
            T r;
            if constexpr (EXACT) {
              r = *wsT(wsb, W::off_rext(R, L), z);
            } else {
              constexpr int j = e - G::row_start(L);
              const uint32_t s = sg[L];
              const T mag = ((s >> 24) == (uint32_t)j) ? m2[L] : m1[L];
              r = ech[L] + FpBits<T>::with_sign(mag, (s >> j) & 1u);
            }
            if (hard) hard[ob + col * zc + pos] = r < (T)0 ? 1 : 0;
            if (belief) belief[ob + col * zc + pos] = r;
          }
        });
      }
    }
    __syncthreads();
  }
}

template <typename T, int BG, bool EXACT>
int32_t launch(const T* llr, int n_cb, const nrx_ldpc_cfg* cfg, int n_iter, int out_cols, uint8_t* hard, T* belief,
               void* ws, size_t ws_bytes, hipStream_t st, int tab_off, int n_rows) {
  using G = BgT<BG>;
  constexpr int NS = EXACT ? 2 : 1;
  const int threads = NS * ((cfg->Zc + 63) / 64) * 64;
  const int n_wg = (n_cb + NS - 1) / NS;
  int grid = n_wg < 512 / NS ? n_wg : 512 / NS;
  if (EXACT) {
    if (const char* e = getenv("NRX_LDPC_WS_GRID")) {          // developer switch: workgroups (= 2 x 494 KB workspace slices) in flight
      const int gcap = atoi(e);
      if (gcap >= 1 && gcap < grid) grid = gcap;
    }
    const size_t per = ExactWs<T>::bytes(G::ROWS);
    NRX_REQUIRE(ws != nullptr && ws_bytes >= NS * per, NRX_E_ARG,
                "nrx_ldpc_decode_f64: workspace of >= %zu bytes required", NS * per);
    const size_t fit = ws_bytes / (NS * per);
    if ((size_t)grid > fit) grid = (int)fit;
  }
  const size_t lds = (size_t)NS * G::CORE * ZMAX * sizeof(T);
  auto kern = ldpc_dec_kernel<T, BG, EXACT>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // per device: every launch
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, llr, n_cb, cfg->Zc, n_iter, out_cols, G::COLS - 2,
                     hard, belief, (char*)ws, tab_off, n_rows);
  NRX_CHECK_LAUNCH("nrx_ldpc_decode");
  return NRX_OK;
}

}  // namespace
int32_t nrx_ldpc_decode_fast_launch(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                    int32_t n_rows, uint8_t* hard, hipStream_t st);  // nrx_ldpc_dec2.hip
int32_t nrx_ldpc_decode_chip64_launch(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                      int32_t n_rows, uint8_t* hard, hipStream_t st, void* ws, size_t ws_bytes,
                                      const int32_t* sel, const int32_t* n_sel);  // nrx_ldpc_dec3.hip
int32_t nrx_ldpc_decode_chipz_launch(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                     int32_t n_rows, uint8_t* hard, hipStream_t st, void* ws, size_t ws_bytes);   // nrx_ldpc_dec4.hip
namespace {

template <typename T, bool EXACT>
int32_t decode_entry(const T* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t out_cols,
                     uint8_t* hard, T* belief, void* ws, size_t ws_bytes, void* stream, int32_t n_rows = 0,
                     const int32_t* sel = nullptr, const int32_t* n_sel = nullptr) {
  NRX_REQUIRE(llr && cfg, NRX_E_ARG, "nrx_ldpc_decode: NULL llr/cfg");
  NRX_REQUIRE(hard || belief, NRX_E_ARG, "nrx_ldpc_decode: need hard_out or belief_out");
  NRX_REQUIRE(cfg->bg == 1 || cfg->bg == 2, NRX_E_ARG, "nrx_ldpc_decode: bg must be 1|2");
  NRX_REQUIRE(cfg->Zc >= 2 && cfg->Zc <= ZMAX && cfg->iLS >= 0 && cfg->iLS < 8, NRX_E_ARG,
              "nrx_ldpc_decode: bad Zc/iLS (%d,%d)", cfg->Zc, cfg->iLS);
  NRX_REQUIRE(n_cb >= 0 && n_iter >= 0, NRX_E_ARG, "nrx_ldpc_decode: negative count");
  const int cols = cfg->bg == 1 ? NRX_BG1_COLS : NRX_BG2_COLS;
  NRX_REQUIRE(out_cols == cfg->K || out_cols == cols * cfg->Zc, NRX_E_SHAPE,
              "nrx_ldpc_decode: out_cols must be K (%d) or all %d columns", cfg->K, cols * cfg->Zc);
  NRX_REQUIRE(cfg->N == (cols - 2) * cfg->Zc, NRX_E_SHAPE, "nrx_ldpc_decode: cfg->N inconsistent");
  const int rows_all = cfg->bg == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  if (n_rows == 0) n_rows = rows_all;
  NRX_REQUIRE(n_rows >= 4 && n_rows <= rows_all, NRX_E_ARG, "nrx_ldpc_decode_rows: n_rows must be in [4, %d]", rows_all);
  NRX_REQUIRE(n_rows == rows_all || out_cols == cfg->K, NRX_E_ARG,
              "nrx_ldpc_decode_rows: dropping rows is only defined for the K information columns");
  if (n_cb == 0) return NRX_OK;
  if constexpr (!EXACT) {
    // throughput kernel: hard decisions of the K information bits (what the link loop consumes)
    static const bool force_v1 = getenv("NRX_LDPC_V1") != nullptr;
    if (hard && !belief && out_cols == cfg->K && !force_v1)
      return nrx_ldpc_decode_fast_launch((const float*)llr, n_cb, cfg, n_iter, n_rows, hard, (hipStream_t)stream);
  }
  if constexpr (EXACT) {
    // float64, hard decisions of the information bits, few enough rows: the whole working set fits on chip
    if (hard && !belief && out_cols == cfg->K) {
      const int32_t rc = nrx_ldpc_decode_chip64_launch((const double*)llr, n_cb, cfg, n_iter, n_rows, hard, (hipStream_t)stream, ws, ws_bytes,
                                                       sel, n_sel);
      if (rc != 1) return rc;      // 1 = no Zc = 384 instantiation for this (bg, Zc, rows)
    }
  }
  if (sel) {
    ::nrx::set_error("nrx_ldpc_decode_rows_sel: selections are built for the float64 Zc = 384 kernels of nrx_ldpc_dec3.hip");
    return NRX_E_UNSUPPORTED;
  }
  int zi = -1;
  for (int i = 0; i < 51; ++i)
    if (kZList.z[i] == cfg->Zc) zi = i;
  NRX_REQUIRE(zi >= 0 && kZList.ils[zi] == cfg->iLS, NRX_E_ARG, "nrx_ldpc_decode: (Zc=%d, iLS=%d) is not a lifting size",
              cfg->Zc, cfg->iLS);
  if constexpr (EXACT) {
    // ... any other lifting size, either base graph, <= 15 rows: the on-chip kernel with the lifting size at run time
    if (hard && !belief && out_cols == cfg->K) {
      const int32_t rc = nrx_ldpc_decode_chipz_launch((const double*)llr, n_cb, cfg, n_iter, n_rows, hard, (hipStream_t)stream, ws, ws_bytes);
      if (rc != 1) return rc;      // 1 = more rows than fit on chip: workspace kernel below
    }
  }
  const int tab = zi * SHIFT_STRIDE;
  hipStream_t st = (hipStream_t)stream;
  if (cfg->bg == 1) return launch<T, 1, EXACT>(llr, n_cb, cfg, n_iter, out_cols, hard, belief, ws, ws_bytes, st, tab, n_rows);
  return launch<T, 2, EXACT>(llr, n_cb, cfg, n_iter, out_cols, hard, belief, ws, ws_bytes, st, tab, n_rows);
}

}  // namespace

extern "C" size_t nrx_ldpc_decode_ws_bytes(const nrx_ldpc_cfg* cfg, int32_t is_f64) {
  if (!cfg || !is_f64) return 0;
  const int rows = cfg->bg == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  return 512 * ExactWs<double>::bytes(rows);
}

extern "C" int32_t nrx_ldpc_decode_f32(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                       int32_t out_cols, uint8_t* hard_out, float* belief_out, void* ws,
                                       size_t ws_bytes, void* stream) {
  return decode_entry<float, false>(llr, n_cb, cfg, n_iter, out_cols, hard_out, belief_out, ws, ws_bytes, stream);
}

extern "C" int32_t nrx_ldpc_decode_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                       int32_t out_cols, uint8_t* hard_out, double* belief_out, void* ws,
                                       size_t ws_bytes, void* stream) {
  return decode_entry<double, true>(llr, n_cb, cfg, n_iter, out_cols, hard_out, belief_out, ws, ws_bytes, stream);
}

// Decoding with the first n_rows rows of the base graph only (hard decisions of the K information bits).
extern "C" int32_t nrx_ldpc_decode_rows_f32(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                            int32_t n_rows, uint8_t* hard_out, void* ws, size_t ws_bytes, void* stream) {
  NRX_REQUIRE(cfg, NRX_E_ARG, "nrx_ldpc_decode_rows: NULL cfg");
  return decode_entry<float, false>(llr, n_cb, cfg, n_iter, cfg->K, hard_out, nullptr, ws, ws_bytes, stream, n_rows);
}

extern "C" int32_t nrx_ldpc_decode_rows_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                            int32_t n_rows, uint8_t* hard_out, void* ws, size_t ws_bytes, void* stream) {
  NRX_REQUIRE(cfg, NRX_E_ARG, "nrx_ldpc_decode_rows: NULL cfg");
  return decode_entry<double, true>(llr, n_cb, cfg, n_iter, cfg->K, hard_out, nullptr, ws, ws_bytes, stream, n_rows);
}

// ... of a selection of the code blocks: sel[0 .. *n_sel) index the n_cb rows of llr / hard_out, list and count on the device
// (the launch covers the worst case, nothing is read by the host); the other rows of hard_out are left untouched.
extern "C" int32_t nrx_ldpc_decode_rows_sel_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                                int32_t n_rows, uint8_t* hard_out, void* ws, size_t ws_bytes, const int32_t* sel,
                                                const int32_t* n_sel, void* stream) {
  NRX_REQUIRE(cfg && sel && n_sel, NRX_E_ARG, "nrx_ldpc_decode_rows_sel: NULL cfg / selection");
  return decode_entry<double, true>(llr, n_cb, cfg, n_iter, cfg->K, hard_out, nullptr, ws, ws_bytes, stream, n_rows, sel, n_sel);
}
