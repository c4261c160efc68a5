// Throughput variant of the layered min-sum LDPC decoder (f32, hard decisions of the K information bits).
//
// Same algorithm and reference quirks as nrx_ldpc_dec.hip (reference ldpc.py:1495-1581); what changes is the
// data movement, designed around the CDNA4 issue model (a SIMD issues one VALU instruction every 2 cycles only
// when >= 2 waves feed it; a lone wave gets one every 4):
//   * two code blocks (2 x 6 wave64) per CU: <= 168 VGPRs, 40 KB LDS per workgroup;
//   * ROTATED COLUMN STORAGE: every core column sits in LDS rotated by the shift of the layer that touched it
//     last, so a layer WRITES at its own lane index (address = lane*4 + immediate) and READS at lane + delta,
//     delta = (shift_this - shift_previous) mod Zc precomputed per (Zc, edge) in constant memory; no address is
//     kept across the two passes of a layer;
//   * check-node state per (layer, lane): 0.75*min1, 0.75*min2 in two VGPRs, argmin + sign bits packed two
//     layers per VGPR for the degree <= 10 layers;
//   * channel LLRs of the degree-1 extension columns are prefetched three layers ahead from L2 instead of living
//     in 42 registers;
//   * min1/min2 by v_min_f32 / v_med3_f32, sign bits shifted in with v_alignbit_b32.
#include <stdlib.h>
#include <utility>
#include "gen_ldpc_bg.h"
#include "nrx_common.h"

namespace nrx_dec2 {

constexpr int ZMAX = 384;
constexpr int NZ = 51;
constexpr int ESTRIDE = 320;

struct ZList {
  int16_t z[NZ];
  int8_t ils[NZ];
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
constexpr ZList kZ = make_zlist();

template <int BG> struct G;
template <> struct G<1> {
  static constexpr int ROWS = NRX_BG1_ROWS, COLS = NRX_BG1_COLS, EDGES = NRX_BG1_EDGES, KB = 22, CORE = 26;
  static constexpr int row_start(int r) { return kBg1RowStart[r]; }
  static constexpr int col(int e) { return kBg1Col[e]; }
  static constexpr int shift(int ils, int e) { return kBg1Shift[ils][e]; }
};
template <> struct G<2> {
  static constexpr int ROWS = NRX_BG2_ROWS, COLS = NRX_BG2_COLS, EDGES = NRX_BG2_EDGES, KB = 10, CORE = 14;
  static constexpr int row_start(int r) { return kBg2RowStart[r]; }
  static constexpr int col(int e) { return kBg2Col[e]; }
  static constexpr int shift(int ils, int e) { return kBg2Shift[ils][e]; }
};

// d4[zi][cls][e]: core edge: byte offset 4*((shift_e - rot_prev) mod Zc) with rot_prev = rotation the column was
// left in by the previous layer that touched it (cls 0: first iteration, columns start unrotated; cls 1: steady
// state, wraps around from the last layer of the previous iteration).  Extension edge: 4*(shift_e mod Zc).
// rho4[zi][c]: 4*(final rotation of core column c).
struct FastTab {
  int32_t d4[NZ][2][ESTRIDE];
  int32_t rho4[NZ][32];
};
template <int BG> constexpr FastTab make_tab() {
  using B = G<BG>;
  FastTab t{};
  for (int zi = 0; zi < NZ; ++zi) {
    const int z = kZ.z[zi], ils = kZ.ils[zi];
    int last[32] = {};
    for (int c = 0; c < 32; ++c) last[c] = 0;
    // final rotation of each column = shift of its last edge
    int fin[32] = {};
    for (int e = 0; e < B::EDGES; ++e)
      if (B::col(e) < B::CORE) fin[B::col(e)] = B::shift(ils, e) % z;
    for (int c = 0; c < 32; ++c) t.rho4[zi][c] = 4 * fin[c];
    for (int cls = 0; cls < 2; ++cls) {
      int rot[32] = {};
      for (int c = 0; c < 32; ++c) rot[c] = cls == 0 ? 0 : fin[c];
      for (int e = 0; e < B::EDGES; ++e) {
        const int s = B::shift(ils, e) % z;
        const int c = B::col(e);
        if (c < B::CORE) {
          t.d4[zi][cls][e] = 4 * (((s - rot[c]) % z + z) % z);
          rot[c] = s;
        } else {
          t.d4[zi][cls][e] = 4 * s;
        }
      }
    }
  }
  return t;
}
constexpr FastTab kTabC1 = make_tab<1>();
constexpr FastTab kTabC2 = make_tab<2>();
__constant__ FastTab kTab1 = kTabC1;
__constant__ FastTab kTab2 = kTabC2;
template <int BG> constexpr int ctab_d4(int zi, int cls, int e) { return BG == 1 ? kTabC1.d4[zi][cls][e] : kTabC2.d4[zi][cls][e]; }
template <int BG> constexpr int ctab_rho4(int zi, int c) { return BG == 1 ? kTabC1.rho4[zi][c] : kTabC2.rho4[zi][c]; }

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float clip10(float x) { return fminf(fmaxf(x, -1e10f), 1e10f); }
__device__ __forceinline__ float sign_from(uint32_t signsrc, float mag) {  // mag >= 0
  return __uint_as_float((signsrc & 0x80000000u) | __float_as_uint(mag));
}
__device__ __forceinline__ uint32_t wrap4(uint32_t a4, uint32_t zc4) {  // a4 in [0, 2*zc4)
  const uint32_t b = a4 - zc4;
  return a4 < b ? a4 : b;
}

// prefetch distance (layers) of the extension-column channel LLRs; must divide the number of extension layers
// (42 for BG1, 38 for BG2) so that ring slot = ordinal mod PFN stays consistent across iterations
template <int BG> constexpr int pfn() { return BG == 1 ? 3 : 2; }

template <int BG> struct Lay {  // compile-time layer facts
  using B = G<BG>;
  static constexpr int deg(int L) { return B::row_start(L + 1) - B::row_start(L); }
  static constexpr bool has_ext(int L) { return B::col(B::row_start(L + 1) - 1) >= B::CORE; }
  static constexpr int ext_col(int L) { return B::col(B::row_start(L + 1) - 1); }
  static constexpr bool wide(int L) { return deg(L) > 10; }  // needs a whole 32-bit sign/argmin word
  static constexpr int n_wide() { int n = 0; for (int l = 0; l < B::ROWS; ++l) n += wide(l) ? 1 : 0; return n; }
  // slot of layer L among the wide / narrow layers
  static constexpr int wide_idx(int L) { int n = 0; for (int l = 0; l < L; ++l) n += wide(l) ? 1 : 0; return n; }
  static constexpr int narrow_idx(int L) { int n = 0; for (int l = 0; l < L; ++l) n += wide(l) ? 0 : 1; return n; }
  static constexpr int n_narrow() { return B::ROWS - n_wide(); }
  static constexpr int ext_idx(int L) { int n = 0; for (int l = 0; l < L; ++l) n += has_ext(l) ? 1 : 0; return n; }
  static constexpr int n_ext() { return ext_idx(B::ROWS); }
  static constexpr int first_ext() { for (int l = 0; l < B::ROWS; ++l) if (has_ext(l)) return l; return -1; }
  // Core columns of layer L as a bit mask (every column of a layer is both read and written by it).
  static constexpr uint32_t core_mask(int L) {
    uint32_t m = 0;
    for (int e = B::row_start(L); e < B::row_start(L + 1); ++e)
      if (B::col(e) < B::CORE) m |= 1u << B::col(e);
    return m;
  }
  // A workgroup barrier is needed before layer L only if L touches a column that some layer since the previous
  // barrier touched.  Most extension rows of both base graphs meet their neighbour in no core column at all, so
  // about a third of the barriers go (BG1: 32 of 46 remain).  Steady-state placement, computed over the cyclic
  // layer order; `barriers_ok` re-checks it from a cold start.
  struct BarPlan { bool need[B::ROWS]; bool ok; int count; };
  static constexpr BarPlan make_plan() {
    BarPlan p{};
    uint32_t mask[B::ROWS] = {};
    for (int l = 0; l < B::ROWS; ++l) mask[l] = core_mask(l);
    uint32_t touched = 0;
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        const bool need = (mask[l] & touched) != 0;
        touched = need ? mask[l] : (touched | mask[l]);
        if (it == 2) p.need[l] = need;
      }
    p.ok = true;
    p.count = 0;
    touched = 0;   // cold start: the initial fill is followed by a barrier
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        if (p.need[l]) touched = 0;
        if (mask[l] & touched) p.ok = false;
        touched |= mask[l];
      }
    for (int l = 0; l < B::ROWS; ++l) p.count += p.need[l] ? 1 : 0;
    return p;
  }
  static constexpr BarPlan plan = make_plan();
  static constexpr bool barrier_before(int L) { return plan.need[L]; }
  static constexpr bool barriers_ok() { return plan.ok; }
  // the k-th layer (cyclically) with an extension column after layer L
  static constexpr int next_ext(int L, int k) {
    int l = L;
    for (int i = 0; i < k; ++i) {
      do { l = (l + 1) % B::ROWS; } while (!has_ext(l));
    }
    return l;
  }
};

// every extension (degree-1 parity) column of both base graphs has shift 0: row z meets element z of its column
template <int BG> constexpr bool ext_shifts_are_zero() {
  using B = G<BG>;
  for (int ils = 0; ils < 8; ++ils)
    for (int e = 0; e < B::EDGES; ++e)
      if (B::col(e) >= B::CORE && B::shift(ils, e) != 0) return false;
  return true;
}

// pointers into the __constant__ tables, typed as constant address space so that loads are scalar (s_load)
typedef const int32_t __attribute__((address_space(4))) * ctab_t;

// ZI >= 0: specialised for lifting size kZ.z[ZI] (every rotation a compile-time immediate, columns stored twice
// back to back so that reads at lane+delta never wrap: zero address arithmetic per edge).  ZI < 0: any Zc.
// NS = code blocks per workgroup.  A 5- or 6-wave workgroup lands 2,2,1,1 on the four SIMDs and, at 168 VGPRs
// (3 waves per SIMD), the hardware never co-schedules a second one (measured: tools/ubench/occ_test.hip), so a CU
// would run 1.5 waves per SIMD.  Two code blocks side by side in one 10-/12-wave workgroup fill 3 waves per SIMD.
template <int BG, int ZI, int NS>
__global__ void __launch_bounds__(ZMAX * NS, 3)
ldpc_dec_fast_kernel(const float* __restrict__ llr, int n_cb, int zc_rt, int n_iter, uint8_t* __restrict__ hard,
                     ctab_t tab0, ctab_t tab1, ctab_t rho4) {
  static_assert(ext_shifts_are_zero<BG>(), "extension columns are expected to be unshifted");
  using B = G<BG>;
  using Y = Lay<BG>;
  constexpr bool SPEC = ZI >= 0;
  constexpr int ZC = SPEC ? kZ.z[SPEC ? ZI : 0] : ZMAX;   // compile-time lifting size (SPEC)
  constexpr int CSTR = SPEC ? 2 * ZC : ZMAX;              // column stride in floats
  // Core-column posteriors, column c rotated by its last layer's shift.  Static allocation: the LDS base is a
  // compile-time constant, so column offsets (and, when SPEC, the rotations) fold into the DS immediates.
  __shared__ float P[NS * B::CORE * CSTR];
  const int zc = SPEC ? ZC : zc_rt;
  // slot = which of the workgroup's NS code blocks this wave works on (wave-uniform: slots are whole waves)
  const int tz = (int)blockDim.x / NS;
  const int slot = NS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / tz);
  const int z = (int)threadIdx.x - slot * tz;
  const uint32_t sb = (uint32_t)slot * (uint32_t)(B::CORE * CSTR * 4);   // byte offset of the slot's columns
  float* const Ps = P + slot * (B::CORE * CSTR);
  const uint32_t zc4 = 4u * (uint32_t)zc;
  const int N = (B::COLS - 2) * zc, K = B::KB * zc;
  constexpr int PFN = pfn<BG>();
  static_assert(Y::n_ext() % PFN == 0, "prefetch ring must divide the number of extension layers");
  static_assert(Y::barriers_ok(), "barrier placement leaves a column hazard");
  constexpr uint32_t HI = 49152;                          // second DS base: immediates are 16 bit

  float m1[B::ROWS], m2[B::ROWS];
  uint32_t sgw[Y::n_wide() > 0 ? Y::n_wide() : 1];
  uint32_t sgn[(Y::n_narrow() + 1) / 2];  // two 16-bit fields per word

  for (int cb0 = blockIdx.x * NS; cb0 < n_cb; cb0 += gridDim.x * NS) {
    const int cb = cb0 + slot;
    const bool active = z < zc && cb < n_cb;
    const float* in = llr + (size_t)(cb < n_cb ? cb : 0) * N;
    if (active) {
      static_for<B::CORE>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        // + 0.0f turns an input -0.0 into +0.0 (the reference's sign test is (v < 0))
        if constexpr (SPEC) {
          // stored pre-rotated by the column's end-of-iteration rotation, so iteration 0 uses the steady-state deltas
          constexpr uint32_t r4 = (uint32_t)ctab_rho4<BG>(SPEC ? ZI : 0, c);
          const float v = (c < 2) ? 0.0f : clip10(in[(c - 2) * zc + (int)(wrap4(4u * (uint32_t)z + r4, zc4) >> 2)]) + 0.0f;
          Ps[c * CSTR + z] = v;
          Ps[c * CSTR + ZC + z] = v;
        } else {
          Ps[c * CSTR + z] = (c < 2) ? 0.0f : clip10(in[(c - 2) * zc + z]) + 0.0f;
        }
      });
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        m1[L] = 0.0f;
        m2[L] = 0.0f;
      });
      static_for<(Y::n_wide() > 0 ? Y::n_wide() : 1)>([&](auto i) __attribute__((always_inline)) { sgw[decltype(i)::value] = 0u; });
      static_for<(Y::n_narrow() + 1) / 2>([&](auto i) __attribute__((always_inline)) { sgn[decltype(i)::value] = 0u; });
    }
    // prefetch ring of extension-column channel LLRs (element z of the column: extension shifts are zero)
    float epf[PFN];
    auto ext_load = [&](int L, uint32_t zb) __attribute__((always_inline)) -> float {
      // zb = 4*lane, opaque to the optimiser: keeps the 42 (loop-invariant) addresses from being hoisted
      const char* col = (const char*)(in + (Y::ext_col(L) - 2) * zc);
      return active ? __builtin_amdgcn_fmed3f(*(const float*)(col + zb), -1e10f, 1e10f) + 0.0f : 0.0f;
    };
    static_for<PFN>([&](auto k) __attribute__((always_inline)) {
      constexpr int Lk = Y::next_ext(Y::first_ext() == 0 ? B::ROWS - 1 : Y::first_ext() - 1, decltype(k)::value + 1);
      epf[decltype(k)::value] = ext_load(Lk, 4u * (uint32_t)z);
    });
    __syncthreads();

    for (int it = 0; it < n_iter; ++it) {
      ctab_t tab_it = (SPEC || it != 0) ? tab1 : tab0;
#ifndef NRX_DEC2_LAYERS
#define NRX_DEC2_LAYERS B::ROWS
#endif
      static_for<NRX_DEC2_LAYERS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = B::row_start(L);
        constexpr int D = Y::deg(L);
        constexpr bool EXT = Y::has_ext(L);
        constexpr int DC = EXT ? D - 1 : D;  // core edges
        constexpr bool WIDE = Y::wide(L);
        // opaque copies keep the (loop-invariant) address arithmetic and the scalar table loads inside the layer
        uint32_t z4 = 4u * (uint32_t)z;
        ctab_t d4 = tab_it;
        asm volatile("" : "+v"(z4), "+s"(d4));
        uint32_t z4s = z4 + sb;                                 // lane's byte address inside its slot's columns
        uint32_t z4hi = z4s + HI;
        if constexpr (SPEC) asm volatile("" : "+v"(z4hi));
        float t[D];
        bool was_min[DC > 0 ? DC : 1];
        float om1 = 0.0f, om2 = 0.0f;
        uint32_t word = 0;
        int top = 31;   // left shift that brings bit 0 of the layer's sign field to bit 31
        if (active) {
          // ---- pass 1a: issue every LDS read of the layer
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            if constexpr (SPEC) {
              constexpr uint32_t off = (uint32_t)(col * CSTR * 4 + ctab_d4<BG>(SPEC ? ZI : 0, 1, E0 + j));
              if constexpr (off < 65536) t[j] = *(const float*)((const char*)P + z4s + off);
              else t[j] = *(const float*)((const char*)P + z4hi + (off - HI));
            } else {
              const uint32_t a = wrap4(z4 + (uint32_t)d4[E0 + j], zc4);
              t[j] = *(const float*)((const char*)Ps + col * CSTR * 4 + a);
            }
          });
          __builtin_amdgcn_sched_barrier(0);
          // ---- old state.  Sign/argmin word of the layer: sign of edge j at bit j of its field, argmin above.
          om1 = m1[L];
          om2 = m2[L];
          uint32_t oidx;
          if constexpr (WIDE) {
            word = sgw[Y::wide_idx(L)];
            oidx = word >> 24;
            top = 31;
          } else {
            constexpr int ni = Y::narrow_idx(L);
            word = sgn[ni / 2];
            oidx = (ni & 1) ? (word >> 28) : ((word >> 12) & 15u);
            top = (ni & 1) ? 15 : 31;
          }
          asm volatile("" : "+v"(oidx));   // keep it a plain VGPR compare (no SDWA byte-select + constant moves)
          // All "was edge j the minimum" tests first, into SGPR pairs: a VALU write of VCC/SGPR needs two wait
          // states before a v_cndmask may read it, batching avoids the s_nops (and fills the LDS latency).
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            was_min[j] = oidx == (uint32_t)j;
          });
        }
        // READ -> WRITE barrier.  Columns are stored rotated: a lane writes its own index but reads lane + delta, i.e.
        // elements other waves are about to overwrite in this very layer.  Every wave's reads of the layer must have
        // returned before any wave writes.  (It sits where the wave would wait for its LDS data anyway.)
        __syncthreads();
        if (active) {
          // ---- pass 1b: t_j = r_j - msg_old_j
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const float mag = was_min[j] ? om2 : om1;
            t[j] = t[j] - sign_from(word << (top - j), mag);
          });
          if constexpr (EXT) {
            constexpr int slot = Y::ext_idx(L) % PFN;
            t[D - 1] = epf[slot];
            constexpr int Ln = Y::next_ext(L, PFN);
            epf[slot] = ext_load(Ln, z4);
          }
          // ---- min-sum: two smallest magnitudes and the sign parity; no compares, no argmin here
          float a1 = __builtin_fabsf(t[0]), a2 = 3.0e38f;
          uint32_t px = __float_as_uint(t[0]);
          static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value + 1;
            const float a = __builtin_fabsf(t[j]);
            a2 = __builtin_amdgcn_fmed3f(a1, a2, a);                       // second minimum so far
            a1 = __builtin_amdgcn_fmed3f(a1, a, -__builtin_inff());        // min(a1, a) without a canonicalising max
            px ^= __float_as_uint(t[j]);
          });
          // QUIRK ldpc.py:1563 (second minimum taken after adding +100000 to the signed argmin entry): it can
          // only bite when every other entry exceeds ~5e4 (filler / saturated LLRs) -- wave-uniform cold path.
          if (__builtin_amdgcn_ballot_w64(a2 > 5.0e4f) != 0) {
            float v = t[D - 1];
            static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 2 - decltype(jc)::value;
              v = __builtin_fabsf(t[j]) == a1 ? t[j] : v;     // ends at the first index holding the minimum
            });
            const float q = __builtin_fabsf(v + 100000.0f);
            a2 = (a2 > 5.0e4f && q < a2) ? q : a2;
          }
          const float nm1 = a1 * 0.75f, nm2 = a2 * 0.75f;
          m1[L] = nm1;
          m2[L] = nm2;
          // ---- pass 2 (last edge first): r_j = t_j + msg_new_j, written at the lane's own index (the column is now
          // rotated by this layer's shift).  The minimum's position falls out of the magnitude test: an entry equal
          // to min1 gets min2 (with ties min2 == min1, so every tied entry may take it); first such index = argmin.
          bool is_min[D];
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            is_min[j] = __builtin_fabsf(t[j]) == a1;
          });
          __builtin_amdgcn_sched_barrier(0);
          uint32_t nsg = 0, idx = 0;
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            const uint32_t sx = px ^ __float_as_uint(t[j]);   // bit 31 = parity ^ sign(t_j)
            nsg = __builtin_amdgcn_alignbit(nsg, sx, 31);     // (nsg << 1) | (sx >> 31): edge j ends at bit j
            idx = is_min[j] ? (uint32_t)j : idx;
            if constexpr (col < B::CORE) {
              const float mag = is_min[j] ? nm2 : nm1;
              const float r = t[j] + sign_from(sx, mag);
              if constexpr (SPEC) {
                constexpr uint32_t off = (uint32_t)(col * CSTR * 4);
                float* w = (float*)((char*)P + (off < 49152 ? z4s : z4hi) + (off < 49152 ? off : off - HI));
                w[0] = r;
                w[ZC] = r;                                    // second copy (reads at lane+delta never wrap)
              } else {
                *(float*)((char*)Ps + col * CSTR * 4 + z4) = r;
              }
            }
          });
          if constexpr (WIDE) {
            sgw[Y::wide_idx(L)] = nsg | (idx << 24);
          } else {
            constexpr int ni = Y::narrow_idx(L);
            const uint32_t f = nsg | (idx << 12);             // 16-bit field: signs [D-1:0], argmin [15:12]
            if constexpr (ni & 1) sgn[ni / 2] = (word & 0x0000ffffu) | (f << 16);
            else sgn[ni / 2] = (word & 0xffff0000u) | f;
          }
        }
        // WRITE -> READ barrier only where the next layer reads a column this one wrote; a later layer is already
        // separated from this one's writes by the read->write barrier in between.
        if constexpr ((Y::core_mask(L) & Y::core_mask((L + 1) % B::ROWS)) != 0) __syncthreads();
      });
    }
    __syncthreads();   // the last layers may have run without one

    // ---- hard decisions of the information columns, un-rotating each column (ldpc.py:1578-1581)
    if (active) {
      static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const uint32_t r4 = SPEC ? (uint32_t)ctab_rho4<BG>(SPEC ? ZI : 0, c) : (uint32_t)rho4[c];
        const uint32_t e4 = wrap4(4u * (uint32_t)z + r4, zc4);
        hard[(size_t)cb * K + c * zc + (e4 >> 2)] = Ps[c * CSTR + z] < 0.0f ? 1 : 0;
      });
    }
    __syncthreads();
  }
}

constexpr int zindex_c(int zc) {
  for (int i = 0; i < NZ; ++i)
    if (kZ.z[i] == zc) return i;
  return -1;
}

int zindex(int zc, int ils) {
  for (int i = 0; i < NZ; ++i)
    if (kZ.z[i] == zc && kZ.ils[i] == ils) return i;
  return -1;
}

}  // namespace nrx_dec2

// Called by nrx_ldpc_decode_f32 (nrx_ldpc_dec.hip) for the (hard bits, K columns) case.
int32_t nrx_ldpc_decode_fast_launch(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                    uint8_t* hard, hipStream_t st) {
  using namespace nrx_dec2;
  const int zi = zindex(cfg->Zc, cfg->iLS);
  NRX_REQUIRE(zi >= 0, NRX_E_ARG, "nrx_ldpc_decode: (Zc=%d, iLS=%d) is not a lifting size", cfg->Zc, cfg->iLS);
  const int tz = ((cfg->Zc + 63) / 64) * 64;     // threads per code block (whole waves)
  const int ns = tz >= 320 ? 2 : 1;              // 5-/6-wave code blocks go two to a workgroup (see the kernel)
  const int threads = tz * ns;
  const int n_wg = (n_cb + ns - 1) / ns;
  const int grid = n_wg < 1024 ? n_wg : 1024;
  // device address of the per-(BG) constant table (resolved once per process and device)
  static const FastTab* base[2] = {nullptr, nullptr};
  const int bi = cfg->bg - 1;
  if (!base[bi]) {
    void* p = nullptr;
    const hipError_t e = bi == 0 ? hipGetSymbolAddress(&p, HIP_SYMBOL(kTab1)) : hipGetSymbolAddress(&p, HIP_SYMBOL(kTab2));
    NRX_REQUIRE(e == hipSuccess && p, NRX_E_HIP, "nrx_ldpc_decode: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    base[bi] = (const FastTab*)p;
  }
  const int32_t* t0 = &base[bi]->d4[zi][0][0];
  const int32_t* t1 = &base[bi]->d4[zi][1][0];
  const int32_t* rh = &base[bi]->rho4[zi][0];
  static const bool no_spec = getenv("NRX_LDPC_NOSPEC") != nullptr;
#define NRX_DEC2_LAUNCH(BGN, ZIV, NSV)                                                                               \
  hipLaunchKernelGGL((ldpc_dec_fast_kernel<BGN, ZIV, NSV>), dim3(grid), dim3(threads), 0, st, llr, n_cb, cfg->Zc, n_iter, \
                     hard, (ctab_t)t0, (ctab_t)t1, (ctab_t)rh)
  // lifting sizes with a specialised instantiation (the sizes of the BASELINE configurations); everything else
  // runs the generic kernel
  constexpr int ZI384 = zindex_c(384), ZI352 = zindex_c(352), ZI256 = zindex_c(256);
  if (cfg->bg == 1) {
    if (!no_spec && zi == ZI384) NRX_DEC2_LAUNCH(1, ZI384, 2);
    else if (!no_spec && zi == ZI352) NRX_DEC2_LAUNCH(1, ZI352, 2);
    else if (ns == 2) NRX_DEC2_LAUNCH(1, -1, 2);
    else NRX_DEC2_LAUNCH(1, -1, 1);
  } else {
    if (!no_spec && zi == ZI256) NRX_DEC2_LAUNCH(2, ZI256, 1);
    else if (ns == 2) NRX_DEC2_LAUNCH(2, -1, 2);
    else NRX_DEC2_LAUNCH(2, -1, 1);
  }
#undef NRX_DEC2_LAUNCH
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_f32(fast)");
  return NRX_OK;
}
