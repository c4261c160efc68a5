// Throughput variant of the layered min-sum LDPC decoder (f32, hard decisions of the K information bits).
//
// Same algorithm and reference quirks as nrx_ldpc_dec.hip (reference ldpc.py:1495-1581); what changes is the
// data movement, designed around the CDNA4 issue model (a SIMD issues one VALU instruction every 2 cycles only
// when >= 2 waves feed it; a lone wave gets one every 4):
//   * two code blocks (2 x 6 wave64) per CU in ONE workgroup: <= 168 VGPRs (3 waves per SIMD), 157.5 KB of LDS
//     (26 core columns x 384 floats x 2 ping-pong buffers x 2 code blocks);
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
#include <mutex>
#include "nrx_ldpc_graph.h"
#include "nrx_common.h"

namespace nrx_dec2 {
using namespace nrx_ldpc;


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

struct OneTab {
  int32_t d4[ESTRIDE];   // steady-state read offsets (cls 1 of FastTab)
  int32_t rho4[32];
};
template <int BG, int RA> constexpr OneTab make_one(int zi) {
  using B = GR<BG, RA>;
  OneTab t{};
  const int z = kZ.z[zi], ils = kZ.ils[zi];
  int fin[32] = {};
  for (int e = 0; e < B::EDGES; ++e)
    if (B::col(e) < B::CORE) fin[B::col(e)] = B::shift(ils, e) % z;
  for (int c = 0; c < 32; ++c) t.rho4[c] = 4 * fin[c];
  int rot[32] = {};
  for (int c = 0; c < 32; ++c) rot[c] = fin[c];
  for (int e = 0; e < B::EDGES; ++e) {
    const int s = B::shift(ils, e) % z;
    const int c = B::col(e);
    if (c < B::CORE) {
      t.d4[e] = 4 * (((s - rot[c]) % z + z) % z);
      rot[c] = s;
    } else {
      t.d4[e] = 4 * s;
    }
  }
  return t;
}
template <int BG, int RA, int ZI> inline constexpr OneTab kOne = make_one<BG, RA>(ZI < 0 ? 0 : ZI);

// Wrap masks of the specialised kernels: m[w][e] = lanes of wave w (of a code block) whose read of edge e, element
// (z + delta_e), runs past the end of the column and wraps to (z + delta_e - Zc).  Wave-uniform 64-bit values: the
// kernel fetches a layer's masks with one scalar load and uses them directly as v_cndmask selectors.
struct WrapTab {
  uint64_t m[ZMAX / 64][ESTRIDE];
};
template <int BG, int ZI, int RA = G<BG>::ROWS> constexpr WrapTab make_wrap() {
  WrapTab t{};
  const int zc = kZ.z[ZI];
  for (int w = 0; w < ZMAX / 64; ++w)
    for (int e = 0; e < GR<BG, RA>::EDGES; ++e) {
      const int d = kOne<BG, RA, ZI>.d4[e] / 4;
      uint64_t m = 0;
      for (int l = 0; l < 64; ++l)
        if (64 * w + l + d >= zc) m |= 1ull << l;
      t.m[w][e] = m;
    }
  return t;
}
typedef const uint64_t __attribute__((address_space(4))) * mtab_t;


__device__ __forceinline__ float clip10(float x) { return fminf(fmaxf(x, -1e10f), 1e10f); }
__device__ __forceinline__ float sign_from(uint32_t signsrc, float mag) {  // mag >= 0
  return __uint_as_float((signsrc & 0x80000000u) | __float_as_uint(mag));
}
__device__ __forceinline__ uint32_t dbl(uint32_t w) {   // w + w as an add the optimiser cannot turn into a shift
  uint32_t r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(w));
  return r;
}
// ---- round 3 (as nrx_ldpc_dec3.hip, section "Messages as unit x pm"): a row keeps pm1 / pm2 = 0.75*min1 / 0.75*min2 with the
// row's sign parity as their sign; a message is u_j * pm with the unit u_j = +-1.0f carrying sign(t_j), applied by a fused
// multiply-add (the product is exact, so it rounds like the add / subtract of the signed message); the argmin lanes are put
// into EXEC by a v_cmpx and redo the operation with pm2.  Two VALU instructions per edge-visit fewer than select + sign insert
// + parity xor + sign collect (the SIMD issues one VALU instruction per 4 cycles whatever the kind).
__device__ __forceinline__ float unit_of(uint32_t signsrc) {   // +-1.0f with the sign of bit 31 of signsrc
  uint32_t h;
  asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(h) : "v"(signsrc), "s"(0x80000000u));
  return __uint_as_float(h);
}
__device__ __forceinline__ float c75_of(uint32_t signbits) {   // +-0.75f, negative for an odd number of set sign bits
  uint32_t h;
  asm("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(h) : "v"((uint32_t)__builtin_popcount(signbits)), "s"(0x3f400000u));
  return __uint_as_float(h);
}
// fma(-u, oidx == J ? p2 : p1, x)   (all 64 lanes are live: EXEC is -1 around the block)
template <int J>
__device__ __forceinline__ float sel_fnma_x(float x, uint32_t oidx, float u, float p1, float p2) {
  float y;
  asm("v_fma_f32 %[y], -%[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_u32_e32 vcc, %[j], %[oidx]\n\t"
      "v_fma_f32 %[y], -%[u], %[p2], %[x]\n\t"
      "s_mov_b64 exec, -1"
      : [y] "=&v"(y) : [x] "v"(x), [j] "n"(J), [oidx] "v"(oidx), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2) : "vcc");
  return y;
}
// fma(u, |x| == a ? p2 : p1, x); idx <- J where |x| == a (every entry equal to the minimum takes p2: with a tie min2 == min1)
template <int J>
__device__ __forceinline__ float sel_fma_x(float x, uint32_t& idx, float a, float u, float p1, float p2) {
  float y;
  asm("v_fma_f32 %[y], %[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_f32_e64 vcc, |%[x]|, %[a]\n\t"
      "v_fma_f32 %[y], %[u], %[p2], %[x]\n\t"
      "v_mov_b32 %[idx], %[j]\n\t"
      "s_mov_b64 exec, -1"
      : [y] "=&v"(y), [idx] "+v"(idx) : [x] "v"(x), [a] "v"(a), [j] "n"(J), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2) : "vcc");
  return y;
}
// min / med3 written out: after an inline-asm producer the compiler cannot prove its inputs canonical and puts a v_max_f32 x, x, x
// in front of every one (one extra VALU instruction per edge); no NaN reaches this code (LLRs are clipped on the way in)
__device__ __forceinline__ float vmed3_abs(float a, float b, float t) {
  float r;
  asm("v_med3_f32 %0, %1, %2, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
__device__ __forceinline__ float vmin_abs(float a, float t) {
  float r;
  asm("v_min_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
// idx <- J where |x| == a (an extension edge: nothing is written back)
template <int J>
__device__ __forceinline__ void mark_min(float x, uint32_t& idx, float a) {
  asm("v_cmpx_eq_f32_e64 vcc, |%[x]|, %[a]\n\t"
      "v_mov_b32 %[idx], %[j]\n\t"
      "s_mov_b64 exec, -1"
      : [idx] "+v"(idx) : [x] "v"(x), [a] "v"(a), [j] "n"(J) : "vcc");
}
// wave priority falls with progress through a layer: the SIMD always serves its oldest ready wave, lowering a wave's priority as
// it advances lets the waves that are behind go first and brings the waves of a SIMD to the layer's barrier together
#define LAYER_PRIO(Q) __builtin_amdgcn_s_setprio(3 - (Q))

__device__ __forceinline__ uint32_t wrap4(uint32_t a4, uint32_t zc4) {  // a4 in [0, 2*zc4)
  const uint32_t b = a4 - zc4;
  return a4 < b ? a4 : b;
}


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

// One workgroup = NS code blocks; lane z of a code block's waves = check row z of every layer.
//
// LDS holds the core-column posteriors, column c stored ROTATED by the shift of the last layer that wrote it: a layer
// reads element (z + delta) mod Zc and writes element z, so no address survives from the read pass to the write pass.
// Because reads cross lanes, every column has two buffers used in ping-pong (a layer reads one and writes the other):
// inside a layer no wave can overwrite what another wave still has to read, and barriers are only needed between
// layers that share a column (Lay::plan).
// ZI >= 0: specialised for lifting size kZ.z[ZI]: every rotation is a compile-time DS immediate and the wrap-around
// is one compare + select between two base registers.  ZI < 0: any Zc, rotations from the constant tables.
// NS = code blocks per workgroup.  A 5- or 6-wave workgroup lands 2,2,1,1 on the four SIMDs and, at 168 VGPRs
// (3 waves per SIMD), the hardware never co-schedules a second one (measured: tools/ubench/occ_test.hip), so a CU
// would run 1.5 waves per SIMD.  Two code blocks side by side in one 10-/12-wave workgroup fill 3 waves per SIMD.
template <int BG, int ZI, int NS, int RA = G<BG>::ROWS>
__global__ void __launch_bounds__(ZMAX * NS, 3)
ldpc_dec_fast_kernel(const float* __restrict__ llr, int n_cb, int zc_rt, int n_iter, uint8_t* __restrict__ hard,
                     ctab_t tab0, ctab_t tab1, ctab_t rho4, mtab_t wtab) {
  static_assert(ext_shifts_are_zero<BG>(), "extension columns are expected to be unshifted");
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  constexpr bool SPEC = ZI >= 0;
  static_assert(SPEC || RA == G<BG>::ROWS, "dropping rows is built for the specialised lifting sizes only");
  constexpr int ZC = SPEC ? kZ.z[SPEC ? ZI : 0] : ZMAX;   // compile-time lifting size (SPEC)
  constexpr int ZS = (ZC + 63) / 64 * 64;                  // column stride in floats (whole waves, see `live`)
  constexpr int BUF = B::CORE * ZS;                        // one buffer of one code block, floats
  constexpr int SLOT = 2 * BUF;                            // both buffers
  // Static allocation: the LDS base is a compile-time constant, so buffer/column offsets (and, when SPEC, the
  // rotations) fold into the DS immediates.  ZS floats of padding in front keep "base - Zc" addresses non-negative.
  __shared__ float Praw[ZS + NS * SLOT];
  const int zc = SPEC ? ZC : zc_rt;
  // slot = which of the workgroup's NS code blocks this wave works on (wave-uniform: slots are whole waves)
  const int tz = (int)blockDim.x / NS;
  const int slot = NS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / tz);
  const int z = (int)threadIdx.x - slot * tz;
  const uint32_t sb = (uint32_t)(ZS + slot * SLOT) * 4u;   // byte offset of the slot's buffer 0 inside Praw
  const mtab_t wm = wtab + __builtin_amdgcn_readfirstlane(z >> 6) * ESTRIDE;   // this wave's row of wrap masks
  const uint32_t zb0 = 4u * (uint32_t)z + sb;
  float* const Ps = Praw + ZS + slot * SLOT;
  const uint32_t zc4 = 4u * (uint32_t)zc;
  const int N = (B::COLS - 2) * zc, K = B::KB * zc;
  constexpr int PFN = pfn_for<BG>(Y::n_ext());
  static_assert(Y::n_ext() % PFN == 0, "prefetch ring must divide the number of extension layers");
  static_assert(Y::barriers_ok(), "barrier placement leaves a column hazard");
  constexpr uint32_t HI = 49152;                           // second DS base: immediates are 16 bit
  static_assert(SLOT * 4 - (int)HI + 4 * ZS < 65536, "DS immediates out of range");

  float m1[B::ROWS], m2[B::ROWS];
  uint32_t sgw[Y::n_wide() > 0 ? Y::n_wide() : 1];
  uint32_t sgn[(Y::n_narrow() + 1) / 2];  // two 16-bit fields per word

  for (int cb0 = blockIdx.x * NS; cb0 < n_cb; cb0 += gridDim.x * NS) {
    const int cb = cb0 + slot;
    // live: wave-uniform, this wave's code block exists.  Inside the layer loop the lanes z >= Zc of a partial last
    // wave simply run along: their LDS writes land in the padding of their own column (stride = whole waves), their
    // reads are never used, their global reads are clamped.  That keeps the loop free of EXEC masking.
    // (for NS == 1 `cb < n_cb` is a tautology; an opaque 1 keeps the per-layer `if (live)` below a real branch:
    //  without those block boundaries the 46 unrolled layers form one region and the register allocator spills)
    int one = 1;
    asm volatile("" : "+s"(one));
    const bool live = cb < n_cb && one != 0;
    const bool active = z < zc && live;
    const float* in = llr + (size_t)(live ? cb : n_cb - 1) * N;
    if (z < zc) {
      static_for<B::CORE>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        // + 0.0f turns an input -0.0 into +0.0 (the reference's sign test is (v < 0))
        if constexpr (SPEC) {
          // stored pre-rotated by the column's end-of-iteration rotation, so iteration 0 uses the steady-state deltas
          constexpr uint32_t r4 = (uint32_t)kOne<BG, RA, ZI>.rho4[c];
          Ps[c * ZS + z] = (c < 2) ? 0.0f : clip10(in[(c - 2) * zc + (int)(wrap4(4u * (uint32_t)z + r4, zc4) >> 2)]) + 0.0f;
        } else {
          Ps[c * ZS + z] = (c < 2) ? 0.0f : clip10(in[(c - 2) * zc + z]) + 0.0f;
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
      // (scalar base + 32-bit lane offset: no per-column 64-bit pointers to keep in SGPRs)
      const uint32_t zq = (SPEC && ZC % 64 == 0) ? zb : (zb < zc4 ? zb : zc4 - 4u);   // partial last wave: stay inside
      const uint32_t off = zq + (uint32_t)(Y::ext_col(L) - 2) * zc4;
      return *(const float*)((const char*)in + off);   // raw: clipped when it is consumed, PFN layers later, so that
                                                        // nothing waits for the load inside the layer that issues it
    };
    static_for<PFN>([&](auto k) __attribute__((always_inline)) {
      constexpr int Lk = Y::next_ext(Y::first_ext() == 0 ? B::ROWS - 1 : Y::first_ext() - 1, decltype(k)::value + 1);
      epf[decltype(k)::value] = ext_load(Lk, 4u * (uint32_t)z);
    });
    __syncthreads();

    // wrap masks of the layer about to run (SGPR pairs), fetched one layer ahead
    constexpr int WN = SPEC ? 19 : 1;
    uint64_t wcur[WN];
    if constexpr (SPEC) {
      static_for<(Y::has_ext(0) ? Y::deg(0) - 1 : Y::deg(0))>([&](auto jc) __attribute__((always_inline)) {
        wcur[decltype(jc)::value] = wm[B::row_start(0) + decltype(jc)::value];
      });
    }

    for (int it = 0; it < n_iter; ++it) {
      ctab_t tab_it = (SPEC || it != 0) ? tab1 : tab0;
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = B::row_start(L);
        constexpr int D = Y::deg(L);
        constexpr bool EXT = Y::has_ext(L);
        constexpr int DC = EXT ? D - 1 : D;  // core edges
        constexpr bool WIDE = Y::wide(L);
        // opaque copies keep the (loop-invariant) address arithmetic and the scalar table loads inside the layer
        uint32_t z4 = 4u * (uint32_t)z;
        ctab_t d4 = tab_it;
        uint32_t wo = 0;   // opaque zero: keeps the layer's mask loads inside the layer (readfirstlane: provably uniform)
        asm volatile("" : "+v"(z4), "+s"(d4), "+s"(wo));
        const mtab_t wml = (mtab_t)((const char __attribute__((address_space(4)))*)wm + __builtin_amdgcn_readfirstlane(wo));
        // byte addresses of element z of column 0 / buffer 0 of this slot: plain, wrapped (- Zc), and both + HI
        // (four live registers; everything else of an address is a DS immediate)
        const uint32_t zb = zb0, zbw = zb0 - zc4, zbh = zb0 + HI, zbwh = zb0 - zc4 + HI;
        float t[D];
        float a1 = 0.0f, a2 = 0.0f;
        uint32_t px = 0, word = 0;
        if (__builtin_expect(live, 1)) {   // (wave-uniform; the per-layer branch also keeps each layer its own scheduling region)
          LAYER_PRIO(0);
          // ---- pass 1a: issue every LDS read of the layer (last edge first: the order pass 1b consumes them in)
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = DC - 1 - decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            constexpr uint32_t cof = (uint32_t)((Y::touch_par(L, col) * B::CORE + col) * ZS * 4);   // read buffer
            if constexpr (SPEC) {
              constexpr uint32_t dl4 = (uint32_t)kOne<BG, RA, ZI>.d4[E0 + j];   // 4 * rotation
              constexpr uint32_t off = cof + dl4;
              const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);            // z + delta >= Zc
              if constexpr (off < 65536) t[j] = *(const float*)((const char*)Praw + (wraps ? zbw : zb) + off);
              else t[j] = *(const float*)((const char*)Praw + (wraps ? zbwh : zbh) + (off - HI));
            } else {
              const uint32_t a = wrap4(z4 + (uint32_t)d4[E0 + j], zc4);
              t[j] = *(const float*)((const char*)Ps + cof + a);
            }
          });
          __builtin_amdgcn_sched_barrier(0);
          // ---- old state.  Sign/argmin word of the layer: argmin in the low bits of its field (so that the argmin
          // selects below work with inline constants), sign of edge j at bit IB + j above it.
          const float om1 = m1[L], om2 = m2[L];
          uint32_t oidx;
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
          // ---- pass 1b: t_j = r_j - msg_old_j = fma(-u_j, argmin ? pm2 : pm1, r_j)  (ldpc.py:1550-1553)
          // sign of t_j (previous iteration) at bit 31 of a running word (last edge first, doubled per edge)
          uint32_t wrun = word << (top - (DC > 0 ? DC - 1 : 0));
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = DC - 1 - decltype(jc)::value;
            const float u = unit_of(wrun);
            if constexpr (j > 0) wrun = dbl(wrun);
            if constexpr (decltype(jc)::value == (2 * DC) / 3) LAYER_PRIO(1);
            t[j] = sel_fnma_x<j>(t[j], oidx, u, om1, om2);
          });
          if constexpr (SPEC) {
            // every LDS read of this layer has been consumed: fetch the next layer's wrap masks (scalar loads share
            // the LDS counter and return out of order, so they must not overlap the counted waits above)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int Ln = (L + 1) % B::ROWS;
            constexpr int DCn = Y::has_ext(Ln) ? Y::deg(Ln) - 1 : Y::deg(Ln);
            static_for<DCn>([&](auto jc) __attribute__((always_inline)) {
              wcur[decltype(jc)::value] = wml[B::row_start(Ln) + decltype(jc)::value];
            });
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (EXT) {
            constexpr int slot = Y::ext_idx(L) % PFN;
            // clip to +-1e10 (ldpc.py:1536); + 0.0f turns -0.0 into +0.0 (the reference's sign test is (v < 0))
            t[D - 1] = __builtin_amdgcn_fmed3f(epf[slot], -1e10f, 1e10f) + 0.0f;
          }
          // ---- min-sum: two smallest magnitudes; the signs of the t_j are collected (edge j ends at bit j), their parity is a
          // population count; no compares, no argmin here
          a1 = __builtin_fabsf(t[D - 1]);
          a2 = 3.0e38f;
          px = __float_as_uint(t[D - 1]) >> 31;
          static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 2 - decltype(jc)::value;
            if constexpr (decltype(jc)::value == (D - 1) / 2) LAYER_PRIO(2);
            a2 = vmed3_abs(a1, a2, t[j]);                                  // second minimum so far
            a1 = vmin_abs(a1, t[j]);
            px = __builtin_amdgcn_alignbit(px, __float_as_uint(t[j]), 31);  // (px << 1) | sign(t_j)
          });
          // QUIRK ldpc.py:1563 (second minimum taken after adding +100000 to the signed argmin entry): it can
          // only bite when every other entry exceeds ~5e4 (filler / saturated LLRs) -- wave-uniform cold path.
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(a2 > 5.0e4f) != 0, 0)) {
            float v = t[D - 1];
            static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 2 - decltype(jc)::value;
              v = __builtin_fabsf(t[j]) == a1 ? t[j] : v;     // ends at the first index holding the minimum
            });
            const float q = __builtin_fabsf(v + 100000.0f);
            a2 = (a2 > 5.0e4f && q < a2) ? q : a2;
          }
        }
        if constexpr (EXT) {
          // Refill the ring slot just consumed.  Issued OUTSIDE the `live` branches: with the load on one side of a
          // branch only, the compiler's wait-count bookkeeping merges "one younger load" with "none" at the join and
          // falls back to s_waitcnt vmcnt(0) at every use, i.e. it waits for the load issued one layer ago and the
          // ring is one deep instead of PFN.  Unconditional, every path has the same load sequence and the use waits
          // with vmcnt(PFN - 1).  (`in` is clamped to an existing code block when the wave is not live.)
          epf[Y::ext_idx(L) % PFN] = ext_load(Y::next_ext(L, PFN), z4);
        }
        if (__builtin_expect(live, 1)) {   // (second region: measured slightly faster than one region per layer)
          const float c75 = c75_of(px);                     // 0.75 (ldpc.py:1573) with the row parity as its sign
          const float nm1 = a1 * c75, nm2 = a2 * c75;
          m1[L] = nm1;
          m2[L] = nm2;
          // ---- pass 2 (last edge first): r_j = t_j + msg_new_j = fma(u_j, |t_j| == min1 ? pm2 : pm1, t_j), written at the lane's
          // own index of the column's other buffer (the column is now rotated by this layer's shift).  An entry equal to min1
          // gets min2 (with ties min2 == min1, so every tied entry may take it); first such index = argmin.
          const uint32_t nsg = px;
          uint32_t idx = 0;
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = D - 1 - decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            if constexpr (decltype(jc)::value == (2 * D) / 3) LAYER_PRIO(3);
            if constexpr (col < B::CORE) {
              const float r = sel_fma_x<j>(t[j], idx, a1, unit_of(__float_as_uint(t[j])), nm1, nm2);
              constexpr uint32_t wof = (uint32_t)(((Y::touch_par(L, col) ^ 1) * B::CORE + col) * ZS * 4);   // write buffer
              if constexpr (SPEC) {
                if constexpr (wof < 65536) *(float*)((char*)Praw + zb + wof) = r;
                else *(float*)((char*)Praw + zbh + (wof - HI)) = r;
              } else {
                *(float*)((char*)Ps + wof + z4) = r;
              }
            } else {
              mark_min<j>(t[j], idx, a1);
            }
          });
          if constexpr (WIDE) {
            sgw[Y::wide_idx(L)] = idx | (nsg << 5);           // argmin [4:0], signs [5+D-1:5]
          } else {
            constexpr int ni = Y::narrow_idx(L);
            const uint32_t f = idx | (nsg << 4);              // 16-bit field: argmin [3:0], signs [4+D-1:4]
            // one v_perm_b32 puts the field into its half of the word (and keeps the optimiser from folding the
            // shift into the argmin constants): bytes 7..4 = f, bytes 3..0 = word
            if constexpr (ni & 1) sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x05040100u);   // f.lo16 : word.lo16
            else sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x03020504u);                    // word.hi16 : f.lo16
          }
        }
        if constexpr (Y::barrier_before(L + 1)) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);   // nothing migrates between layers (register pressure)
      });
      // ---- copy-back: columns touched an odd number of times sit in buffer 1 now (own lane, no cross-lane access)
      {
        static_for<B::CORE>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          if constexpr ((Y::odd_mask() >> c) & 1u) Ps[c * ZS + z] = Ps[BUF + c * ZS + z];
        });
      }
      if constexpr (Y::barrier_before(0)) __syncthreads();
    }
    __syncthreads();

    // ---- hard decisions of the information columns (buffer 0), un-rotating each column (ldpc.py:1578-1581)
    if (active) {
      static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const uint32_t r4 = SPEC ? (uint32_t)kOne<BG, RA, ZI>.rho4[c] : (uint32_t)rho4[c];
        const uint32_t e4 = wrap4(4u * (uint32_t)z + r4, zc4);
        hard[(size_t)cb * K + c * zc + (e4 >> 2)] = Ps[c * ZS + z] < 0.0f ? 1 : 0;
      });
    }
    __syncthreads();
  }
}


__constant__ WrapTab kWrap1_384 = make_wrap<1, zindex_c(384)>();
__constant__ WrapTab kWrap1_352 = make_wrap<1, zindex_c(352)>();
__constant__ WrapTab kWrap2_256 = make_wrap<2, zindex_c(256)>();
// BG1, Zc = 384 with the first 13 / 15 / 16 / 22 / 31 rows only (15 = the metric configuration, with a one-deep
// prefetch ring: 11 extension layers)
__constant__ WrapTab kWrap1_384_r13 = make_wrap<1, zindex_c(384), 13>();
__constant__ WrapTab kWrap1_384_r15 = make_wrap<1, zindex_c(384), 15>();
__constant__ WrapTab kWrap1_384_r16 = make_wrap<1, zindex_c(384), 16>();
__constant__ WrapTab kWrap1_384_r22 = make_wrap<1, zindex_c(384), 22>();
__constant__ WrapTab kWrap1_384_r31 = make_wrap<1, zindex_c(384), 31>();

int zindex(int zc, int ils) {
  for (int i = 0; i < NZ; ++i)
    if (kZ.z[i] == zc && kZ.ils[i] == ils) return i;
  return -1;
}

}  // namespace nrx_dec2

// Called by nrx_ldpc_decode_f32 / nrx_ldpc_decode_rows_f32 (nrx_ldpc_dec.hip) for the (hard bits, K columns) case.
// n_rows < all rows: the caller guarantees that the dropped rows' extension columns are all-zero (see the header); the
// smallest built row count >= n_rows runs (extra rows are exact no-ops too).
int32_t nrx_ldpc_decode_fast_launch(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                    int32_t n_rows, uint8_t* hard, hipStream_t st) {
  using namespace nrx_dec2;
  const int zi = zindex(cfg->Zc, cfg->iLS);
  NRX_REQUIRE(zi >= 0, NRX_E_ARG, "nrx_ldpc_decode: (Zc=%d, iLS=%d) is not a lifting size", cfg->Zc, cfg->iLS);
  const int tz = ((cfg->Zc + 63) / 64) * 64;     // threads per code block (whole waves)
  const int ns = tz >= 320 ? 2 : 1;              // 5-/6-wave code blocks go two to a workgroup (see the kernel)
  const int threads = tz * ns;
  const int n_wg = (n_cb + ns - 1) / ns;
  const int grid = n_wg < 1024 ? n_wg : 1024;
  // device addresses of the constant tables: one set per device (a __constant__ symbol has one address per device),
  // resolved on first use on that device
  constexpr int MAX_DEV = 16;
  struct DevTabs { const FastTab* base[2]; const uint64_t* wrap[8]; bool wrap_ok; };
  static DevTabs tabs[MAX_DEV] = {};
  static std::mutex mu;
  int dev = 0;
  NRX_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEV, NRX_E_HIP, "nrx_ldpc_decode: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(mu);
  const FastTab** base = tabs[dev].base;
  const uint64_t** wrap = tabs[dev].wrap;
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
  static const bool all_rows = getenv("NRX_LDPC_ALLROWS") != nullptr;   // developer switch: ignore n_rows
  if (!tabs[dev].wrap_ok) {
    void* p[8] = {};
    const hipError_t e[8] = {hipGetSymbolAddress(&p[0], HIP_SYMBOL(kWrap1_384)), hipGetSymbolAddress(&p[1], HIP_SYMBOL(kWrap1_352)),
                             hipGetSymbolAddress(&p[2], HIP_SYMBOL(kWrap2_256)), hipGetSymbolAddress(&p[3], HIP_SYMBOL(kWrap1_384_r13)),
                             hipGetSymbolAddress(&p[4], HIP_SYMBOL(kWrap1_384_r16)), hipGetSymbolAddress(&p[5], HIP_SYMBOL(kWrap1_384_r22)),
                             hipGetSymbolAddress(&p[6], HIP_SYMBOL(kWrap1_384_r31)), hipGetSymbolAddress(&p[7], HIP_SYMBOL(kWrap1_384_r15))};
    for (int i = 0; i < 8; ++i)
      NRX_REQUIRE(e[i] == hipSuccess && p[i], NRX_E_HIP, "nrx_ldpc_decode: hipGetSymbolAddress(wrap masks) failed");
    for (int i = 7; i >= 0; --i) wrap[i] = (const uint64_t*)p[i];
    tabs[dev].wrap_ok = true;
  }
  const uint64_t* wt = nullptr;
#define NRX_DEC2_LAUNCH(BGN, ZIV, NSV, RAV)                                                                               \
  hipLaunchKernelGGL((ldpc_dec_fast_kernel<BGN, ZIV, NSV, RAV>), dim3(grid), dim3(threads), 0, st, llr, n_cb, cfg->Zc,     \
                     n_iter, hard, (ctab_t)t0, (ctab_t)t1, (ctab_t)rh, (mtab_t)wt)
  // lifting sizes with a specialised instantiation (the sizes of the BASELINE configurations); everything else
  // runs the generic kernel
  constexpr int ZI384 = zindex_c(384), ZI352 = zindex_c(352), ZI256 = zindex_c(256);
  if (all_rows) n_rows = cfg->bg == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  if (cfg->bg == 1) {
    if (!no_spec && zi == ZI384) {
      if (n_rows <= 13) { wt = wrap[3]; NRX_DEC2_LAUNCH(1, ZI384, 2, 13); }
      else if (n_rows <= 15) { wt = wrap[7]; NRX_DEC2_LAUNCH(1, ZI384, 2, 15); }
      else if (n_rows <= 16) { wt = wrap[4]; NRX_DEC2_LAUNCH(1, ZI384, 2, 16); }
      else if (n_rows <= 22) { wt = wrap[5]; NRX_DEC2_LAUNCH(1, ZI384, 2, 22); }
      else if (n_rows <= 31) { wt = wrap[6]; NRX_DEC2_LAUNCH(1, ZI384, 2, 31); }
      else { wt = wrap[0]; NRX_DEC2_LAUNCH(1, ZI384, 2, NRX_BG1_ROWS); }
    }
    else if (!no_spec && zi == ZI352) { wt = wrap[1]; NRX_DEC2_LAUNCH(1, ZI352, 2, NRX_BG1_ROWS); }
    else if (ns == 2) NRX_DEC2_LAUNCH(1, -1, 2, NRX_BG1_ROWS);
    else NRX_DEC2_LAUNCH(1, -1, 1, NRX_BG1_ROWS);
  } else {
    if (!no_spec && zi == ZI256) { wt = wrap[2]; NRX_DEC2_LAUNCH(2, ZI256, 1, NRX_BG2_ROWS); }
    else if (ns == 2) NRX_DEC2_LAUNCH(2, -1, 2, NRX_BG2_ROWS);
    else NRX_DEC2_LAUNCH(2, -1, 1, NRX_BG2_ROWS);
  }
#undef NRX_DEC2_LAUNCH
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_f32(fast)");
  return NRX_OK;
}
