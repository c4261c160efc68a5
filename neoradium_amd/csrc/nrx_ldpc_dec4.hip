// On-chip float64 layered min-sum LDPC decoder for ANY lifting size (hard decisions of the K information bits, bit-identical
// to the reference's float64 arithmetic, ldpc.py:1495-1581): the kernel of nrx_ldpc_dec3.hip with the lifting size at run time.
//
// nrx_ldpc_dec3.hip is specialised for Zc = 384 (every (column, shift) a DS immediate).  Everything else exact used to run the
// workspace kernel (nrx_ldpc_dec.hip), which streams the check-node state of every row through L2 and is 1.4-1.7x slower per
// row.  The structure that makes the on-chip kernel fast does not depend on the lifting size: which edges a layer has, which
// column is lane-private (Lay::sigma), where a column is handed over in a register (Lay::fwd1), where barriers go
// (Lay::plan_rot), the message representation (unit x pm under EXEC).  Only NUMBERS do -- the LDS offset of an edge's element,
// the lanes whose element index wraps, a layer's row rotation -- and those come from a per-lifting-size table built on the
// host (ZTab): an edge's address is  (wraps ? base - 8 Zc : base) + offset  with the offset in an SGPR, two VALU instructions
// where the specialised kernel has one.
//   * a workgroup is always 12 waves: NS = 12 / ceil(Zc / 64) code blocks side by side (2 at Zc > 320 ... 12 at Zc <= 64), LDS
//     NS x 26 x ZS x 8 B <= 156 KB (ZS = Zc rounded up to whole waves);
//   * lanes z >= Zc of a code block's last wave are masked off (they would address live elements: storage is in place at
//     (z + shift) mod Zc), so the EXEC-mask selects restore EXEC to the wave's valid lanes, not to -1.
// Rows: the first RA = 15 rows of the base graph (fewer: the rows beyond n_rows get zero extension LLRs and are exact
// no-ops, as in nrx_ldpc_dec3.hip); BG1 and BG2.
#include <stddef.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "nrx_ldpc_graph.h"
#include "nrx_common.h"

namespace nrx_dec4 {
using namespace nrx_ldpc;

struct ZTab {
  int32_t off[ESTRIDE];               // core edge e of the truncated graph: 8 * (col * ZS + eff_shift) bytes from a slot's column 0
  int32_t sig[64];                    // row rotation of layer L (Lay::sigma)
  uint32_t crcw[32];                  // (unused by the unfused entry; kept for a fused one)
  uint64_t wrap[ZMAX / 64][ESTRIDE];  // lanes of wave w (of a code block) with z + eff_shift >= Zc
};
typedef const ZTab __attribute__((address_space(4))) * ztab_t;
typedef const uint64_t __attribute__((address_space(4))) * mtab_t;
typedef const int32_t __attribute__((address_space(4))) * otab_t;

template <int BG, int RA> static void fill_table(ZTab& t, int zc, int ils) {
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  const int zs = (zc + 63) / 64 * 64;
  for (int L = 0; L < RA; ++L) {
    t.sig[L] = Y::sigma(ils, zc, L);
    for (int e = B::row_start(L); e < B::row_start(L + 1); ++e) {
      const int s = Y::eff_shift(ils, zc, L, e);
      t.off[e] = B::col(e) < B::CORE ? 8 * (B::col(e) * zs + s) : 0;
      for (int w = 0; w < ZMAX / 64; ++w) {
        uint64_t m = 0;
        for (int l = 0; l < 64; ++l)
          if (64 * w + l + s >= zc) m |= 1ull << l;
        t.wrap[w][e] = m;
      }
    }
  }
}

__device__ __forceinline__ double clip10(double x) {
  const double c = 1e10;
  return x < -c ? -c : (x > c ? c : x);
}
__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t dbl(uint32_t w) {
  uint32_t r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(w));
  return r;
}
// ---- messages as unit x pm, selected through EXEC: see nrx_ldpc_dec3.hip.  `vm` = the wave's valid lanes (EXEC around the block).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double unit_of(u32x2& uv, uint32_t signsrc) {
  uint32_t h;
  asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(h) : "v"(signsrc), "s"(0x80000000u));
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
__device__ __forceinline__ double c96_of(u32x2& uv, uint32_t signbits) {
  uint32_t h;
  asm("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(h) : "v"((uint32_t)__builtin_popcount(signbits)), "s"(0x40580000u));
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
__device__ __forceinline__ double vmin_abs(double a, double t) {
  double r;
  asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
__device__ __forceinline__ double vmax_abs(double a, double t) {
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(t));
  return r;
}
__device__ __forceinline__ double vmin_abs2(double a, double b) {
  double r;
  asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax_abs2(double a, double b) {
  double r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint64_t cmp_abs_eq(double t, double a) {
  uint64_t m;
  asm("v_cmp_eq_f64 %0, |%1|, %2" : "=s"(m) : "v"(t), "v"(a));
  return m;
}
template <int J>
__device__ __forceinline__ double sel_fnma_x(double x, uint32_t oidx, double u, double p1, double p2, uint64_t vm) {
  double y;
  asm("v_fma_f64 %[y], -%[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_u32_e32 vcc, %[j], %[oidx]\n\t"
      "v_fma_f64 %[y], -%[u], %[p2], %[x]\n\t"
      "s_mov_b64 exec, %[vm]"
      : [y] "=&v"(y) : [x] "v"(x), [j] "n"(J), [oidx] "v"(oidx), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2), [vm] "s"(vm) : "vcc");
  return y;
}
template <int J>
__device__ __forceinline__ double sel_fma_x(double x, uint32_t& idx, double a, double u, double p1, double p2, uint64_t vm) {
  double y;
  asm("v_fma_f64 %[y], %[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_f64_e64 vcc, |%[x]|, %[a]\n\t"
      "v_fma_f64 %[y], %[u], %[p2], %[x]\n\t"
      "v_mov_b32 %[idx], %[j]\n\t"
      "s_mov_b64 exec, %[vm]"
      : [y] "=&v"(y), [idx] "+v"(idx) : [x] "v"(x), [a] "v"(a), [j] "n"(J), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2), [vm] "s"(vm) : "vcc");
  return y;
}
template <int J>
__device__ __forceinline__ void sel_fma_first(double& x, uint32_t& idx, uint64_t& seen, uint64_t im, double u, double p1, double p2,
                                              uint64_t vm) {
  asm volatile("s_andn2_b64 exec, %[im], %[seen]\n\t"
               "s_or_b64 %[seen], %[seen], %[im]\n\t"
               "v_fma_f64 %[x], %[u], %[p2], %[x]\n\t"
               "v_mov_b32 %[idx], %[j]\n\t"
               "s_andn2_b64 exec, %[vm], exec\n\t"
               "v_fma_f64 %[x], %[u], %[p1], %[x]\n\t"
               "s_mov_b64 exec, %[vm]"
               : [x] "+v"(x), [idx] "+v"(idx), [seen] "+s"(seen) : [im] "s"(im), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2), [j] "n"(J), [vm] "s"(vm) : "scc");
}
#define LAYER_PRIO(Q) __builtin_amdgcn_s_setprio(3 - (Q))

// RC = rows whose check-node state stays in registers (default: all RA of them).  RC < RA: the HYBRID of nrx_ldpc_dec3.hip at a
// run-time lifting size -- the rows >= RC (the sparse ones) keep pm1 / pm2 / extension posterior in the caller's workspace,
// [workgroup][slot][row - RC][3][ZS] doubles, read PF streamed layers ahead into a register ring (slot (L - RC) mod PF, hence
// RA - RC a multiple of PF) and written back when the layer has them; sign / argmin words of all rows stay in registers.
template <int BG, int RA, int RC = RA>
__global__ void __launch_bounds__(768, 3)
ldpc_dec_chipz_kernel(const double* __restrict__ llr, int n_cb, int n_iter, uint8_t* __restrict__ hard, ztab_t tab, int zc,
                      int rows_live, double* __restrict__ ws) {
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  static_assert(Y::plan_rot.ok, "barrier placement leaves a column hazard");
  constexpr bool HYB = RC < RA;
  // layer L of a launch that needs the first n rows (as nrx_ldpc_dec3.hip's MODE bit 3): a layer that takes column 1 from its
  // predecessor's register runs as long as the predecessor does (the pair is left out together: column 1 then stays in LDS); the last
  // layer is kept when it hands column 1 to layer 0
  auto runs_at = [](int L, int n) constexpr -> bool {
    if (!Y::has_ext(L)) return true;
    if (Y::fwd1(L)) return L <= n;
    if (Y::give1(L) && L == B::ROWS - 1) return true;
    return L < n;
  };
  constexpr int PF = 2;                                    // streamed layers fetched ahead
  static_assert(!HYB || (RC >= 4 && RA - RC > PF && (RA - RC) % PF == 0), "the ring slot of a streamed layer is (L - RC) mod PF in every iteration");
  extern __shared__ __attribute__((aligned(16))) double Praw[];     // [ZS padding][NS][CORE][ZS]
  const int ZS = (zc + 63) & ~63;
  const int NS = 12 / (ZS >> 6);
  const int BUF = B::CORE * ZS;
  const int slot = __builtin_amdgcn_readfirstlane((int)threadIdx.x / ZS);     // wave-uniform: slots are whole waves
  const int z = (int)threadIdx.x - slot * ZS;
  const uint32_t sb = (uint32_t)(ZS + slot * BUF) * 8u;
  const int wv = __builtin_amdgcn_readfirstlane(z >> 6);
  const mtab_t wm = (mtab_t)&tab->wrap[0][0] + wv * ESTRIDE;
  const otab_t ot = (otab_t)&tab->off[0];
  const otab_t st = (otab_t)&tab->sig[0];
  const uint32_t zb0 = 8u * (uint32_t)z + sb;
  double* const Ps = Praw + ZS + slot * BUF;
  const uint32_t zc8 = 8u * (uint32_t)zc;
  const int N = (B::COLS - 2) * zc, K = B::KB * zc;
  const bool lane_ok = z < zc;
  constexpr int NEXT = Y::n_ext() > 0 ? Y::n_ext() : 1;

  double m1[RC], m2[RC];
  double rext[NEXT];
  double pf_m1[PF] = {}, pf_m2[PF] = {}, pf_rx[PF] = {};
  typedef double __attribute__((address_space(1))) * gD;
  const uint32_t zs8 = 8u * (uint32_t)ZS;
  char* wsb = nullptr;                                     // this slot's streamed state
  if constexpr (HYB) wsb = (char*)ws + ((size_t)blockIdx.x * NS + slot) * (size_t)(RA - RC) * 3 * zs8;
  // (uniform base + uniform offset) made an opaque SGPR pair first: `global_load/store v, v_lane_offset, s[base]`
  auto wsL = [&](int L, int a, uint32_t lane_off) __attribute__((always_inline)) -> gD {
    char* b = wsb + (size_t)((L - RC) * 3 + a) * zs8;
    asm volatile("" : "+s"(b));
    return (gD)(b + (size_t)lane_off);
  };
  uint32_t sgw[Y::n_wide() > 0 ? Y::n_wide() : 1];
  uint32_t sgn[(Y::n_narrow() + 1) / 2];
  u32x2 uv = {0u, 0u};
  double c0 = 0.0, f1 = 0.0;

  for (int cb0 = blockIdx.x * NS; cb0 < n_cb; cb0 += gridDim.x * NS) {
    const int cb = cb0 + slot;
    int one = 1;
    asm volatile("" : "+s"(one));
    const bool live = cb < n_cb && one != 0;
    const double* in = llr + (size_t)(live ? cb : n_cb - 1) * N;
    int zl0 = z < zc ? z : 0;                                // (masked lanes load an existing element)
    asm volatile("" : "+v"(zl0));
    auto fetch = [&](int p0, int rot) __attribute__((always_inline)) -> double {
      int zr = zl0 + rot;
      zr -= zr >= zc ? zc : 0;
      return clip10(in[p0 + zr]) + 0.0;
    };
    int zq = z;                                              // (opaque per code block, as zl0: keeps the 26 column addresses
    asm volatile("" : "+v"(zq));                             //  from being hoisted in front of the code-block loop and spilled)
    if (lane_ok) {
      static_for<B::CORE>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c == 1) Ps[c * ZS + zq] = 0.0;
        else if constexpr (c >= 2) Ps[c * ZS + zq] = fetch((c - 2) * zc, 0);
      });
    }
    static_for<RC>([&](auto lc) __attribute__((always_inline)) {
      constexpr int L = decltype(lc)::value;
      m1[L] = 0.0;
      m2[L] = 0.0;
      if constexpr (Y::has_ext(L)) {
        rext[Y::ext_idx(L)] = fetch((Y::ext_col(L) - 2) * zc, st[L]);
        rext[Y::ext_idx(L)] = L < rows_live ? rext[Y::ext_idx(L)] : 0.0;
      }
    });
    if constexpr (HYB) {       // streamed rows: zero minima, the extension LLR (or 0 beyond the rows that run) straight to the workspace
      const uint32_t zo8 = 8u * (uint32_t)zq;
      static_for<RA - RC>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = RC + decltype(lc)::value;
        *wsL(L, 0, zo8) = 0.0;
        *wsL(L, 1, zo8) = 0.0;
        if constexpr (Y::has_ext(L)) {
          const double e = fetch((Y::ext_col(L) - 2) * zc, st[L]);
          *wsL(L, 2, zo8) = L < rows_live ? e : 0.0;
        }
      });
    }
    c0 = 0.0;
    static_for<(Y::n_wide() > 0 ? Y::n_wide() : 1)>([&](auto i) __attribute__((always_inline)) { sgw[decltype(i)::value] = 0u; });
    static_for<(Y::n_narrow() + 1) / 2>([&](auto i) __attribute__((always_inline)) { sgn[decltype(i)::value] = 0u; });
    __syncthreads();
    if constexpr (HYB) {      // (the barrier above waited for the fill's stores)
      int zr0 = z;
      asm volatile("" : "+v"(zr0));
      static_for<PF>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, Lk = RC + k;
        pf_m1[k] = *wsL(Lk, 0, 8u * (uint32_t)zr0);
        pf_m2[k] = *wsL(Lk, 1, 8u * (uint32_t)zr0);
        if constexpr (Y::has_ext(Lk)) pf_rx[k] = *wsL(Lk, 2, 8u * (uint32_t)zr0);
      });
    }

    // wrap masks and offsets of the layer about to run (SGPRs)
    uint64_t wcur[19];
    int32_t ocur[19];
    static_for<(Y::has_ext(0) ? Y::deg(0) - 1 : Y::deg(0))>([&](auto jc) __attribute__((always_inline)) {
      wcur[decltype(jc)::value] = wm[B::row_start(0) + decltype(jc)::value];
      ocur[decltype(jc)::value] = ot[B::row_start(0) + decltype(jc)::value];
    });

    for (int it = 0; it < n_iter; ++it) {
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = B::row_start(L);
        constexpr int D = Y::deg(L);
        constexpr bool EXT = Y::has_ext(L);
        constexpr int DC = EXT ? D - 1 : D;
        constexpr bool WIDE = Y::wide(L);
        uint32_t zbo = zb0;
        uint32_t wo = 0;
        asm volatile("" : "+v"(zbo), "+s"(wo));
        const int wo_u = __builtin_amdgcn_readfirstlane(wo);          // opaque zero: keeps the layer's table loads inside the layer
        const mtab_t wml = (mtab_t)((const char __attribute__((address_space(4)))*)wm + wo_u);
        const otab_t otl = (otab_t)((const char __attribute__((address_space(4)))*)ot + wo_u);
        const uint32_t zb = zbo, zbw = zbo - zc8;
        // hybrid, streamed layer: its state out of the ring, and the ring slot refilled with the streamed layer PF ahead (of the
        // next iteration at the end of this one).  Outside the `live` branch: inside it the values become phis at the join.
        double cm1 = 0.0, cm2 = 0.0, crx = 0.0;
        uint32_t zo8 = 0;
        if constexpr (HYB && L >= RC) {
          constexpr int k = (L - RC) % PF;
          constexpr int Lp = L + PF < RA ? L + PF : RC + (L + PF - RA);
          zo8 = zbo - sb;                                   // 8 * z
          cm1 = pf_m1[k];
          cm2 = pf_m2[k];
          crx = pf_rx[k];
          if (runs_at(Lp, rows_live)) {                     // (kernel-uniform: a layer that is left out is not fetched either)
            pf_m1[k] = *wsL(Lp, 0, zo8);
            pf_m2[k] = *wsL(Lp, 1, zo8);
            if constexpr (Y::has_ext(Lp)) pf_rx[k] = *wsL(Lp, 2, zo8);
          }
        }
        // A layer beyond the caller's row count has all-zero extension LLRs: every one of its rows is the exact no-op of DESIGN 4.2a and
        // the layer is left out (kernel-uniform test; its barrier and the next layer's table loads stay).
        const bool runs = runs_at(L, rows_live);
        if (__builtin_expect(live, 1)) {
          if (lane_ok && runs) {
            // the wave's valid lanes.  (Read by an asm the compiler cannot see through: as ballot(true) it is a COPY of exec,
            //  which copy propagation hands to the blocks below as their operand -- "s_mov_b64 exec, exec" restores nothing.)
            uint64_t vm;
            asm volatile("s_mov_b64 %0, exec" : "=s"(vm));
            double t[D];
            constexpr int PQ1 = (7 * D) / 10 < D - 1 ? (7 * D) / 10 : D - 1, PQ2 = D / 2 < 2 ? 2 : D / 2, PQ3 = (3 * D) / 10 < 1 ? 1 : (3 * D) / 10;
            LAYER_PRIO(Y::prio_q(L, 0));
            // ---- pass 1a: the LDS reads of the layer (column 0: the lane's register; a handed-over column 1: the predecessor's)
            static_for<DC>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              constexpr int col = B::col(E0 + j);
              if constexpr (col == 0) {
                t[j] = c0;
              } else if constexpr (col == 1 && Y::fwd1(L)) {
                t[j] = f1;
              } else {
                const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);
                t[j] = *(const double*)((const char*)Praw + ((wraps ? zbw : zb) + (uint32_t)ocur[j]));
              }
            });
            __builtin_amdgcn_sched_barrier(0);
            double om1, om2;
            if constexpr (L < RC) { om1 = m1[L]; om2 = m2[L]; } else { om1 = cm1; om2 = cm2; }
            uint32_t word, oidx;
            int top;
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
            // ---- pass 1b: t_j = fma(-u_j, argmin ? pm2 : pm1, r_j)  (ldpc.py:1550-1553)
            if constexpr (EXT) t[D - 1] = L < RC ? rext[Y::ext_idx(L)] : crx;
            uint32_t wrun = word << (top - (D - 1));
            static_for<D>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              const double u = unit_of(uv, wrun);
              if constexpr (j < D - 1) wrun = dbl(wrun);
              if constexpr (j == PQ1 && Y::prio_q(L, 1) != Y::prio_q(L, 0)) LAYER_PRIO(Y::prio_q(L, 1));
              t[j] = sel_fnma_x<j>(t[j], oidx, u, om1, om2, vm);
            });
            // ---- min-sum (ldpc.py:1556-1564)
            static_assert(D >= 2, "a check row has at least two edges");
            double a1 = vmin_abs2(t[0], t[1]);
            double a2 = vmax_abs2(t[0], t[1]);
            uint32_t nsg = __builtin_amdgcn_alignbit(hi32(t[0]) >> 31, hi32(t[1]), 31);
            static_for<D - 2>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value + 2;
              if constexpr (j == PQ2 && Y::prio_q(L, 2) != Y::prio_q(L, 1)) LAYER_PRIO(Y::prio_q(L, 2));
              a2 = vmin(a2, vmax_abs(a1, t[j]));
              a1 = vmin_abs(a1, t[j]);
              nsg = __builtin_amdgcn_alignbit(nsg, hi32(t[j]), 31);
            });
            // QUIRK ldpc.py:1563 (+1e5 on the signed argmin entry): wave-uniform cold path
            bool tie_quirk = false;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(a2 > 5.0e4) != 0, 0)) {
              double v = t[D - 1];
              static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = D - 2 - decltype(jc)::value;
                v = (__double_as_longlong(t[j]) & 0x7fffffffffffffffll) == __double_as_longlong(a1) ? t[j] : v;
              });
              const double q = __builtin_fabs(v + 100000.0);
              tie_quirk = __builtin_amdgcn_ballot_w64(q < a2 && a2 == a1) != 0;
              a2 = q < a2 ? q : a2;
            }
            const double c96 = c96_of(uv, nsg);
            const double nm1 = a1 * c96, nm2 = a2 * c96;
            if constexpr (L < RC) {
              m1[L] = nm1;
              m2[L] = nm2;
            } else {
              *wsL(L, 0, zo8) = nm1;
              *wsL(L, 1, zo8) = nm2;
            }
            // ---- pass 2: r_j = fma(u_j, first argmin ? pm2 : pm1, t_j), back to the element it was read from
            uint32_t idx = 0;
            auto put = [&](auto jc2) __attribute__((always_inline)) {
              constexpr int j = decltype(jc2)::value;
              constexpr int col = B::col(E0 + j);
              if constexpr (col == 0) {
                c0 = t[j];
              } else if constexpr (col == 1 && Y::give1(L)) {
                f1 = t[j];
              } else if constexpr (col < B::CORE) {
                const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);
                *(double*)((char*)Praw + ((wraps ? zbw : zb) + (uint32_t)ocur[j])) = t[j];
              } else {
                if constexpr (L < RC) rext[Y::ext_idx(L)] = t[j];
                else *wsL(L, 2, zo8) = t[j];
              }
            };
            if (__builtin_expect(!tie_quirk, 1)) {
              static_for<D>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = D - 1 - decltype(jc)::value;
                if constexpr (decltype(jc)::value == PQ3 && Y::prio_q(L, 3) != Y::prio_q(L, 2)) LAYER_PRIO(Y::prio_q(L, 3));
                const double u = unit_of(uv, hi32(t[j]));
                t[j] = sel_fma_x<j>(t[j], idx, a1, u, nm1, nm2, vm);
                put(std::integral_constant<int, j>{});
              });
            } else {
              uint64_t seen = 0;
              static_for<D>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const uint64_t is_min = cmp_abs_eq(t[j], a1);
                const double u = unit_of(uv, hi32(t[j]));
                sel_fma_first<j>(t[j], idx, seen, is_min, u, nm1, nm2, vm);
                put(std::integral_constant<int, j>{});
              });
            }
            if constexpr (WIDE) {
              sgw[Y::wide_idx(L)] = idx | (nsg << 5);
            } else {
              constexpr int ni = Y::narrow_idx(L);
              const uint32_t f = idx | (nsg << 4);
              if constexpr (ni & 1) sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x05040100u);
              else sgn[ni / 2] = __builtin_amdgcn_perm(f, word, 0x03020504u);
            }
          }
          // masks and offsets of the next layer (wave-uniform: outside the lane branch)
          __builtin_amdgcn_sched_barrier(0);
          constexpr int Ln = (L + 1) % B::ROWS;
          constexpr int DCn = Y::has_ext(Ln) ? Y::deg(Ln) - 1 : Y::deg(Ln);
          static_for<DCn>([&](auto jc) __attribute__((always_inline)) {
            wcur[decltype(jc)::value] = wml[B::row_start(Ln) + decltype(jc)::value];
            ocur[decltype(jc)::value] = otl[B::row_start(Ln) + decltype(jc)::value];
          });
        }
        if constexpr (Y::plan_rot.need[(L + 1) % B::ROWS]) __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    __syncthreads();
    // ---- hard decisions of the information columns (ldpc.py:1578-1581)
    if (live && lane_ok) {
      int zh = z;
      asm volatile("" : "+v"(zh));
      static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c == 0) hard[(size_t)cb * K + zh] = c0 < 0.0 ? 1 : 0;
        else hard[(size_t)cb * K + c * zc + zh] = Ps[c * ZS + zh] < 0.0 ? 1 : 0;
      });
    }
    __syncthreads();
  }
}

// device copy of the table of a (base graph, lifting size), built on first use, per device
struct TabKey { int dev, bg, zc, ra; };
struct TabEnt { TabKey k; ZTab* d; };
static std::vector<TabEnt> g_tabs;
static std::mutex g_mu;

template <int BG, int RA> static int32_t get_table(int zc, int ils, hipStream_t st, const ZTab** out) {
  int dev = 0;
  NRX_REQUIRE(hipGetDevice(&dev) == hipSuccess, NRX_E_HIP, "nrx_ldpc_decode_f64: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(g_mu);
  for (const TabEnt& e : g_tabs)
    if (e.k.dev == dev && e.k.bg == BG && e.k.zc == zc && e.k.ra == RA) { *out = e.d; return NRX_OK; }
  ZTab* h = new ZTab();
  fill_table<BG, RA>(*h, zc, ils);
  ZTab* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(ZTab)) != hipSuccess) { delete h; NRX_REQUIRE(false, NRX_E_HIP, "nrx_ldpc_decode_f64: hipMalloc(table) failed"); }
  // (synchronous copy on first use of a lifting size: the table must be complete before the launch that follows on `st`)
  const hipError_t e = hipMemcpy(d, h, sizeof(ZTab), hipMemcpyHostToDevice);
  delete h;
  if (e != hipSuccess) { (void)hipFree(d); NRX_REQUIRE(false, NRX_E_HIP, "nrx_ldpc_decode_f64: table upload failed"); }
  g_tabs.push_back({{dev, BG, zc, RA}, d});
  *out = d;
  return NRX_OK;
}

template <int BG, int RA, int RC = RA>
static int32_t launch(const double* llr, int n_cb, const nrx_ldpc_cfg* cfg, int n_iter, int n_rows, uint8_t* hard, hipStream_t st,
                      void* ws = nullptr, size_t ws_bytes = 0) {
  const int zs = (cfg->Zc + 63) / 64 * 64, ns = 12 / (zs / 64);
  const size_t lds = sizeof(double) * ((size_t)zs + (size_t)ns * G<BG>::CORE * zs);
  const int n_wg = (n_cb + ns - 1) / ns;
  int grid = n_wg < 1024 ? n_wg : 1024;
  if constexpr (RC < RA) {      // hybrid: one workgroup per CU is all that fits; its streamed state lives in the caller's workspace
    const size_t per_wg = sizeof(double) * (size_t)(RA - RC) * 3 * (size_t)ns * zs;
    if (grid > 256) grid = 256;
    if (ws == nullptr || ws_bytes < per_wg) return 1;            // no workspace: the workspace kernel reports it
    if ((size_t)grid > ws_bytes / per_wg) grid = (int)(ws_bytes / per_wg);
  }
  const ZTab* tab = nullptr;
  const int32_t rc = get_table<BG, RA>(cfg->Zc, cfg->iLS, st, &tab);
  if (rc) return rc;
  auto kern = ldpc_dec_chipz_kernel<BG, RA, RC>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(ns * zs), lds, st, llr, n_cb, n_iter, hard, (ztab_t)tab, cfg->Zc, n_rows, (double*)ws);
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_f64(on-chip, any lifting size)");
  return NRX_OK;
}

}  // namespace nrx_dec4

// Called by nrx_ldpc_decode_rows_f64 (nrx_ldpc_dec.hip) after nrx_ldpc_dec3.hip's Zc = 384 instantiations declined: hard decisions
// of the K information bits, any lifting size.  <= 15 rows: everything on chip.  More rows: the hybrid instantiations (BG1: the
// first 31 or all 46 rows, BG2: the first 22 or all 42; the rows beyond n_rows run as exact no-ops on zeroed extension LLRs) with
// the sparse rows' state in the caller's workspace.  Returns 1 when not covered: the caller runs the workspace kernel.
int32_t nrx_ldpc_decode_chipz_launch(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                     uint8_t* hard, hipStream_t st, void* ws, size_t ws_bytes) {
  using namespace nrx_dec4;
  if (getenv("NRX_LDPC_NOCHIP64") != nullptr || n_rows < 4 || cfg->Zc < 2 || cfg->Zc > ZMAX) return 1;
  constexpr int RCH = 10;                 // resident rows of the hybrids
  if (n_rows > 15) {
    if (getenv("NRX_LDPC_NOHYBRID") != nullptr) return 1;
    if (cfg->bg == 1) {
      if (n_rows <= 31) return launch<1, 31, RCH + 1>(llr, n_cb, cfg, n_iter, n_rows, hard, st, ws, ws_bytes);
      return launch<1, 46, RCH>(llr, n_cb, cfg, n_iter, n_rows, hard, st, ws, ws_bytes);
    }
    if (cfg->bg == 2) {
      if (n_rows <= 22) return launch<2, 22, RCH>(llr, n_cb, cfg, n_iter, n_rows, hard, st, ws, ws_bytes);
      return launch<2, 42, RCH>(llr, n_cb, cfg, n_iter, n_rows, hard, st, ws, ws_bytes);
    }
    return 1;
  }
  if (cfg->bg == 1) return launch<1, 15>(llr, n_cb, cfg, n_iter, n_rows, hard, st);
  if (cfg->bg == 2) return launch<2, 15>(llr, n_cb, cfg, n_iter, n_rows, hard, st);
  return 1;
}
