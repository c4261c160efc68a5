// The stability certificate evaluated INSIDE the decoder's stage kernel, on the state the workgroup still holds:
// posterior columns in LDS, check-node state in registers, twelve waves per CU.  The only thing that does not fit on chip is the
// per-element slack sums W (38 KB per code block: LDS is full) -- they live in an L2-resident scratch of the launch.
// Same conditions, same relaxation as ldpc_certify_kernel (nrx_ldpc_cert.hip), which stays as the stand-alone form.
#pragma once
#include <type_traits>
#include "nrx_ldpc_graph.h"

namespace nrx_certcore {
using namespace nrx_ldpc;

// The slack sums W_c (float32) are kept by round-to-nearest updates W <- fl(W + fl(w_new - w_old)): each update can leave W below the
// true sum of the (exactly stored, float32) slacks by at most 2^-23 of the new sum (two roundings, both operands <= the sum; slacks
// only grow, so the sum never shrinks).  An element of W is updated at most once per row and sweep: <= 46 rows x 16 sweeps (the
// entries clamp max_sweeps to 16) = 736 updates, so  sum_j w_j <= W (1 + 736 x 2^-23)(1 + 2^-24 for the initial product zeta x degree)
// < W (1 + 2^-13.4); the tuned kernels (15 rows, 4 sweeps by default) stay below W (1 + 2^-17).  Every USE of a sum therefore
// takes W x WINFL, WINFL = 1 + 2^-13: the loss W - w + 2E of (S) and (M) and the posterior test rho >= W + G are evaluated on an
// upper bound of the true sums -- what the proof ("Arithmetic of the certificate itself") needs.  The fused multiply-add
// that forms tau - W x WINFL rounds once (relative 2^-53 of a quantity <= beta: inside E's 1.0625).
constexpr double WINFL = 1.0 + 0x1p-13;

struct Params {
  double gamma, gamma1;      // a-priori magnitude bounds per unit of the LLR maxima (nrx_ldpc_cert_bounds)
  int32_t dmax, n_iter_total, max_sweeps, flags, iter_now;
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hi32(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ double unit_of(u32x2& uv, uint32_t signsrc) {
  uint32_t h;
  asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(h) : "v"(signsrc), "s"(0x80000000u));
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
template <int J>
__device__ __forceinline__ double sel_fnma_x(double x, uint32_t oidx, double u, double p1, double p2) {
  double y;
  asm("v_fma_f64 %[y], -%[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_u32_e32 vcc, %[j], %[oidx]\n\t"
      "v_fma_f64 %[y], -%[u], %[p2], %[x]\n\t"
      "s_mov_b64 exec, -1"
      : [y] "=&v"(y) : [x] "v"(x), [j] "n"(J), [oidx] "v"(oidx), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2) : "vcc");
  return y;
}
// x - (oidx == J ? k2 : k1); the lanes of the old argmin also copy the result into tlo (EXEC instead of five selects)
template <int J>
__device__ __forceinline__ double sel_sub_x(double x, uint32_t oidx, double k1, double k2, double& tlo) {
  double y;
  asm("v_add_f64 %[y], %[x], -%[k1]\n\t"
      "v_cmpx_eq_u32_e32 vcc, %[j], %[oidx]\n\t"
      "v_add_f64 %[y], %[x], -%[k2]\n\t"
      "v_mov_b64 %[tlo], %[y]\n\t"
      "s_mov_b64 exec, -1"
      : [y] "=&v"(y), [tlo] "+v"(tlo) : [x] "v"(x), [j] "n"(J), [oidx] "v"(oidx), [k1] "v"(k1), [k2] "v"(k2) : "vcc");
  return y;
}
__device__ __forceinline__ double flip_by(double x, double s) {
  u32x2 v = __builtin_bit_cast(u32x2, x);
  v.y ^= hi32(s) & 0x80000000u;
  return __builtin_bit_cast(double, v);
}
__device__ __forceinline__ double fmin2(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmax2(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double uniform(double x) {
  u32x2 v = __builtin_bit_cast(u32x2, x);
  v.x = __builtin_amdgcn_readfirstlane(v.x);
  v.y = __builtin_amdgcn_readfirstlane(v.y);
  return __builtin_bit_cast(double, v);
}
template <int BG, int RA> constexpr int col_degree(int c) {
  int n = 0;
  for (int e = 0; e < GR<BG, RA>::EDGES; ++e) n += GR<BG, RA>::col(e) == c ? 1 : 0;
  return n;
}
// slack sums in the launch's scratch.  Only the waves of ONE workgroup exchange them, between two workgroup barriers with a
// workgroup-scope release fence in front: they share the CU's vector L1, so workgroup scope is enough (agent scope sent every access
// past the L2 of the XCD: 2.5 us per row step).
__device__ __forceinline__ float wload(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wstore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Called by EVERY wave of the workgroup (it contains workgroup barriers); `want` = this slot's block is a candidate (wave-uniform per
// slot).  The check-node state is read from `st` = this lane's fields of the block's PARKED state (StateLay; the kernel parks before it
// calls this, each lane reading back what it wrote itself): with the state in registers the certificate's own registers did not fit
// beside it and the allocator's spills landed INSIDE the decoder's iteration loop (72 scratch accesses per iteration).
// Ps = the slot's posterior columns in LDS (column c at Ps[c * ZC], element order), c0 = element z of column 0, Wg = the
// slot's scratch: CORE * ZC floats of slack sums, then 2 * ROWS * ZC floats for the rows' own slacks (lane order: they do not fit the
// register file beside the decoder's state -- kept in registers they pushed spills INTO the decoder's iteration loop), flags = 6 LDS
// words per workgroup.  Returns whether the slot's block holds the certificate.
typedef const double __attribute__((address_space(3))) * lds_cd;
// NS = the slots that take part in `bar` together: 2 = the workgroup's two blocks behind workgroup barriers (the stage kernels), 1 = this slot
// alone behind its own barrier (the persistent schedule: the other slot is somewhere else in its own block).
template <int BG, int ZI, int RA, bool HASF, int NS, class BarF>
__device__ __forceinline__ bool certify_on_chip(const Params& cp, double lam_all, double lam_pe, const double* st, double c0, lds_cd Ps,
                                                float* Wg, int z, int slot, bool want, uint32_t* flags, BarF&& bar) {
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  using SL = StateLay<BG, RA>;
  constexpr int ZC = kZ.z[ZI];
  constexpr int ILS = kZ.ils[ZI];
  const double beta = cp.gamma * lam_all;
  const double E_ = 2.0 * cp.n_iter_total * cp.dmax * 0x1p-53 * beta * 1.0625;
  const double rmin = (cp.gamma1 > 1.0 ? cp.gamma1 : 1.0) * lam_pe * (1 + 1e-9) + E_;
  const double mcap_ = 0.75 * (1.0e5 - rmin) * (1 - 1e-9) - E_;
  const double mcapx_ = 0.75 * (1.0e5 - (lam_pe * (1 + 1e-9) + E_)) * (1 - 1e-9) - E_;
  const double E = uniform(E_), zeta = uniform(4.0 * E_), G = uniform(4.0 * E_);
  const double mcap = uniform(mcap_), mcapx = uniform(mcapx_);
  const bool bounds_ok = beta < 2.5e8 && lam_all < 1e9 && mcap > 0.0 && beta == beta && mcap == mcap;
  bool active = want && bounds_ok;
  bool certified = false;
  const float zf = (float)zeta;
  float* const wrow = Wg + B::CORE * ZC + z;      // this lane's slacks: row L at wrow[(2 L) * ZC], wrow[(2 L + 1) * ZC]
  float W0 = zf * col_degree<BG, RA>(0);
  if (active) {
    static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
      wrow[(2 * decltype(lc)::value) * ZC] = zf;
      wrow[(2 * decltype(lc)::value + 1) * ZC] = zf;
    });
    static_for<B::CORE - 1>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value + 1;
      constexpr int DEGC = col_degree<BG, RA>(c);
      wstore(Wg + c * ZC + z, zf * DEGC);
    });
  }
  u32x2 uv = {0u, 0u};
  bool dead = false;
  double n_pm1 = 0.0, n_pm2 = 0.0, n_rx = 0.0;
  uint32_t n_word = 0;
  float n_fw1 = 0.f, n_fw2 = 0.f;
  auto fetch_row = [&](auto lc) __attribute__((always_inline)) {      // row L's check-node state (parked) and its two slacks
    constexpr int L = decltype(lc)::value;
    const double* sq = st;
    asm volatile("" : "+v"(sq));
    n_pm1 = (*(&sq[(size_t)(SL::M1 + L) * ZC]));
    n_pm2 = (*(&sq[(size_t)(SL::M2 + L) * ZC]));
    if constexpr (Y::has_ext(L)) n_rx = sq[(size_t)(SL::REXT + Y::ext_idx(L)) * ZC];
    if constexpr (Y::wide(L)) n_word = (uint32_t)__double_as_longlong(sq[(size_t)(SL::WORDS + Y::wide_idx(L)) * ZC]);
    else n_word = (uint32_t)__double_as_longlong(sq[(size_t)(SL::WORDS + SL::NW + Y::narrow_idx(L) / 2) * ZC]);
    n_fw1 = wrow[(2 * L) * ZC];
    n_fw2 = wrow[(2 * L + 1) * ZC];
  };
  if (active) fetch_row(std::integral_constant<int, 0>{});
  bar();
  for (int sweep = 0; sweep < cp.max_sweeps; ++sweep) {
    bool raised = false;
    if (z == 0) { flags[2 * slot] = 0u; flags[2 * slot + 1] = 0u; }
    static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
      constexpr int L = decltype(lc)::value;
      constexpr int E0 = B::row_start(L);
      constexpr int D = Y::deg(L);
      constexpr bool EXT = Y::has_ext(L);
      constexpr int DC = EXT ? D - 1 : D;
      constexpr bool WIDE = Y::wide(L);
      if (active) {
        int zq = z;
        asm volatile("" : "+v"(zq));
        uint32_t field, oidx;
        constexpr int SB = WIDE ? 5 : 4;
        // this row's state and slacks were fetched while the previous row ran (they do not depend on the slack sums)
        double pm1 = n_pm1, pm2 = n_pm2;
        const double rx = n_rx;
        if constexpr (WIDE) {
          field = n_word;
          oidx = field & 31u;
        } else {
          constexpr int ni = Y::narrow_idx(L);
          field = (ni & 1) ? (n_word >> 16) : (n_word & 0xffffu);
          oidx = field & 15u;
        }
        const float fw1 = n_fw1, fw2 = n_fw2;
        float W[DC];
        static_for<DC>([&](auto jc) __attribute__((always_inline)) {      // the slack sums first: they come from L2
          constexpr int j = decltype(jc)::value;
          constexpr int col = B::col(E0 + j);
          if constexpr (col == 0) {
            W[j] = W0;
          } else {
            int el = zq + Y::eff_shift(ILS, ZC, L, E0 + j);
            el -= el >= ZC ? ZC : 0;
            W[j] = wload(Wg + col * ZC + el);
          }
        });
        const double ow1 = (double)fw1, ow2 = (double)fw2;
        const double k1 = 2.0 * E - ow1, k2 = 2.0 * E - ow2;
        uint32_t wrun = field << (31 - (SB + D - 1));
        uint32_t srw = 0, hgw = 0;
        double a1 = 1.0e300, a2 = 1.0e300, tlo = 1.0e300, pmin = 1.0e300;
        static_for<D>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          constexpr int col = B::col(E0 + j);
          double rj;
          if constexpr (col == 0) {
            rj = c0;
          } else if constexpr (col < B::CORE) {
            int el = zq + Y::eff_shift(ILS, ZC, L, E0 + j);
            el -= el >= ZC ? ZC : 0;
            rj = Ps[col * ZC + el];
          } else {
            rj = rx;
          }
          const double u = unit_of(uv, wrun);
          if constexpr (j < D - 1) wrun += wrun;
          const double t = sel_fnma_x<j>(rj, oidx, u, pm1, pm2);
          const double tau = flip_by(t, rj);
          srw = __builtin_amdgcn_alignbit(srw, hi32(rj), 31);
          double tl;
          if constexpr (col < B::CORE) {
            // the slack sums are float32, kept by round-to-nearest updates: W * WINFL is an upper bound of the true sum (see WINFL)
            const double x = __builtin_fma(-(double)W[j], WINFL, tau);
            tl = sel_sub_x<j>(x, oidx, k1, k2, tlo);      // x - (j == oidx ? k2 : k1), and tlo <- that where j == oidx
            double pr = __builtin_fma(-(double)W[j], WINFL, __builtin_fabs(rj));
            if constexpr (HASF) {
              const bool hg = __builtin_fabs(rj) >= 5.0e8;
              hgw = (hgw << 1) | (hg ? 1u : 0u);
              tl = hg ? 1.0e300 : tl;
              pr = hg ? 1.0e300 : pr;
            }
            pmin = fmin2(pmin, pr);
          } else {
            tl = tau - 2.0 * E;
            if constexpr (HASF) hgw <<= 1;
          }
          a2 = fmin2(a2, fmax2(a1, tl));
          a1 = fmin2(a1, tl);
        });
        const double am1 = __builtin_fabs(pm1) * 0x1p-7, am2 = __builtin_fabs(pm2) * 0x1p-7;
        const bool bad = (__builtin_popcount(srw) & 1) != 0 || !(a2 > 0.0) || pmin < G || am1 > (EXT ? mcapx : mcap) || am2 > (EXT ? mcapx : mcap);
        if ((cp.flags & 1) == 0) dead |= bad;
        constexpr uint32_t ALL = (1u << D) - 1u, COREM = EXT ? (ALL & ~1u) : ALL;
        const uint32_t sold = (field >> SB) & ALL;
        const uint32_t obit = oidx < (uint32_t)D ? (1u << (D - 1 - oidx)) : 0u;
        const uint32_t neg1 = sold ^ srw ^ ((hi32(pm1) >> 31) ? ALL : 0u);
        uint32_t tgt1 = COREM & ~obit;
        if constexpr (HASF) tgt1 &= ~hgw;
        const bool any1 = tgt1 != 0u, pos1 = (tgt1 & ~neg1) != 0u;
        const double nu1 = pos1 ? am1 : -am1;
        const bool neg2 = (((sold ^ srw) & obit) != 0u) != ((hi32(pm2) >> 31) != 0u);
        bool has2 = (obit & COREM) != 0u;
        if constexpr (HASF) has2 = has2 && (obit & hgw) == 0u;
        const double nu2 = neg2 ? -am2 : am2;
        const double need1 = any1 ? nu1 - 0.75 * a1 + E : -1.0e300;
        const double need2 = has2 ? nu2 - 0.75 * (tlo == a1 ? a2 : a1) + E : -1.0e300;
        if ((cp.flags & 2) == 0) {
          const bool up1 = need1 > ow1, up2 = need2 > ow2;
          float d1 = 0.f, d2 = 0.f;
          if (up1) { const float nw = (float)((2.0 * need1 + zeta) * (1.0 + 0x1p-20)); d1 = nw - fw1; wrow[(2 * L) * ZC] = nw; }
          if (up2) { const float nw = (float)((2.0 * need2 + zeta) * (1.0 + 0x1p-20)); d2 = nw - fw2; wrow[(2 * L + 1) * ZC] = nw; }
          raised |= up1 || up2;
          if (__builtin_amdgcn_ballot_w64(up1 || up2) != 0) {
            static_for<DC>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              constexpr int col = B::col(E0 + j);
              const float dj = (uint32_t)j == oidx ? d2 : d1;
              if constexpr (col == 0) {
                W0 += dj;
              } else {
                int el = zq + Y::eff_shift(ILS, ZC, L, E0 + j);
                el -= el >= ZC ? ZC : 0;
                wstore(Wg + col * ZC + el, W[j] + dj);
              }
            });
          }
        }
      }
      if (active) fetch_row(std::integral_constant<int, (L + 1) % B::ROWS>{});      // (the next row's, of the next sweep after the last row)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      bar();                                       // the next row reads the sums this one wrote
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_sched_barrier(0);
    });
    if (active && raised) atomicOr(&flags[2 * slot], 1u);
    if (active && dead) atomicOr(&flags[2 * slot + 1], 1u);
    bar();
    if (active) {
      const bool any_dead = flags[2 * slot + 1] != 0u, any_raised = flags[2 * slot] != 0u;
      if (any_dead) active = false;
      else if (!any_raised) { certified = true; active = false; }
    }
    bar();                                         // (the flags are reset at the top of the next sweep)
    if (z == 0) flags[4 + slot] = active ? 1u : 0u;
    bar();
    uint32_t cont = 0;
    if constexpr (NS == 1) cont = flags[4 + slot];
    else for (int s = 0; s < NS; ++s) cont |= flags[4 + s];
    if (cont == 0u) break;                         // (uniform over the waves behind `bar`)
  }
  return certified;
}

}  // namespace nrx_certcore
