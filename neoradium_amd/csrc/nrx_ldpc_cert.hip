// Stability certificate for early termination of the float64 layered min-sum decoder (ldpc.py:1495-1581 runs a fixed number of
// iterations, ldpc.py:1545, and has no early stop): the a-priori magnitude bounds (host), the certificate kernel on the state a
// stage of nrx_ldpc_dec3.hip parked, and the entry points.  DESIGN 4.3 has the statement and the proof; oracle/certificate.py is
// the CPU restatement the tests check this against.
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>
#include "nrx_ldpc_graph.h"
#include "nrx_common.h"
#include "nrx_ldpc_certcore.h"      // WINFL: the float32 slack sums' conservative inflation

int32_t nrx_ldpc_fused_rows_run(const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm, int32_t llr_len, int32_t n_rows, int* rows_run);   // nrx_ldpc_dec3.hip

namespace nrx_cert {
using namespace nrx_ldpc;

// STABILITY CERTIFICATE for early termination (DESIGN 4.3 has the statement and the proof; oracle/certificate.py is its CPU
// restatement).  The reference runs a fixed number of iterations (ldpc.py:1545) and has no early stop; this kernel decides, from
// the FROZEN decoder state a stage parked, whether every later iteration of that same float64 recursion provably leaves every
// hard decision where it is.  One code block per workgroup, in the decoder's own lane frame (lane z of layer L = check row
// z + sigma_L; its edge e meets element z + eff_shift(L, e) of the column; column 0 = element z in every layer).
//
// Normalised by the frozen hard decision s_c = sign(r_c) of its column:  tau = s_c (r_c - m) is what a row would read next,
// nu = s_c m the stored message.  A slack w >= 0 per message defines the floor nu - w; W_c = sum of the slacks of the messages into
// column element c; a future tau can lose at most  loss = W_c - w_own + 2E  against the frozen one (2E alone for the row's own
// degree-1 column).  The certificate holds when some w satisfies, on every check row,
//   (S) even parity of the hard decisions, |r_c| >= W_c + G, at most one edge with tau - loss <= 0, |m| <= mcap, and
//   (M) 0.75 min_{k != c} (tau_k - loss_k) >= nu_c - w_c + E  for every edge c into a core column.
// w is looked for by Gauss-Seidel relaxation (a row raises its slacks to what (M) asks, doubled, and updates W at once); a sweep
// in which nothing was raised and nothing failed has verified the final w on a state that did not move: certified.
// A row's messages take two values (pm1 to every edge but the old argmin, pm2 to that one), so it keeps two slacks.
// (Tried and not kept: two wave groups per code block -- 768 threads, group g the rows L = g mod 2, LDS atomics -- to put twelve waves
//  on a CU instead of six: at 168 registers per lane the row body spilled, and a step that is Jacobi inside needed four sweeps where
//  Gauss-Seidel needs two: 36.8 against 28.8 ms per 256-slot step of the whole chain.)
struct CertParams {
  double gamma, gamma1;      // a-priori magnitude bounds per unit of the LLR maxima (nrx_ldpc_cert_bounds)
  int32_t dmax, n_iter_total, max_sweeps, flags, iter_now;
};
// ---- helpers (the decoder's message representation, nrx_ldpc_dec3.hip: a message is unit * pm with unit = +-2^-7 carrying the
// sign of the old extrinsic value and pm = +-96 * min carrying the row parity; the old argmin's lanes take pm2 under EXEC)
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
// x with the sign of s folded in: x if s >= 0 else -x  (bit operation on the high word)
__device__ __forceinline__ double flip_by(double x, double s) {
  u32x2 v = __builtin_bit_cast(u32x2, x);
  v.y ^= hi32(s) & 0x80000000u;
  return __builtin_bit_cast(double, v);
}
__device__ __forceinline__ double fmin2(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmax2(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// a workgroup-uniform double as a scalar (SGPR pair) value
__device__ __forceinline__ double uniform(double x) {
  u32x2 v = __builtin_bit_cast(u32x2, x);
  v.x = __builtin_amdgcn_readfirstlane(v.x);
  v.y = __builtin_amdgcn_readfirstlane(v.y);
  return __builtin_bit_cast(double, v);
}

template <int BG, int RA> constexpr int cert_col_degree(int c) {
  int n = 0;
  for (int e = 0; e < GR<BG, RA>::EDGES; ++e) n += GR<BG, RA>::col(e) == c ? 1 : 0;
  return n;
}

// developer counters (nrx_debug_cert_sweeps): [k] = blocks certified in sweep k (k < 15), [15] = blocks refused
__device__ unsigned long long g_cert_hist[16];

// HASF: the configuration has filler bits (their posteriors sit near 1e10: known bits, never a minimum, no floor asked of or for them)
template <int BG, int ZI, int RA, bool HASF>
__global__ void __launch_bounds__(kZ.z[ZI], 1)
ldpc_certify_kernel(const double* __restrict__ state, int n_cb, const int32_t* __restrict__ sel, const int32_t* __restrict__ n_sel,
                    const uint8_t* __restrict__ cb_ok, const double* __restrict__ lam, CertParams cp, uint8_t* __restrict__ exit_iter) {
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  using SL = StateLay<BG, RA>;
  constexpr int ZC = kZ.z[ZI];
  constexpr int ILS = kZ.ils[ZI];
  static_assert(ZC % 64 == 0, "whole waves only");
  __shared__ double R[(B::CORE - 1) * ZC];      // posteriors of columns 1 .. CORE-1 (column 0: a register, element z in lane z)
  __shared__ float Wl[(B::CORE - 1) * ZC];      // slack sums of the same elements (single precision, like the slacks)
  __shared__ double Rx[SL::NEXT * ZC];          // posteriors of the rows' own degree-1 columns (lane order): registers are short
  static_assert((sizeof(double) + sizeof(float)) * (B::CORE - 1) * ZC + sizeof(double) * SL::NEXT * ZC <= 160 * 1024, "LDS budget");
  const int z = (int)threadIdx.x;
  if (sel) n_cb = *n_sel;
  for (int i = blockIdx.x; i < n_cb; i += gridDim.x) {
    const int cb = sel ? sel[i] : i;
    if ((cp.flags & 4) == 0 && cb_ok[cb] == 0) continue;       // (workgroup-uniform) a block whose CRC fails goes on decoding
    // ---- the error budget of this block (oracle/certificate.py:margins)
    const double lam_all = lam[2 * (size_t)cb], lam_pe = lam[2 * (size_t)cb + 1];
    const double beta = cp.gamma * lam_all;
    const double E_ = 2.0 * cp.n_iter_total * cp.dmax * 0x1p-53 * beta * 1.0625;
    const double rmin = (cp.gamma1 > 1.0 ? cp.gamma1 : 1.0) * lam_pe * (1 + 1e-9) + E_;
    const double mcap_ = 0.75 * (1.0e5 - rmin) * (1 - 1e-9) - E_;                       // core rows: their smallest |t| stays below gamma1 * lam_pe
    const double mcapx_ = 0.75 * (1.0e5 - (lam_pe * (1 + 1e-9) + E_)) * (1 - 1e-9) - E_;   // rows with their own degree-1 column: below that column's |LLR| + E
    const double E = uniform(E_), zeta = uniform(4.0 * E_), G = uniform(4.0 * E_);
    const double mcap = uniform(mcap_), mcapx = uniform(mcapx_);
    const bool bounds_ok = beta < 2.5e8 && lam_all < 1e9 && mcap > 0.0 && beta == beta && mcap == mcap;     // (NaN / inf refuse)
    // ---- the frozen state
    const double* st = state + (size_t)cb * SL::NF * ZC + z;
    double m1[B::ROWS], m2[B::ROWS];
    float w1[B::ROWS], w2[B::ROWS];            // (single precision is plenty for a slack: the same stored value is used everywhere)
    uint32_t sgw[SL::NW], sgn[SL::NN];
    __syncthreads();                             // (the previous block's sweeps are done with R / Wl)
    // every message starts with the slack zeta: a column element's sum is zeta times the column's degree.
    // (All loads of the state are issued before anything uses one: written one field at a time the compiler waited for every
    //  load where it issued it -- 78 dependent round trips per block, most of the kernel's time.)
    const float zf = (float)zeta;
    double cols[B::CORE - 1];
    static_for<B::CORE - 1>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      cols[c] = st[(size_t)(SL::COL + c) * ZC];
    });
    const double c0 = st[(size_t)SL::C0 * ZC];
    constexpr int DEG0 = cert_col_degree<BG, RA>(0);
    float W0 = zf * DEG0;
    __builtin_amdgcn_sched_barrier(0);
    static_for<B::CORE - 1>([&](auto cc) __attribute__((always_inline)) {
      constexpr int c = decltype(cc)::value;
      constexpr int DEGC = cert_col_degree<BG, RA>(c + 1);     // (constexpr variable: as a plain call it was evaluated at run time)
      R[c * ZC + z] = cols[c];
      Wl[c * ZC + z] = zf * DEGC;
    });
    __builtin_amdgcn_sched_barrier(0);           // (two batches: all 78 fields in flight at once do not fit the register file)
    double rx[SL::NEXT];
    double wd[SL::NW + SL::NN];
    static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
      constexpr int L = decltype(lc)::value;
      m1[L] = st[(size_t)(SL::M1 + L) * ZC];
      m2[L] = st[(size_t)(SL::M2 + L) * ZC];
      if constexpr (Y::has_ext(L)) rx[Y::ext_idx(L)] = st[(size_t)(SL::REXT + Y::ext_idx(L)) * ZC];
    });
    static_for<SL::NW + SL::NN>([&](auto ic) __attribute__((always_inline)) {
      wd[decltype(ic)::value] = st[(size_t)(SL::WORDS + decltype(ic)::value) * ZC];
    });
    __builtin_amdgcn_sched_barrier(0);
    static_for<SL::NEXT>([&](auto ic) __attribute__((always_inline)) { Rx[decltype(ic)::value * ZC + z] = rx[decltype(ic)::value]; });
    static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
      w1[decltype(lc)::value] = (float)zeta;
      w2[decltype(lc)::value] = (float)zeta;
    });
    static_for<SL::NW>([&](auto ic) __attribute__((always_inline)) {
      sgw[decltype(ic)::value] = (uint32_t)__double_as_longlong(wd[decltype(ic)::value]);
    });
    static_for<SL::NN>([&](auto ic) __attribute__((always_inline)) {
      sgn[decltype(ic)::value] = (uint32_t)__double_as_longlong(wd[SL::NW + decltype(ic)::value]);
    });
    __syncthreads();
    bool certified = false;
    int last_sweep = 0;
    bool dead = !bounds_ok;                      // a failure that more slack cannot cure (slacks only grow)
    u32x2 uv = {0u, 0u};
    for (int sweep = 0; sweep < cp.max_sweeps && !certified; ++sweep) {
      bool raised = false;
      last_sweep = sweep;
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        constexpr int E0 = B::row_start(L);
        constexpr int D = Y::deg(L);
        constexpr bool EXT = Y::has_ext(L);
        constexpr int DC = EXT ? D - 1 : D;        // edges into core columns
        constexpr bool WIDE = Y::wide(L);
        int zq = z;                                // (opaque per row and sweep: otherwise every element address and everything derived
        asm volatile("" : "+v"(zq));               //  from the frozen state alone is hoisted in front of the sweep loop and spilled there)
        uint32_t field, oidx;
        constexpr int SB = WIDE ? 5 : 4;           // the D sign bits sit at [SB + D - 1 : SB], edge 0 highest
        if constexpr (WIDE) {
          field = sgw[Y::wide_idx(L)];
          asm volatile("" : "+v"(field));
          oidx = field & 31u;
        } else {
          constexpr int ni = Y::narrow_idx(L);
          field = (ni & 1) ? (sgn[ni / 2] >> 16) : (sgn[ni / 2] & 0xffffu);
          asm volatile("" : "+v"(field));
          oidx = field & 15u;
        }
        double pm1 = m1[L], pm2 = m2[L];
        asm volatile("" : "+v"(pm1), "+v"(pm2));
        const double ow1 = (double)w1[L], ow2 = (double)w2[L];
        const double k1 = 2.0 * E - ow1, k2 = 2.0 * E - ow2;       // loss = W + k: the sum of the OTHER messages' slacks + 2E
        // ---- every read of the row first
        double r[D];
        float W[DC];
        static_for<D>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          constexpr int col = B::col(E0 + j);
          if constexpr (col == 0) {
            r[j] = c0;
            W[j] = W0;
          } else if constexpr (col < B::CORE) {
            int el = zq + Y::eff_shift(ILS, ZC, L, E0 + j);
            el -= el >= ZC ? ZC : 0;
            r[j] = R[(col - 1) * ZC + el];
            W[j] = Wl[(col - 1) * ZC + el];
          } else {
            r[j] = Rx[Y::ext_idx(L) * ZC + zq];
          }
        });
        // ---- one pass: tl = s (r - m) - loss of every edge (what the row would read next, ldpc.py:1550, less what it can lose), its two
        // smallest values, the value at the old argmin, the signs of the posteriors, the smallest |r| - W
        uint32_t wrun = field << (31 - (SB + D - 1));      // sign of edge 0's old extrinsic value at bit 31; doubled per edge
        uint32_t srw = 0, hgw = 0;                          // bit D-1-j: sign of r_j / r_j is a filler
        double a1 = 1.0e300, a2 = 1.0e300, tlo = 1.0e300, pmin = 1.0e300;
        static_for<D>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          constexpr int col = B::col(E0 + j);
          const double u = unit_of(uv, wrun);
          if constexpr (j < D - 1) wrun += wrun;
          const double t = sel_fnma_x<j>(r[j], oidx, u, pm1, pm2);      // r - m, the decoder's own rounding
          const double tau = flip_by(t, r[j]);
          srw = __builtin_amdgcn_alignbit(srw, hi32(r[j]), 31);         // (srw << 1) | sign(r_j)
          double tl;
          if constexpr (col < B::CORE) {
            // float32 slack sums: W x WINFL bounds the true sum from above (nrx_ldpc_certcore.h)
            const double x = __builtin_fma(-(double)W[j], nrx_certcore::WINFL, tau);
            tl = x - ((uint32_t)j == oidx ? k2 : k1);
            double pr = __builtin_fma(-(double)W[j], nrx_certcore::WINFL, __builtin_fabs(r[j]));
            if constexpr (HASF) {
              const bool hg = __builtin_fabs(r[j]) >= 5.0e8;
              hgw = (hgw << 1) | (hg ? 1u : 0u);
              tl = hg ? 1.0e300 : tl;
              pr = hg ? 1.0e300 : pr;
            }
            pmin = fmin2(pmin, pr);
            tlo = (uint32_t)j == oidx ? tl : tlo;
          } else {
            tl = tau - 2.0 * E;                                         // nothing else writes the row's own degree-1 column
            if constexpr (HASF) hgw <<= 1;
          }
          a2 = fmin2(a2, fmax2(a1, tl));
          a1 = fmin2(a1, tl);
        });
        // ---- (S): even parity of the hard decisions, the second smallest floor positive (at most one edge may sit at or under
        // zero), every posterior clear of the slack it may lose, no message above mcap
        const double am1 = __builtin_fabs(pm1) * 0x1p-7, am2 = __builtin_fabs(pm2) * 0x1p-7;      // |messages| (exact)
        const bool bad = (__builtin_popcount(srw) & 1) != 0 || !(a2 > 0.0) || pmin < G || am1 > (EXT ? mcapx : mcap) || am2 > (EXT ? mcapx : mcap);
        if ((cp.flags & 1) == 0) dead |= bad;
        // ---- (M): the normalised messages nu = s m.  Their signs as bit words (bit D-1-j = edge j): old extrinsic sign ^ row parity of
        // the message ^ sign of the posterior.  pm1-messages go to every edge but the old argmin, the pm2-message to that one.
        constexpr uint32_t ALL = (1u << D) - 1u, COREM = EXT ? (ALL & ~1u) : ALL;
        const uint32_t sold = (field >> SB) & ALL;
        const uint32_t obit = oidx < (uint32_t)D ? (1u << (D - 1 - oidx)) : 0u;
        const uint32_t neg1 = sold ^ srw ^ ((hi32(pm1) >> 31) ? ALL : 0u);      // bit set: that pm1-message is negative after normalisation
        uint32_t tgt1 = COREM & ~obit;
        if constexpr (HASF) tgt1 &= ~hgw;
        const bool any1 = tgt1 != 0u, pos1 = (tgt1 & ~neg1) != 0u;
        const double nu1 = pos1 ? am1 : -am1;      // the largest of them
        const bool neg2 = (((sold ^ srw) & obit) != 0u) != ((hi32(pm2) >> 31) != 0u);
        bool has2 = (obit & COREM) != 0u;
        if constexpr (HASF) has2 = has2 && (obit & hgw) == 0u;
        const double nu2 = neg2 ? -am2 : am2;
        // every pm1-message is held against 0.75 * a1 (the edge that holds the smallest floor could be held against a2 >= a1: the
        // stronger demand costs at most some slack); the pm2-message against a2 when its own edge holds the smallest floor
        const double need1 = any1 ? nu1 - 0.75 * a1 + E : -1.0e300;
        const double need2 = has2 ? nu2 - 0.75 * (tlo == a1 ? a2 : a1) + E : -1.0e300;
        double d1 = 0.0, d2 = 0.0;                 // what this row adds to its two slacks
        if ((cp.flags & 2) == 0) {
          if (need1 > ow1) { const float nw = (float)((2.0 * need1 + zeta) * (1.0 + 0x1p-20)); d1 = (double)nw - ow1; w1[L] = nw; }
          if (need2 > ow2) { const float nw = (float)((2.0 * need2 + zeta) * (1.0 + 0x1p-20)); d2 = (double)nw - ow2; w2[L] = nw; }
        }
        // ---- ... and into W at once (Gauss-Seidel).  A layer's lanes meet distinct elements of a column.
        const bool up = d1 != 0.0 || d2 != 0.0;
        raised |= up;
        if (__builtin_amdgcn_ballot_w64(up) != 0) {
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            const double dj = (uint32_t)j == oidx ? d2 : d1;
            if constexpr (col == 0) {
              W0 += (float)dj;
            } else {
              int el = zq + Y::eff_shift(ILS, ZC, L, E0 + j);
              el -= el >= ZC ? ZC : 0;
              Wl[(col - 1) * ZC + el] = W[j] + (float)dj;
            }
          });
        }
        __syncthreads();                           // the next layer reads the sums this one wrote
        __builtin_amdgcn_sched_barrier(0);         // nothing migrates between rows (register pressure)
      });
      const int any_dead = __syncthreads_or(dead ? 1 : 0);
      const int any_raised = __syncthreads_or(raised ? 1 : 0);
      if (any_dead) break;
      if (!any_raised) certified = true;
    }
    if (certified && z == 0) exit_iter[cb] = (uint8_t)(cp.iter_now < 255 ? cp.iter_now : 255);
    if (z == 0) atomicAdd(&g_cert_hist[certified ? (last_sweep < 15 ? last_sweep : 14) : 15], 1ull);
  }
}


}  // namespace nrx_cert

// ---- Certified early exit.  A-priori magnitude bounds of the recursion on the first n_rows rows of the base graph,
// per unit of the LLR maxima (scale invariant, independent of the lifting size):  |t_{i,c}| <= |L_c| + sum_{j != i} |m_{j,c}|  and
// |m_{i,c}| <= 0.75 min_{k != c} |t_{i,k}|  hold in every iteration whatever the signs (the +1e5 quirk, ldpc.py:1563, only lowers a
// second minimum), so every magnitude stays below the least fixed point of  V_{i,c} = lam_c + sum_{j != i} U_{j,c},
// U_{i,c} = 0.75 min_{k != c} V_{i,k}  (Kleene iteration from 0).  out[0] = gamma: bound of |r| with lam = 1 on every received
// column (infinity on the columns that hold fillers); out[1] = gamma1: bound of the smallest |t| of the four core rows with lam = 1
// on the core-parity and extension columns only (infinity on every information column); out[2] = the largest column degree.
// Infinity when the iteration does not settle (no certificate then).  Host only.
namespace {
struct CertGraph {
  int rows, ncol, kb, core;
  const int16_t* row_start;
  const int16_t* col;
};
bool cert_lfp(const CertGraph& g, const double* lam, double* V /* per edge */, double* U /* per edge */) {
  const int ne = g.row_start[g.rows];
  for (int e = 0; e < ne; ++e) U[e] = 0.0;
  const double INF = __builtin_inf();
  for (int it = 0; it < 200000; ++it) {
    double S[80];
    int ninf[80];
    for (int c = 0; c < g.ncol; ++c) { S[c] = 0.0; ninf[c] = 0; }
    for (int e = 0; e < ne; ++e) {
      if (U[e] < INF) S[g.col[e]] += U[e]; else ninf[g.col[e]] += 1;
    }
    double change = 0.0, umax = 0.0;
    for (int i = 0; i < g.rows; ++i) {
      const int e0 = g.row_start[i], e1 = g.row_start[i + 1];
      int k1 = -1, k2 = -1;          // the smallest and the second smallest V of the row (first index wins ties)
      for (int e = e0; e < e1; ++e) {
        const int c = g.col[e];
        const bool fin = U[e] < INF;
        double v = lam[c] + S[c] - (fin ? U[e] : 0.0);
        if (ninf[c] - (fin ? 0 : 1) > 0) v = INF;
        V[e] = v;
        if (k1 < 0 || v < V[k1]) { k2 = k1; k1 = e; }
        else if (k2 < 0 || v < V[k2]) k2 = e;
      }
      for (int e = e0; e < e1; ++e) {
        const double nu = 0.75 * (e == k1 ? V[k2] : V[k1]);
        const bool f0 = U[e] < INF, f1 = nu < INF;
        if (f0 != f1) change = INF;
        else if (f0) { const double d = nu > U[e] ? nu - U[e] : U[e] - nu; if (d > change) change = d; }
        U[e] = nu;
        if (f1 && nu > umax) umax = nu;
      }
    }
    if (umax > 1e9) return false;
    if (change < 1e-13) return true;
  }
  return false;
}
}  // namespace

extern "C" int32_t nrx_ldpc_cert_bounds(const nrx_ldpc_cfg* cfg, int32_t n_rows, double* out3) {
  NRX_REQUIRE(cfg && out3, NRX_E_ARG, "nrx_ldpc_cert_bounds: NULL argument");
  NRX_REQUIRE(cfg->bg == 1 || cfg->bg == 2, NRX_E_ARG, "nrx_ldpc_cert_bounds: base graph %d", cfg->bg);
  CertGraph g;
  if (cfg->bg == 1) g = CertGraph{NRX_BG1_ROWS, NRX_BG1_COLS, 22, 26, kBg1RowStart, kBg1Col};
  else g = CertGraph{NRX_BG2_ROWS, NRX_BG2_COLS, 10, 14, kBg2RowStart, kBg2Col};
  NRX_REQUIRE(n_rows >= 4 && n_rows <= g.rows, NRX_E_ARG, "nrx_ldpc_cert_bounds: %d rows", n_rows);
  g.rows = n_rows;
  const int ne = g.row_start[g.rows];
  const double INF = __builtin_inf();
  double lam[80], V[320], U[320];
  int deg[80] = {};
  for (int e = 0; e < ne; ++e) deg[g.col[e]] += 1;
  int dmax = 0;
  for (int c = 0; c < g.ncol; ++c) dmax = deg[c] > dmax ? deg[c] : dmax;
  out3[0] = out3[1] = INF;
  out3[2] = dmax;
  // gamma: lam = 1 on every column (0 on the two punctured ones), infinity on the columns that hold a filler bit
  const int first_filler = cfg->F > 0 ? (cfg->K - cfg->F) / cfg->Zc : g.kb;
  for (int c = 0; c < g.ncol; ++c) lam[c] = c < 2 ? 0.0 : ((c >= first_filler && c < g.kb) ? INF : 1.0);
  if (cert_lfp(g, lam, V, U)) {
    double S[80] = {};
    bool fin[80];
    for (int c = 0; c < g.ncol; ++c) fin[c] = true;
    for (int e = 0; e < ne; ++e) { if (U[e] < INF) S[g.col[e]] += U[e]; else fin[g.col[e]] = false; }
    double gm = 0.0;
    bool ok = true;
    for (int c = 0; c < g.ncol; ++c) {
      if (deg[c] == 0) continue;
      if (!fin[c]) { ok = false; break; }                       // (a message of unbounded size into a column whose elements are not all fillers)
      const double v = (c < 2 ? 0.0 : 1.0) + S[c];              // an element of a filler column that is not a filler has |L| <= the maximum too
      gm = v > gm ? v : gm;
    }
    if (ok) out3[0] = gm * (1 + 1e-9);
  }
  // gamma1: lam = 1 on the core-parity and extension columns, infinity on the transmitted information columns
  for (int c = 0; c < g.ncol; ++c) lam[c] = c < 2 ? 0.0 : (c < g.kb ? INF : 1.0);
  if (cert_lfp(g, lam, V, U)) {
    double g1 = 0.0;
    for (int i = 0; i < 4; ++i) {
      double mn = INF;
      for (int e = g.row_start[i]; e < g.row_start[i + 1]; ++e) mn = V[e] < mn ? V[e] : mn;
      g1 = mn > g1 ? mn : g1;
    }
    out3[1] = g1 < INF ? g1 * (1 + 1e-9) : INF;
  }
  return NRX_OK;
}

// The certificate on the parked state of the blocks sel[0 .. *n_sel) (all n_tb * C blocks when sel is NULL) whose cb_ok is 1:
// exit_iter[cb] = min(iter_now, 255) where it holds (left alone elsewhere: the caller zeroes the array once per batch, and
// nrx_select_failed of it lists the blocks that go on).  flags: bit 2 (with the others: tests only) also takes blocks whose CRC fails; bit 0
// skips condition (S), bit 1 condition (M) -- deliberately broken certificates for the tests, never for a measurement.
extern "C" int32_t nrx_ldpc_certify_f64(const void* state, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm,
                                        int32_t n_rows, const int32_t* sel, const int32_t* n_sel, const uint8_t* cb_ok, const double* lam,
                                        int32_t iter_now, int32_t n_iter_total, int32_t max_sweeps, int32_t flags, uint8_t* exit_iter,
                                        void* stream) {
  using namespace nrx_cert;
  NRX_REQUIRE(state && cfg && cb_ok && lam && exit_iter, NRX_E_ARG, "nrx_ldpc_certify: NULL buffer");
  NRX_REQUIRE((sel == nullptr) == (n_sel == nullptr), NRX_E_ARG, "nrx_ldpc_certify: selection list without its count");
  NRX_REQUIRE(iter_now >= 1 && n_iter_total >= iter_now && max_sweeps >= 1, NRX_E_ARG, "nrx_ldpc_certify: bad iteration counts");
  int rows_run = 0;
  const int32_t rc = nrx_ldpc_fused_rows_run(cfg, nl, qm, llr_len, n_rows, &rows_run);
  if (rc) return rc;
  double b3[3];
  const int32_t rb = nrx_ldpc_cert_bounds(cfg, rows_run, b3);
  if (rb) return rb;
  CertParams cp;
  cp.gamma = b3[0];
  cp.gamma1 = b3[1];
  cp.dmax = (int32_t)b3[2];
  cp.n_iter_total = n_iter_total;
  cp.max_sweeps = max_sweeps < 16 ? max_sweeps : 16;      // (WINFL's update count assumes <= 16 sweeps)
  cp.flags = flags;
  cp.iter_now = iter_now;
  if (n_tb == 0) return NRX_OK;
  const int n_cb = n_tb * cfg->C;
  const int grid = n_cb < 2048 ? n_cb : 2048;
  constexpr int ZI384 = zindex_c(384);
  hipStream_t st = (hipStream_t)stream;
#define NRX_CERT_LAUNCH(RA_, HASF_) \
  hipLaunchKernelGGL((ldpc_certify_kernel<1, ZI384, RA_, HASF_>), dim3(grid), dim3(384), 0, st, (const double*)state, n_cb, sel, n_sel, cb_ok, lam, cp, exit_iter)
  if (rows_run == 13) { if (cfg->F > 0) NRX_CERT_LAUNCH(13, true); else NRX_CERT_LAUNCH(13, false); }
  else { if (cfg->F > 0) NRX_CERT_LAUNCH(15, true); else NRX_CERT_LAUNCH(15, false); }
#undef NRX_CERT_LAUNCH
  NRX_CHECK_LAUNCH("nrx_ldpc_certify_f64");
  return NRX_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The GENERIC form: ldpc.py:1495-1581 for any base graph, lifting size and row count, with the certificate evaluated after the
// iterations the caller names; a block that holds it stops there (its hard decisions are the final ones).  One code
// block per workgroup, lane z = check row z of every layer, ALL state in the caller's workspace (posteriors, stored messages, slack
// sums, slacks): plain code, no tuning -- the throughput path for the metric configuration is the stage kernel of nrx_ldpc_dec3.hip;
// this entry makes the certified schedule available (and testable) for every code the library decodes.
namespace nrx_cert {
struct GraphTab {
  int16_t row_start[48];
  int16_t col[320];
  int16_t shift[8][320];
};
template <int BG> constexpr GraphTab make_graph_tab() {
  GraphTab t{};
  using Gt = G<BG>;
  for (int r = 0; r <= Gt::ROWS; ++r) t.row_start[r] = (int16_t)Gt::row_start(r);
  for (int e = 0; e < Gt::EDGES; ++e) {
    t.col[e] = (int16_t)Gt::col(e);
    for (int i = 0; i < 8; ++i) t.shift[i][e] = (int16_t)Gt::shift(i, e);
  }
  return t;
}
__constant__ GraphTab kGraph1 = make_graph_tab<1>();
__constant__ GraphTab kGraph2 = make_graph_tab<2>();

struct GenArgs {
  int n_cb, Zc, ils, n_rows, N, K, F, kb, core, n_iter, n_checks;
  int checks[8];
  double gamma, gamma1;
  int dmax, max_sweeps, flags;
  size_t ws_doubles;       // per code block
};

template <int BG>
__global__ void __launch_bounds__(ZMAX, 1)
cert_generic_kernel(const double* __restrict__ llr, GenArgs a, double* __restrict__ ws, uint8_t* __restrict__ hard, uint8_t* __restrict__ exit_iter) {
  const GraphTab& gt = BG == 1 ? kGraph1 : kGraph2;
  const int Zc = a.Zc, z = (int)threadIdx.x;
  const bool on = z < Zc;
  const int ncol = a.core + a.n_rows - 4;                       // columns the rows that run touch
  __shared__ double redm[2 * (ZMAX / 64)];
  __shared__ int flag[4];
  constexpr int DMAX = 19;
  for (int cb = blockIdx.x; cb < a.n_cb; cb += gridDim.x) {
    double* R = ws + (size_t)cb * a.ws_doubles;                 // [ncol][Zc]
    double* W = R + (size_t)ncol * Zc;                          // [ncol][Zc]
    double* M1 = W + (size_t)ncol * Zc;                         // [rows][Zc]: 0.75 * min1 of the row's last visit
    double* M2 = M1 + (size_t)a.n_rows * Zc;                    // 0.75 * min2 (the message to the argmin)
    double* S1 = M2 + (size_t)a.n_rows * Zc;                    // slack of the pm1-messages
    double* S2 = S1 + (size_t)a.n_rows * Zc;                    // slack of the pm2-message
    unsigned long long* WD = (unsigned long long*)(S2 + (size_t)a.n_rows * Zc);   // bits 0..18: sign of the message to edge j, 24..28: argmin
    const double* in = llr + (size_t)cb * a.N;
    // ---- load (ldpc.py:1536-1538): two punctured columns of zeros in front, clip to +-1e10; the block's LLR maxima on the way
    double la = 0.0, lp = 0.0;
    if (on) {
      for (int c = 0; c < ncol; ++c) {
        double v = 0.0;
        const int p = (c - 2) * Zc + z;
        if (c >= 2 && p < a.N) {
          v = in[p];
          v = v < -1e10 ? -1e10 : (v > 1e10 ? 1e10 : v);
          v += 0.0;
          const bool filler = p >= a.K - a.F - 2 * Zc && p < a.K - 2 * Zc;
          const double av = filler ? 0.0 : __builtin_fabs(v);
          la = av > la ? av : la;
          if (c >= a.kb) lp = av > lp ? av : lp;
        }
        R[(size_t)c * Zc + z] = v;
      }
      for (int L = 0; L < a.n_rows; ++L) {
        M1[(size_t)L * Zc + z] = 0.0;
        M2[(size_t)L * Zc + z] = 0.0;
        WD[(size_t)L * Zc + z] = 0ull;
      }
    }
    for (int k = 1; k < 64; k <<= 1) {
      la = __builtin_fmax(la, __shfl_xor(la, k, 64));
      lp = __builtin_fmax(lp, __shfl_xor(lp, k, 64));
    }
    if ((z & 63) == 0) { redm[2 * (z >> 6)] = la; redm[2 * (z >> 6) + 1] = lp; }
    __syncthreads();
    la = 0.0; lp = 0.0;
    for (int w = 0; w < (int)blockDim.x / 64; ++w) { la = __builtin_fmax(la, redm[2 * w]); lp = __builtin_fmax(lp, redm[2 * w + 1]); }
    const double beta = a.gamma * la;
    const double E = 2.0 * a.n_iter * a.dmax * 0x1p-53 * beta * 1.0625;
    const double zeta = 4.0 * E, G = 4.0 * E;
    const double mcap = 0.75 * (1.0e5 - ((a.gamma1 > 1.0 ? a.gamma1 : 1.0) * lp * (1 + 1e-9) + E)) * (1 - 1e-9) - E;
    const double mcapx = 0.75 * (1.0e5 - (lp * (1 + 1e-9) + E)) * (1 - 1e-9) - E;
    const bool bounds_ok = beta < 2.5e8 && la < 1e9 && mcap > 0.0 && beta == beta && mcap == mcap;
    int next_check = 0, certified_at = 0;
    for (int it = 1; it <= a.n_iter && certified_at == 0; ++it) {
      // ---- one iteration (ldpc.py:1546-1576)
      for (int L = 0; L < a.n_rows; ++L) {
        __syncthreads();
        if (on) {
          const int e0 = gt.row_start[L], D = gt.row_start[L + 1] - e0;
          const unsigned long long wd = WD[(size_t)L * Zc + z];
          const int oidx = (int)((wd >> 24) & 31ull);
          const double pm1 = M1[(size_t)L * Zc + z], pm2 = M2[(size_t)L * Zc + z];
          double t[DMAX];
          int el[DMAX];
          int nneg = 0, am = 0;
          double m1 = 0.0;
          for (int j = 0; j < D; ++j) {
            const int c = gt.col[e0 + j];
            int x = z + gt.shift[a.ils][e0 + j] % Zc;
            x -= x >= Zc ? Zc : 0;
            el[j] = c * Zc + x;
            const double mag = j == oidx ? pm2 : pm1;
            const double m = ((wd >> j) & 1ull) ? -mag : mag;
            t[j] = R[el[j]] - m;
            nneg += t[j] < 0.0 ? 1 : 0;
            const double aj = __builtin_fabs(t[j]);
            if (j == 0 || aj < m1) { m1 = aj; am = j; }          // first index of the minimum (np.argmin)
          }
          double m2 = __builtin_fabs(t[am] + 100000.0);          // QUIRK ldpc.py:1563
          for (int j = 0; j < D; ++j)
            if (j != am) { const double aj = __builtin_fabs(t[j]); m2 = aj < m2 ? aj : m2; }
          const bool par = (nneg & 1) != 0;
          unsigned long long nw = (unsigned long long)am << 24;
          for (int j = 0; j < D; ++j) {
            const double mag = j == am ? m2 : m1;
            const bool ng = (t[j] < 0.0) != par;
            const double nm = (ng ? -mag : mag) * 0.75;            // (mag * sgn) * 0.75, ldpc.py:1570-1573
            R[el[j]] = t[j] + nm;
            nw |= (unsigned long long)((__double_as_longlong(nm) >> 63) & 1ll) << j;
          }
          M1[(size_t)L * Zc + z] = m1 * 0.75;
          M2[(size_t)L * Zc + z] = m2 * 0.75;
          WD[(size_t)L * Zc + z] = nw;
        }
      }
      __syncthreads();
      if (next_check >= a.n_checks || a.checks[next_check] != it) continue;
      ++next_check;
      // ---- the certificate on the frozen state (same conditions and relaxation as ldpc_certify_kernel)
      if (on) {
        for (int c = 0; c < ncol; ++c) W[(size_t)c * Zc + z] = 0.0;
      }
      __syncthreads();
      bool dead = !bounds_ok;
      bool certified = false;
      for (int sweep = -1; sweep < a.max_sweeps && !certified; ++sweep) {
        bool raised = false;
        for (int L = 0; L < a.n_rows; ++L) {
          if (on) {
            const int e0 = gt.row_start[L], D = gt.row_start[L + 1] - e0;
            const bool ext = gt.col[e0 + D - 1] >= a.core;
            const int DC = ext ? D - 1 : D;
            const unsigned long long wd = WD[(size_t)L * Zc + z];
            const int oidx = (int)((wd >> 24) & 31ull);
            double d1 = 0.0, d2 = 0.0;
            int el[DMAX];
            for (int j = 0; j < D; ++j) {
              const int c = gt.col[e0 + j];
              int x = z + gt.shift[a.ils][e0 + j] % Zc;
              x -= x >= Zc ? Zc : 0;
              el[j] = c * Zc + x;
            }
            if (sweep < 0) {
              S1[(size_t)L * Zc + z] = zeta;
              S2[(size_t)L * Zc + z] = zeta;
              d1 = d2 = zeta;
            } else {
              const double pm1 = M1[(size_t)L * Zc + z], pm2 = M2[(size_t)L * Zc + z];
              const double ow1 = S1[(size_t)L * Zc + z], ow2 = S2[(size_t)L * Zc + z];
              double a1 = 1e300, a2 = 1e300, tlo = 1e300, nu1 = -1e300, nu2 = 0.0;
              bool any1 = false, has2 = false, bad = false;
              int parity = 0, nonpos = 0;
              for (int j = 0; j < D; ++j) {
                const double r = R[el[j]];
                const double mag = j == oidx ? pm2 : pm1;
                const double m = ((wd >> j) & 1ull) ? -mag : mag;
                const double tj = r - m;
                const bool sr = __builtin_signbit(r);
                const bool hg = __builtin_fabs(r) >= 5.0e8;
                const bool core = j < DC;
                const double Wj = core ? W[el[j]] : 0.0;
                const double loss = core ? (Wj - (j == oidx ? ow2 : ow1)) + 2.0 * E : 2.0 * E;
                const double tl = hg ? 1e300 : (sr ? -tj : tj) - loss;
                const double nu = sr ? -m : m;
                parity ^= sr ? 1 : 0;
                nonpos += tl <= 0.0 ? 1 : 0;
                bad |= mag > (ext ? mcapx : mcap);
                if (core && !hg) {
                  bad |= __builtin_fabs(r) < Wj + G;
                  if (j == oidx) { nu2 = nu; has2 = true; tlo = tl; }
                  else { any1 = true; nu1 = nu > nu1 ? nu : nu1; }
                }
                if (tl < a1) { a2 = a1; a1 = tl; } else if (tl < a2) a2 = tl;
              }
              bad |= parity != 0 || nonpos > 1;
              if ((a.flags & 1) == 0) dead |= bad;
              const double need1 = any1 ? nu1 - 0.75 * a1 + E : -1e300;
              const double need2 = has2 ? nu2 - 0.75 * (tlo == a1 ? a2 : a1) + E : -1e300;
              if ((a.flags & 2) == 0) {
                if (need1 > ow1) { const double nwv = (2.0 * need1 + zeta) * (1.0 + 0x1p-20); d1 = nwv - ow1; S1[(size_t)L * Zc + z] = nwv; raised = true; }
                if (need2 > ow2) { const double nwv = (2.0 * need2 + zeta) * (1.0 + 0x1p-20); d2 = nwv - ow2; S2[(size_t)L * Zc + z] = nwv; raised = true; }
              }
            }
            if (d1 != 0.0 || d2 != 0.0)
              for (int j = 0; j < DC; ++j) W[el[j]] += j == oidx ? d2 : d1;
          }
          __syncthreads();
        }
        if (sweep < 0) continue;
        if (z == 0) { flag[0] = 0; flag[1] = 0; }
        __syncthreads();
        if (dead) atomicOr(&flag[0], 1);
        if (raised) atomicOr(&flag[1], 1);
        __syncthreads();
        const bool any_dead = flag[0] != 0, any_raised = flag[1] != 0;
        __syncthreads();
        if (any_dead) break;
        if (!any_raised) certified = true;
      }
      if (certified) certified_at = it;
    }
    // ---- hard decisions of the information columns (ldpc.py:1578-1581)
    __syncthreads();
    if (on) {
      for (int c = 0; c < a.kb; ++c) hard[(size_t)cb * a.K + c * Zc + z] = R[(size_t)c * Zc + z] < 0.0 ? 1 : 0;
    }
    if (z == 0) exit_iter[cb] = (uint8_t)(certified_at < 255 ? certified_at : 255);
    __syncthreads();
  }
}
}  // namespace nrx_cert

extern "C" int64_t nrx_ldpc_decode_certified_ws_bytes(const nrx_ldpc_cfg* cfg, int32_t n_rows) {
  if (!cfg || (cfg->bg != 1 && cfg->bg != 2)) return -1;
  const int total = cfg->bg == 1 ? 46 : 42, core = cfg->bg == 1 ? 26 : 14;
  const int rows = n_rows <= 0 || n_rows > total ? total : (n_rows < 4 ? 4 : n_rows);
  return (int64_t)sizeof(double) * ((size_t)2 * (core + rows - 4) + (size_t)5 * rows) * cfg->Zc;
}

// ldpc.py:1495-1581 decode (hard decisions of the K information bits) with the certified early exit for ANY configuration:
// llr (n_cb, N) rate-recovered LLRs, checks[0 .. n_checks) ascending iteration counts (host array, <= 8) after which the certificate is
// evaluated; hard_out (n_cb, K); exit_iter[cb] = the iteration the block was certified at (0: it ran all n_iter).  ws: n_cb *
// nrx_ldpc_decode_certified_ws_bytes(cfg, n_rows) bytes.
extern "C" int32_t nrx_ldpc_decode_certified_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                                 const int32_t* checks, int32_t n_checks, uint8_t* hard_out, uint8_t* exit_iter, void* ws,
                                                 size_t ws_bytes, int32_t max_sweeps, int32_t flags, void* stream) {
  using namespace nrx_cert;
  NRX_REQUIRE(llr && cfg && hard_out && exit_iter && ws && (checks || n_checks == 0), NRX_E_ARG, "nrx_ldpc_decode_certified: NULL buffer");
  NRX_REQUIRE(cfg->bg == 1 || cfg->bg == 2, NRX_E_ARG, "nrx_ldpc_decode_certified: base graph %d", cfg->bg);
  NRX_REQUIRE(n_cb >= 0 && n_iter >= 0 && n_checks >= 0 && n_checks <= 8 && max_sweeps >= 1, NRX_E_ARG, "nrx_ldpc_decode_certified: bad argument");
  NRX_REQUIRE(cfg->Zc >= 2 && cfg->Zc <= ZMAX && cfg->iLS >= 0 && cfg->iLS < 8, NRX_E_ARG, "nrx_ldpc_decode_certified: bad lifting size");
  const int total = cfg->bg == 1 ? 46 : 42;
  const int rows = n_rows <= 0 || n_rows > total ? total : (n_rows < 4 ? 4 : n_rows);
  const int64_t per = nrx_ldpc_decode_certified_ws_bytes(cfg, rows);
  NRX_REQUIRE(per > 0 && ws_bytes >= (size_t)per * (size_t)n_cb, NRX_E_ARG, "nrx_ldpc_decode_certified: workspace of %zu bytes, %lld needed", ws_bytes,
              (long long)per * n_cb);
  if (n_cb == 0) return NRX_OK;
  double b3[3];
  const int32_t rb = nrx_ldpc_cert_bounds(cfg, rows, b3);
  if (rb) return rb;
  GenArgs a{};
  a.n_cb = n_cb; a.Zc = cfg->Zc; a.ils = cfg->iLS; a.n_rows = rows; a.N = cfg->N; a.K = cfg->K; a.F = cfg->F;
  a.kb = cfg->bg == 1 ? 22 : 10; a.core = cfg->bg == 1 ? 26 : 14; a.n_iter = n_iter; a.n_checks = n_checks;
  for (int i = 0; i < n_checks; ++i) {
    NRX_REQUIRE(checks[i] >= 1 && (i == 0 || checks[i] > checks[i - 1]), NRX_E_ARG, "nrx_ldpc_decode_certified: checks must ascend from 1");
    a.checks[i] = checks[i];
  }
  a.gamma = b3[0]; a.gamma1 = b3[1]; a.dmax = (int)b3[2]; a.max_sweeps = max_sweeps; a.flags = flags;
  a.ws_doubles = (size_t)per / sizeof(double);
  const int threads = ((cfg->Zc + 63) / 64) * 64;
  const int grid = n_cb < 4096 ? n_cb : 4096;
  hipStream_t st = (hipStream_t)stream;
  if (cfg->bg == 1) hipLaunchKernelGGL(cert_generic_kernel<1>, dim3(grid), dim3(threads), 0, st, llr, a, (double*)ws, hard_out, exit_iter);
  else hipLaunchKernelGGL(cert_generic_kernel<2>, dim3(grid), dim3(threads), 0, st, llr, a, (double*)ws, hard_out, exit_iter);
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_certified_f64");
  return NRX_OK;
}

// developer hook: histogram of the sweep a block was certified in ([15] = refused); reset != 0 clears it
extern "C" int32_t nrx_debug_cert_sweeps(unsigned long long* out16, int32_t reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(nrx_cert::g_cert_hist), sizeof(unsigned long long) * 16) != hipSuccess) return NRX_E_HIP;
  if (reset) {
    const unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(nrx_cert::g_cert_hist), z, sizeof(z)) != hipSuccess) return NRX_E_HIP;
  }
  return NRX_OK;
}
