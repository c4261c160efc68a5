// Compile-time facts of the two NR LDPC base graphs shared by the decoder kernels (nrx_ldpc_dec2.hip: float32
// throughput kernel, nrx_ldpc_dec3.hip: float64 on-chip kernel): lifting sizes, (truncated) base graphs, per-layer
// structure and the barrier plan.  Reference: ldpc.py:46-666 (tables), ldpc.py:1495-1581 (layer schedule).
#pragma once
#include <stdint.h>
#include <utility>
#include <hip/hip_runtime.h>
#include "gen_ldpc_bg.h"

namespace nrx_ldpc {

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

// The first RA rows of a base graph (RA = all rows: the graph itself).  NR LDPC codes are raptor-like: the code of a
// higher rate is the sub-matrix of the rows whose extension parity was transmitted; a row whose extension column is
// punctured (all-zero LLRs) sends +-0 to its other columns, so dropping it changes nothing (nrx_ldpc_decode_rows_*).
template <int BG, int RA> struct GR : G<BG> {
  static_assert(RA >= 4 && RA <= G<BG>::ROWS, "active rows out of range");
  static constexpr int ROWS = RA;
  static constexpr int EDGES = G<BG>::row_start(RA);
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

// prefetch distance (layers) of the extension-column channel LLRs; must divide the number of extension layers
// (42 for BG1, 38 for BG2) so that ring slot = ordinal mod PFN stays consistent across iterations
template <int BG> constexpr int pfn() { return BG == 1 ? 3 : 2; }
// ... for a truncated graph with n_ext extension layers: the largest ring <= pfn<BG>() that divides n_ext
template <int BG> constexpr int pfn_for(int n_ext) {
  for (int k = pfn<BG>(); k > 1; --k)
    if (n_ext % k == 0) return k;
  return 1;
}

template <int BG, int RA = G<BG>::ROWS> struct Lay {  // compile-time layer facts (of the first RA rows)
  using B = GR<BG, RA>;
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
  // Ping-pong buffers: the k-th layer of an iteration that touches column c reads buffer (k & 1) and writes the other.
  static constexpr int touch_par(int L, int c) {
    int k = 0;
    for (int l = 0; l < L; ++l) k += (core_mask(l) >> c) & 1u;
    return k & 1;
  }
  // columns touched an odd number of times per iteration end up in buffer 1 and are copied back (step ROWS)
  static constexpr uint32_t odd_mask() {
    uint32_t m = 0;
    for (int c = 0; c < B::CORE; ++c) m |= (uint32_t)touch_par(B::ROWS, c) << c;
    return m;
  }
  static constexpr uint32_t step_mask(int s) { return s < B::ROWS ? core_mask(s) : odd_mask(); }
  // A workgroup barrier goes before step s (layers 0..ROWS-1, then the copy-back) only if s touches a column that
  // some step since the previous barrier touched: then both hazards between two touchers of a column (the later
  // one reads what the earlier wrote, and overwrites what the earlier read) are closed, and inside a layer reads and
  // writes hit different buffers.  Most extension rows meet their neighbours in no core column, so about a third of
  // the barriers go (BG1: 33 of 47 steps).  Steady-state placement over the cyclic order, re-checked from a cold start.
  static constexpr int STEPS = B::ROWS + 1;
  struct BarPlan { bool need[B::ROWS + 1]; bool ok; int count; };
  static constexpr BarPlan make_plan() {
    BarPlan p{};
    uint32_t mask[STEPS] = {};
    for (int l = 0; l < STEPS; ++l) mask[l] = step_mask(l);
    uint32_t touched = 0;
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < STEPS; ++l) {
        const bool need = (mask[l] & touched) != 0;
        touched = need ? mask[l] : (touched | mask[l]);
        if (it == 2) p.need[l] = need;
      }
    p.ok = true;
    p.count = 0;
    touched = 0;   // cold start: the initial fill is followed by a barrier
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < STEPS; ++l) {
        if (p.need[l]) touched = 0;
        if (mask[l] & touched) p.ok = false;
        touched |= mask[l];
      }
    for (int l = 0; l < STEPS; ++l) p.count += p.need[l] ? 1 : 0;
    return p;
  }
  static constexpr BarPlan plan = make_plan();
  static constexpr bool barrier_before(int s) { return plan.need[s]; }
  static constexpr bool barriers_ok() { return plan.ok; }
  // In-place storage (nrx_ldpc_dec3.hip: one buffer per column, a lane reads and writes the SAME element (z + shift)
  // mod Zc, so nothing crosses lanes inside a layer): a barrier goes before layer L iff L touches a column that some
  // layer since the previous barrier touched.  No copy-back step.
  struct BarPlanIn { bool need[B::ROWS]; bool ok; int count; };
  static constexpr BarPlanIn make_plan_inplace() {
    BarPlanIn p{};
    uint32_t touched = 0;
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        const uint32_t m = core_mask(l);
        const bool need = (m & touched) != 0;
        touched = need ? m : (touched | m);
        if (it == 2) p.need[l] = need;
      }
    p.ok = true;
    p.count = 0;
    touched = 0;   // cold start: the initial fill is followed by a barrier
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        if (p.need[l]) touched = 0;
        if (core_mask(l) & touched) p.ok = false;
        touched |= core_mask(l);
      }
    for (int l = 0; l < B::ROWS; ++l) p.count += p.need[l] ? 1 : 0;
    return p;
  }
  static constexpr BarPlanIn plan_in = make_plan_inplace();
  static constexpr bool barrier_in_before(int l) { return plan_in.need[l]; }
  // ---- Rotated rows (nrx_ldpc_dec3.hip, round 3).  Lane z of layer L handles check row (z + sigma_L) mod Zc instead of row z:
  // the layer's accesses to column c then hit element (z + sigma_L + shift_L(c)) mod Zc.  With sigma_L = -shift_L(0) in every
  // layer that has column 0, lane z always meets element z of column 0: that column lives in a REGISTER (no LDS access, no
  // address select) and stops forcing a barrier between layers that share nothing else.  A layer without column 0 that shares
  // only column 1 with its predecessor is rotated so that it meets the element of column 1 its predecessor just updated in
  // the same lane (`fwd1`): the value is handed over in a register and the barrier in front of that layer goes as well.
  // (The check-node state of a layer is per row, and a layer's rows never change lanes, so nothing else moves.)
  static constexpr bool has_col(int L, int c) {
    for (int e = B::row_start(L); e < B::row_start(L + 1); ++e)
      if (B::col(e) == c) return true;
    return false;
  }
  static constexpr int shift_of(int ils, int zc, int L, int c) {
    for (int e = B::row_start(L); e < B::row_start(L + 1); ++e)
      if (B::col(e) == c) return B::shift(ils, e) % zc;
    return 0;
  }
  static constexpr bool fwd1(int L) {          // layer L takes column 1 from the register its predecessor left it in
    const int P = (L + B::ROWS - 1) % B::ROWS;
    return !has_col(L, 0) && has_col(L, 1) && has_col(P, 1) && has_col(P, 0);
  }
  static constexpr bool give1(int L) { return fwd1((L + 1) % B::ROWS); }   // ... and this layer leaves it there (no LDS write)
  static constexpr int sigma(int ils, int zc, int L) {
    if (has_col(L, 0)) return (zc - shift_of(ils, zc, L, 0)) % zc;
    if (fwd1(L)) {
      const int P = (L + B::ROWS - 1) % B::ROWS;
      return ((zc - shift_of(ils, zc, P, 0)) % zc + shift_of(ils, zc, P, 1) + zc - shift_of(ils, zc, L, 1)) % zc;
    }
    return 0;
  }
  static constexpr int eff_shift(int ils, int zc, int L, int e) { return (B::shift(ils, e) % zc + sigma(ils, zc, L)) % zc; }
  // columns of layer L that go through LDS (column 0 never does)
  static constexpr uint32_t lds_mask(int L) { return core_mask(L) & ~1u; }
  // ... of which the layer READS from LDS (a handed-over column 1 is not read) -- what a barrier in front of it has to cover
  static constexpr uint32_t lds_read_mask(int L) { return lds_mask(L) & ~(fwd1(L) ? 2u : 0u); }
  static constexpr BarPlanIn make_plan_rot() {
    BarPlanIn p{};
    uint32_t touched = 0;
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        const bool need = (lds_read_mask(l) & touched) != 0;
        touched = need ? lds_mask(l) : (touched | lds_mask(l));
        if (it == 2) p.need[l] = need;
      }
    p.ok = true;
    p.count = 0;
    touched = 0;   // cold start: the initial fill is followed by a barrier
    for (int it = 0; it < 3; ++it)
      for (int l = 0; l < B::ROWS; ++l) {
        if (p.need[l]) touched = 0;
        if (lds_read_mask(l) & touched) p.ok = false;
        touched |= lds_mask(l);
      }
    for (int l = 0; l < B::ROWS; ++l) p.count += p.need[l] ? 1 : 0;
    return p;
  }
  static constexpr BarPlanIn plan_rot = make_plan_rot();
  // Priority level (0 = first quarter ... 3 = last) of the point `q4` quarters into layer L, counted over the whole stretch
  // between two barriers of plan_rot (a stretch may span several layers now): work ~ number of edges.
  static constexpr int prio_q(int L, int q4) {
    int first = L;                                       // first layer of the stretch: the nearest one (backwards) with a barrier in front
    for (int i = 0; i < B::ROWS && !plan_rot.need[first]; ++i) first = (first + B::ROWS - 1) % B::ROWS;
    int before = 0, total = 0;
    bool seen = false;
    for (int i = 0, l = first; i < B::ROWS; ++i, l = (l + 1) % B::ROWS) {
      if (i > 0 && plan_rot.need[l]) break;
      if (l == L) seen = true;
      if (!seen) before += deg(l);
      total += deg(l);
    }
    const int q = (4 * (4 * before + q4 * deg(L))) / (4 * total);
    return q > 3 ? 3 : q;
  }
  // the k-th layer (cyclically) with an extension column after layer L
  static constexpr int next_ext(int L, int k) {
    int l = L;
    for (int i = 0; i < k; ++i) {
      do { l = (l + 1) % B::ROWS; } while (!has_ext(l));
    }
    return l;
  }
};

// Layout of a code block's parked decoder state (nrx_ldpc_dec3.hip parks and resumes it, nrx_ldpc_cert.hip reads it): NF fields of Zc
// doubles -- posterior columns 1 .. CORE-1 (element order), column 0 (lane order = element order), the handed-over column 1, the scaled
// minima pm1 / pm2 of every row, the extension posteriors, the sign / argmin words (low half of a double) -- all but the columns in
// the decoder's lane frame (lane z of layer L = check row z + sigma_L).
template <int BG, int RA> struct StateLay {
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
  static constexpr int NEXT = Y::n_ext() > 0 ? Y::n_ext() : 1;
  static constexpr int NW = Y::n_wide() > 0 ? Y::n_wide() : 1, NN = (Y::n_narrow() + 1) / 2;
  static constexpr int COL = 0;                       // columns 1 .. CORE-1 (column 0 lives in a register)
  static constexpr int C0 = COL + B::CORE - 1, F1 = C0 + 1, M1 = F1 + 1, M2 = M1 + B::ROWS, REXT = M2 + B::ROWS;
  static constexpr int WORDS = REXT + NEXT;           // one 32-bit word per slot (low half)
  static constexpr int NF = WORDS + NW + NN;          // fields of Zc doubles
};

constexpr int zindex_c(int zc) {
  for (int i = 0; i < NZ; ++i)
    if (kZ.z[i] == zc) return i;
  return -1;
}

}  // namespace nrx_ldpc
