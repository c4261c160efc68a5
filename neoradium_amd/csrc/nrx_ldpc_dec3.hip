// Float64 layered min-sum LDPC decoder with the WHOLE working set on chip (hard decisions of the K information bits,
// bit-identical to the reference's float64 arithmetic, ldpc.py:1495-1581).
//
// nrx_ldpc_dec.hip keeps the float64 check-node state of all 46 (42) rows in an L2/MALL-resident workspace, because
// 527 KB per code block cannot live on chip.  NR LDPC is raptor-like, though: at the code rates the link actually
// runs, only the first RA rows have their extension parity transmitted and the others are exact no-ops for the
// information bits (see nrx_ldpc_decode_rows_*).  For RA <= 16 everything fits:
//   * LDS: the core columns of TWO code blocks, float64, one buffer per column, stored UNROTATED:
//     2 x 26 x 384 x 8 B = 156 KB.  A lane reads element (row + shift) mod Zc and writes its result back to the same element,
//     so nothing crosses lanes inside a layer and a workgroup barrier is only needed between layers that share a column.
//     Round 3: lane z of layer L handles row (z + sigma_L) mod Zc with sigma_L chosen so that it always meets element z of
//     COLUMN 0 (which every row but one of the truncated graph touches): column 0 lives in a register, not in LDS, and no longer
//     forces a barrier between layers that share nothing else; column 1 is handed from a layer to its successor in a register
//     where that is their only link (Lay::sigma / fwd1 / plan_rot): 11 barriers per iteration instead of 15;
//   * VGPRs (<= 168, three waves per SIMD): 0.75*min1 and 0.75*min2 of every layer (4 registers per layer), the packed
//     sign/argmin words, and the posterior of every layer's degree-1 extension column (2 registers per layer; in
//     float64 (r - m) + m' is not the channel LLR again, so it has to be carried);
//   * the lifting size is a template parameter: every (column, shift) pair is a DS immediate, the wrap-around of
//     (z + shift) is one v_cndmask between two base registers, selected by a wave-uniform 64-bit mask that a scalar
//     load fetches from a constant table.
// No HBM traffic between the initial fill and the hard decisions.  Arithmetic and its order are those of
// ldpc_dec_kernel<double, BG, true> (which stays the path for every other lifting size / row count and for soft output).
#include <stddef.h>
#include <stdlib.h>
#include <mutex>
#include "nrx_ldpc_graph.h"
#include "nrx_ldpc_certcore.h"
#include "nrx_common.h"

// SKIPZ (see the kernel): a wave whose 64 check rows of the LAST layer all have an extension LLR of exactly 0 leaves that layer out --
// each of those rows is the exact no-op of DESIGN 4.2a (min1 = 0 at the extension edge: +-0 to every other column), here per wave
// instead of per layer.  The last transmitted extension column is the one that is partly filled: 48 of 384 rows of layer 14 at the
// metric configuration (E_r - 34 Zc, no fillers), all in one wave: five of a code block's six waves skip 7 of the 157 edges.  Two copies of the iteration loop (with and
// without the last layer's body, same barriers) behind a wave-uniform branch: a test per layer inside ONE loop turned the layer's
// register updates into copies at the join (+112 VALU instructions per iteration).  Whole decodes only (MODE 0): a parked state keeps
// the posteriors of the extension columns, from which the zero LLR cannot be read back.
namespace nrx_dec3 {
using namespace nrx_ldpc;

// m[w][e]: lanes of wave w (of a code block) whose element (z + shift_e mod Zc) runs past the end of the column
struct WrapTab {
  uint64_t m[ZMAX / 64][ESTRIDE];
};
template <int BG, int ZI, int RA> constexpr WrapTab make_wrap() {
  WrapTab t{};
  using Y = Lay<BG, RA>;
  const int zc = kZ.z[ZI], ils = kZ.ils[ZI];
  for (int w = 0; w < ZMAX / 64; ++w)
    for (int L = 0; L < RA; ++L)
      for (int e = GR<BG, RA>::row_start(L); e < GR<BG, RA>::row_start(L + 1); ++e) {
        const int s = Y::eff_shift(ils, zc, L, e);      // shift of the edge + the layer's row rotation (Lay::sigma)
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
__device__ __forceinline__ uint32_t dbl(uint32_t w) {   // w + w as an add the optimiser cannot turn into a shift
  uint32_t r;
  asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(w));
  return r;
}

// ---- Messages as  unit * pm,  selected by EXEC (round 3).
// A check-to-variable message is  sign_j * 0.75 * (j == argmin ? min2 : min1)  with sign_j = parity ^ sign(t_j)
// (ldpc.py:1556-1573).  The row keeps pm1 / pm2 = the two scaled minima WITH THE ROW PARITY AS THEIR SIGN, and the sign of
// t_j becomes a unit u_j = +-2^-7 (pm carries the 2^7), so
//     r_j = t_j + msg_j = fma( u_j, pm, t_j)          t_j = r_j - msg_j = fma(-u_j, pm, r_j)
// u_j * pm is exact, so the fused operation rounds exactly like the reference's add / subtract of the signed message.
// Which of pm1 / pm2 a lane takes is decided by EXEC: every lane takes pm1, then a v_cmpx puts the argmin lanes into
// EXEC and they redo the operation with pm2 -- 3 VALU + 1 SALU instructions where two 64-bit selects (4 v_cndmask), a
// sign insert and an add were 6.  Why instruction count is the currency: the SIMD issues ONE VALU instruction per 4 cycles
// whatever the mix (float64, VOP3, VOP2 in a mixed stream) and, among its ready waves, always the oldest
// (tools/ubench/issue_probe.hip, profiles/r3_issue_probe*.txt); scalar instructions, s_nop and waits issue beside the vector pipe.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// +-2^-7 with the sign of bit 31 of signsrc.  uv.x stays 0: only the high word of the register pair is rewritten.  The high
// word of 2^-7 is 0x3f800000 = the inline constant 1.0f, so ONE v_and_or_b32 with the sign mask in an SGPR builds it (VOP3
// reads a single scalar operand on gfx9; the high word of 1.0, 0x3ff00000, would have to sit in a VGPR).  Scaling by a
// power of two commutes with rounding: 2^-7 * fl(96 * min) = fl(0.75 * min) exactly (short of 0.75 * min < 2^-1022, which a
// clipped LLR cannot produce other than as an exact zero).
__device__ __forceinline__ double unit_of(u32x2& uv, uint32_t signsrc) {
  uint32_t h;
  asm("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(h) : "v"(signsrc), "s"(0x80000000u));
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
// +-96.0 = +-0.75 * 2^7, negative for an odd number of set sign bits
__device__ __forceinline__ double c96_of(u32x2& uv, uint32_t signbits) {
  uint32_t h;
  asm("v_lshl_or_b32 %0, %1, 31, %2" : "=v"(h) : "v"((uint32_t)__builtin_popcount(signbits)), "s"(0x40580000u));
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
// min / max written out: after an inline-asm producer the compiler cannot prove its inputs canonical and puts a
// v_max_f64 x, x, x in front of every llvm.minnum / maxnum (one extra VALU instruction per edge); no NaN reaches this code
// (the LLRs are clipped to +-1e10 on the way in, ldpc.py:1536)
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
// One step of the running two-smallest (ldpc.py:1556-1564) with the sign collection IN BETWEEN: a float64 result cannot be read
// by the instruction (or the second instruction, for a float64 producer) behind it -- the compiler pads such pairs with s_nop 0, 105
// of them per iteration in this loop when the three min / max of a step stand together and the v_alignbit of all edges follow as a
// run (round 6: issue slots of the WAVE, not of the SIMD).  As one block the step needs none: max -> (min, alignbit) -> min.
#define NRX_MS_STEP(T, TH)                          \
  "v_max_f64 %[tmp], %[a1], |%[" T "]|\n\t"         \
  "v_min_f64 %[a1], %[a1], |%[" T "]|\n\t"          \
  "v_alignbit_b32 %[nsg], %[nsg], %[" TH "], 31\n\t" \
  "v_min_f64 %[a2], %[a2], %[tmp]\n\t"
__device__ __forceinline__ void minsum_step(double& a1, double& a2, uint32_t& nsg, double t) {
  double tmp;
  asm(NRX_MS_STEP("t", "th") : [tmp] "=&v"(tmp), [a1] "+v"(a1), [a2] "+v"(a2), [nsg] "+v"(nsg) : [t] "v"(t), [th] "v"(hi32(t)));
}
// ... four steps as one block: between two blocks the compiler cannot see which instruction wrote a1 last and pads again
__device__ __forceinline__ void minsum_step4(double& a1, double& a2, uint32_t& nsg, double t0, double t1, double t2, double t3) {
  double tmp;
  asm(NRX_MS_STEP("t0", "h0") NRX_MS_STEP("t1", "h1") NRX_MS_STEP("t2", "h2") NRX_MS_STEP("t3", "h3")
      : [tmp] "=&v"(tmp), [a1] "+v"(a1), [a2] "+v"(a2), [nsg] "+v"(nsg)
      : [t0] "v"(t0), [h0] "v"(hi32(t0)), [t1] "v"(t1), [h1] "v"(hi32(t1)), [t2] "v"(t2), [h2] "v"(hi32(t2)), [t3] "v"(t3), [h3] "v"(hi32(t3)));
}
__device__ __forceinline__ void minsum_step2(double& a1, double& a2, uint32_t& nsg, double t0, double t1) {
  double tmp;
  asm(NRX_MS_STEP("t0", "h0") NRX_MS_STEP("t1", "h1")
      : [tmp] "=&v"(tmp), [a1] "+v"(a1), [a2] "+v"(a2), [nsg] "+v"(nsg)
      : [t0] "v"(t0), [h0] "v"(hi32(t0)), [t1] "v"(t1), [h1] "v"(hi32(t1)));
}
#undef NRX_MS_STEP
// unit_of as an ordered statement ("memory": stores of the caller stay on their side of it): pass 2 puts the LDS write of the edge
// before between this unit build and the fma that reads it, where the compiler otherwise pads with an s_nop (156 per iteration)
__device__ __forceinline__ double unit_of_ordered(u32x2& uv, uint32_t signsrc) {
  uint32_t h;
  asm volatile("v_and_or_b32 %0, %1, %2, 1.0" : "=v"(h) : "v"(signsrc), "s"(0x80000000u) : "memory");
  uv.y = h;
  return __builtin_bit_cast(double, uv);
}
// lanes where |t| == a, as a mask in SGPRs (cold path only)
__device__ __forceinline__ uint64_t cmp_abs_eq(double t, double a) {
  uint64_t m;
  asm("v_cmp_eq_f64 %0, |%1|, %2" : "=s"(m) : "v"(t), "v"(a));
  return m;
}
// fma(-u, oidx == J ? p2 : p1, x)   (all 64 lanes are live: EXEC is -1 around the block)
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
// fma(u, |x| == a ? p2 : p1, x); idx <- J where |x| == a.  EVERY entry equal to the minimum takes p2: with a tie min2 == min1,
// so p2 == p1 bit for bit -- unless the +1e5 quirk has moved min2 under a tie, which the caller's cold path handles.
template <int J>
__device__ __forceinline__ double sel_fma_x(double x, uint32_t& idx, double a, double u, double p1, double p2) {
  double y;
  asm("v_fma_f64 %[y], %[u], %[p1], %[x]\n\t"
      "v_cmpx_eq_f64_e64 vcc, |%[x]|, %[a]\n\t"
      "v_fma_f64 %[y], %[u], %[p2], %[x]\n\t"
      "v_mov_b32 %[idx], %[j]\n\t"
      "s_mov_b64 exec, -1"
      : [y] "=&v"(y), [idx] "+v"(idx) : [x] "v"(x), [a] "v"(a), [j] "n"(J), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2) : "vcc");
  return y;
}
// Cold path: x <- fma(u, p2 or p1, x) with p2 for the lanes of im that are not in seen yet -- the FIRST minimum of a lane only,
// np.argmin's index (ldpc.py:1558-1570) -- and idx <- J there; seen |= im
template <int J>
__device__ __forceinline__ void sel_fma_first(double& x, uint32_t& idx, uint64_t& seen, uint64_t im, double u, double p1, double p2) {
  asm volatile("s_andn2_b64 exec, %[im], %[seen]\n\t"
               "s_or_b64 %[seen], %[seen], %[im]\n\t"
               "v_fma_f64 %[x], %[u], %[p2], %[x]\n\t"
               "v_mov_b32 %[idx], %[j]\n\t"
               "s_not_b64 exec, exec\n\t"
               "v_fma_f64 %[x], %[u], %[p1], %[x]\n\t"
               "s_mov_b64 exec, -1"
               : [x] "+v"(x), [idx] "+v"(idx), [seen] "+s"(seen) : [im] "s"(im), [u] "v"(u), [p1] "v"(p1), [p2] "v"(p2), [j] "n"(J) : "scc");
}


// Wave priority by progress inside a layer.  The SIMD issues one VALU instruction per 4 cycles and, among ready waves of
// equal priority, always picks the oldest (tools/ubench/issue_probe: three waves of one stream finish at T, 2T, 3T).  With a
// barrier per layer that leaves the youngest wave to run the end of every layer alone, at its own serial speed.  Lowering
// a wave's priority as it advances through the layer (3 -> 0) lets the waves that are behind go first, so the three waves
// of a SIMD reach the barrier together.
#define LAYER_PRIO(Q) __builtin_amdgcn_s_setprio(3 - (Q))

// ---- Barrier of ONE code block's six waves (the persistent certified schedule, MODE bit 4: the two code-block slots of a workgroup run
// different blocks at different iterations and must not wait for each other; a 384-thread workgroup per code block is not an option, a CU
// never co-schedules two of them at this register count).  Every wave keeps a progress word in LDS (the padding in front of the
// columns); arriving = storing the barrier's number there (one lane; the LDS executes a wave's instructions in order, so the word lands
// behind the wave's column writes and no drain is needed), waiting = polling the slot's six words until none is behind.  A wait that
// does not end (a bug, never the data) gives up after 2^22 polls and makes every later barrier of the wave fall through: the kernel
// always terminates, and the caller finds the error word set.
constexpr uint32_t SLOTBAR_BASE = 2048;      // byte offset inside the padding: [slot][8] progress words
constexpr uint32_t SLOTQ_BASE = 2048 + 64;   // [slot] the work-list position the slot drew
__device__ __forceinline__ void slot_barrier(uint32_t& k, uint32_t own_addr, uint32_t poll_addr, bool& aborted) {
  k += 1;
  asm volatile("s_mov_b64 exec, 1\n\t"
               "ds_write_b32 %0, %1\n\t"
               "s_mov_b64 exec, -1" ::"v"(own_addr), "v"(k) : "memory");
  if (aborted) return;
  for (uint32_t spin = 0;; ++spin) {
    uint32_t seen;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(poll_addr) : "memory");
    if (__builtin_amdgcn_ballot_w64((int32_t)(k - seen) > 0) == 0) break;
    if (spin > (1u << 22)) { aborted = true; break; }
  }
}

template <int BG> constexpr bool ext_shifts_are_zero() {
  using B = G<BG>;
  for (int ils = 0; ils < 8; ++ils)
    for (int e = 0; e < B::EDGES; ++e)
      if (B::col(e) >= B::CORE && B::shift(ils, e) != 0) return false;
  return true;
}

// ---- CRC24B over GF(2) (chancodebase.py:37-44, 83-128), usable at compile time
constexpr uint32_t CRC24B_LOW = 0x800063u;
constexpr uint32_t gf2_mulmod24(uint32_t a, uint32_t b) {      // (a * b) mod g, degrees < 24
  uint32_t r = 0;
  for (int i = 23; i >= 0; --i) {
    const uint32_t top = (r >> 23) & 1u;
    r = ((r << 1) & 0xFFFFFFu) ^ (top ? CRC24B_LOW : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
constexpr uint32_t gf2_xpow24(uint32_t e) {                    // x^e mod g
  uint32_t result = 1u, base = 2u;
  while (e) {
    if (e & 1u) result = gf2_mulmod24(result, base);
    base = gf2_mulmod24(base, base);
    e >>= 1;
  }
  return result;
}
// a * K mod g for a compile-time K: xor of the precomputed K * x^i mod g over the set bits of a
template <uint32_t K> __device__ __forceinline__ uint32_t gf2_mulc24(uint32_t a) {
  uint32_t r = 0;
  static_for<24>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    constexpr uint32_t ki = gf2_mulmod24(K, gf2_xpow24(i));
    r ^= (0u - ((a >> i) & 1u)) & ki;
  });
  return r;
}

// Fused front and back: rate recovery (ldpc.py:1330-1418, first transmission: rv 0, empty soft buffer,
// no wrap-around repetition) is done by the initial fill, which reads the LLRs straight from the demapper's
// (n_tb, llr_len) output -- written per code block in de-interleaved order by the demapper itself (nrx_qam_demap_* with
// the code-block geometry: its stores stay coalesced, and this kernel's loads become contiguous; gathering the
// symbol-major order here cost 3.6x the algorithmic HBM traffic, profiles/r2_decoder_traffic.json); the code-block CRC24B check and the merge into the transport block
// (ldpc.py:1584-1619) are done by the tail.  Geometry as nrx_ldpc_enc.hip's RmGeom.
struct FuseGeom {
  int C, e_small, n_small, f, qm, sys_len, F, llr_len, cb_len, payload;
  int rows_live;   // the row count the launch needs (<= RA of the instantiation that runs).  Unfused entry: the extension LLRs of the
                   // rows beyond it are taken as zero, which makes those rows the no-ops the workspace kernel does not run at all;
                   // MODE bit 3 (hybrids, fused whole decodes): the layers beyond it are left out
};
struct FuseArgs {
  FuseGeom g;
  uint8_t* tb_out;
  uint8_t* cb_ok;
  // FUSED, optional: decode only the code blocks sel[0 .. *n_sel) (indices into the n_tb * C blocks; the count is read on the
  // device, so the launch covers the worst case and needs no host read): the second pass of the two-pass schedule
  const int32_t* sel;
  const int32_t* n_sel;
  // FUSED, MODE 1 / 2: where a code block's decoder state is parked between two launches (see StateLay), indexed by code block
  double* state;
  // RC < RA (hybrid instantiations): the check-node state (pm1, pm2, extension posterior) of the rows >= RC streams through this
  // workspace, [workgroup][slot][row - RC][3][Zc] doubles
  double* ws;
  // MODE bit 2 (stages of the certified early exit): lam[2 cb], lam[2 cb + 1] = the largest |LLR| of the code block over every
  // received position that is not a filler, and over the core-parity + extension columns alone (written by the stage that fills
  // from the LLRs; nrx_ldpc_certify_f64 prices its error budget and the +1e5 quirk with them)
  double* lam;
  // MODE bit 2, optional (cert_w != NULL): the stability certificate is evaluated in the kernel's tail on the state the workgroup
  // still holds (nrx_ldpc_certcore.h); a block that holds it gets exit_iter[cb] = cp.iter_now and does NOT park.  cert_w = scratch for
  // the slack sums and the rows' slacks, (CORE + 2 ROWS) * Zc floats per code-block slot of the launch (grid * NS slots)
  float* cert_w;
  uint8_t* exit_iter;
  nrx_certcore::Params cp;
  // MODE bit 4 (the persistent certified schedule): queue[0] = the next work-list position (zeroed by the caller's launch), queue[1] = error
  // word (a slot barrier gave up); the iterations of each stage (the certificate is evaluated behind every stage but the last)
  int32_t* queue;
  int32_t stage_iters[4];
  int32_t n_stages;
};
// The kernel reads FuseArgs through the kernarg segment pointer at the two places that need it (initial fill, tail)
// instead of through its parameter: as a parameter its ten scalars and two pointers stay live across the whole layer
// loop, and at 168 VGPRs / 106 SGPRs that pushed 99 scratch accesses INTO the loop.
struct KernArgs {
  const double* llr; int n_cb; int n_iter; uint8_t* hard; const uint64_t* wtab; FuseArgs fa;
};
typedef const FuseArgs __attribute__((address_space(4))) * fargs_t;
__device__ __forceinline__ fargs_t fuse_args() {
  const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return (fargs_t)(ka + offsetof(KernArgs, fa));
}

// NS code blocks per workgroup (NS x Zc/64 waves).  Lane z of a code block's waves = check row z of every layer.
// NS = 2 (<= 168 VGPRs, three waves per SIMD) serves the truncated graphs.  (All 46 rows on chip was tried with NS = 1: their state,
// ~330 registers per lane, does not fit -- a 6-wave workgroup puts two waves on two of the SIMDs, so a wave gets 256 of the 512
// unified registers, the rest went to scratch and the kernel was slower than the workspace kernel, 146.6 against 137.5 ms.)
// FUSED = false: llr = rate-recovered LLRs (n_cb, N), hard = (n_cb, K) hard decisions.
// FUSED = true:  llr = demapper output (n_tb, llr_len), tb_out = (n_tb, C*payload) merged hard bits, cb_ok = (n_cb,).
// MODE (FUSED only), two bits: 0 = a whole decode.  Bit 0 (park): a block whose CRC fails at the end leaves its complete decoder
// state (posterior columns, check-node minima, sign / argmin words) in fa.state.  Bit 1 (resume): the initial fill is replaced by
// reading that state back, and n_iter more iterations follow.  park(n1), then resume(n2) on the failing blocks, IS one run of
// n1 + n2 iterations for them -- the continuation form of the multi-pass schedule: no pass repeats an earlier pass's iterations.
// Bit 2 (with bit 0): EVERY block parks its state, whatever its CRC says, and a launch that fills from the LLRs also leaves the
// block's two LLR maxima in fa.lam: the stages of the certified early exit (ldpc_certify_kernel reads the parked state).
// Bit 3 (alone: a whole decode): the layers beyond FuseGeom::rows_live are left out (a kernel-uniform test per layer) instead of run as
// no-ops -- the copies a launch takes when it needs fewer rows than its instantiation has.
// RC = rows whose check-node state stays in registers (default: all RA of them).  RC < RA: the HYBRID for more rows than fit --
// rates below ~0.6, HARQ retransmissions, all 46 rows: everything of this kernel (rotated rows, unit x pm under EXEC, DS
// immediates, the barrier plan) applies to every row, and the rows >= RC, the sparse ones, keep pm1 / pm2 / extension posterior in
// a workspace instead: read two streamed layers ahead into a small register ring, written back when the layer has them (sign /
// argmin words of all rows stay in registers).  Replaces the workspace kernel of nrx_ldpc_dec.hip for Zc = 384.
template <int BG, int ZI, int RA, bool FUSED, int NS = 2, int MODE = 0, int RC = RA>
__global__ void __launch_bounds__(NS * kZ.z[ZI], NS == 2 ? 3 : 1)
ldpc_dec_chip64_kernel(const double* __restrict__ llr, int n_cb, int n_iter, uint8_t* __restrict__ hard, mtab_t wtab,
                       FuseArgs /* read through fuse_args() */) {
  static_assert(ext_shifts_are_zero<BG>(), "extension columns are expected to be unshifted");
  using B = GR<BG, RA>;
  using Y = Lay<BG, RA>;
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
  static_assert(Y::plan_rot.ok, "barrier placement leaves a column hazard");
  constexpr int NEXT = Y::n_ext() > 0 ? Y::n_ext() : 1;

  double m1[B::ROWS], m2[B::ROWS];
  double rext[NEXT];                                       // posterior of each layer's extension column, element z
  uint32_t sgw[Y::n_wide() > 0 ? Y::n_wide() : 1];
  uint32_t sgn[(Y::n_narrow() + 1) / 2];                   // two 16-bit fields per word
  u32x2 uv = {0u, 0u};                                     // register pair of the +-1.0 / +-0.75 units: low word stays 0
  double c0 = 0.0;                                         // element z of column 0 (Lay::sigma: every layer meets it in its own lane)
  double f1 = 0.0;                                         // column 1 handed from a layer to its successor (Lay::fwd1)
  static_assert(RC == RA || (!FUSED && (MODE == 0 || MODE == 8) && RC >= 4 && RC < RA), "the hybrid is built for the unfused entry");
  static_assert((MODE & 8) == 0 || MODE == 8, "MODE bit 3 goes with a whole decode");
  constexpr bool HYB = RC < RA;
  // MODE bit 4 (with bits 0 and 2): the PERSISTENT certified schedule -- each code-block slot draws blocks from a device queue and takes a
  // block through all its stages (n iterations, CRC, certificate, go on or stop) before it draws the next; the two slots of a workgroup
  // never synchronise with each other (slot_barrier), nothing is parked between stages and nothing reloaded.
  constexpr bool PERS = (MODE & 16) != 0;
  static_assert(!PERS || (FUSED && NS == 2 && (MODE & 15) == 5 && RC == RA), "the persistent schedule is the stage mode of the fused entry");
  const uint32_t sb_own = SLOTBAR_BASE + 32u * (uint32_t)slot + 4u * (uint32_t)__builtin_amdgcn_readfirstlane(z >> 6);
  const uint32_t sb_poll = SLOTBAR_BASE + 32u * (uint32_t)slot + 4u * (uint32_t)((z & 63) % (ZC / 64));
  uint32_t bar_k = 0;
  bool aborted = false;
  auto BAR = [&]() __attribute__((always_inline)) {
    if constexpr (PERS) slot_barrier(bar_k, sb_own, sb_poll, aborted);
    else __syncthreads();
  };
  // MODE bit 3 (whole decodes: the hybrids and the fused entry): the layers beyond the caller's row count (FuseGeom::rows_live) are left
  // out instead of run as no-ops.  A separate instantiation: the test costs 1.5-3 % more VALU instructions per iteration (copies at the
  // joins), which a launch that runs every row need not pay.
  constexpr bool SKIPR = (MODE & 8) != 0;
  // layer L of a launch that needs the first n rows: a layer that takes column 1 from its predecessor's register (fwd1) runs as long as
  // the predecessor does -- the pair is left out together, column 1 then simply stays in LDS; the last layer is kept when it hands
  // column 1 to layer 0
  auto runs_at = [](int L, int n) constexpr -> bool {
    if (!SKIPR || !Y::has_ext(L)) return true;
    if (Y::fwd1(L)) return L <= n;
    if (Y::give1(L) && L == B::ROWS - 1) return true;
    return L < n;
  };
  constexpr int LAST = B::ROWS - 1;
  constexpr bool SKIPZ = MODE == 0 && !HYB && Y::has_ext(LAST) && !Y::fwd1(LAST) && !Y::give1(LAST) && !Y::fwd1(0);
  constexpr int PF = 2;                                    // streamed layers fetched ahead (1 / 3 / 4 measured slower, LAB_NOTES 4.1h)
  static_assert(!HYB || (RA - RC > PF && (RA - RC) % PF == 0), "the ring slot of a streamed layer is (L - RC) mod PF in every iteration");
  double pf_m1[PF] = {}, pf_m2[PF] = {}, pf_rx[PF] = {};
  typedef double __attribute__((address_space(1))) * gD;
  char* wsb = nullptr;                                     // this slot's streamed state
  if constexpr (HYB) wsb = (char*)fuse_args()->ws + ((size_t)blockIdx.x * NS + slot) * (size_t)(RA - RC) * 3 * ZC * sizeof(double);
  // (uniform base + compile-time offset) made an opaque SGPR pair first: `global_load/store v, v_lane_offset, s[base]`
  auto wsL = [](char* base, size_t off, uint32_t lane_off) __attribute__((always_inline)) -> gD {
    char* b = base + off;
    asm volatile("" : "+s"(b));
    return (gD)(b + (size_t)lane_off);
  };
  auto ws_off = [](int L, int a) constexpr -> size_t { return ((size_t)(L - RC) * 3 + a) * ZC * sizeof(double); };

  const int32_t* sel = nullptr;
  {
    const fargs_t fa0 = fuse_args();
    sel = fa0->sel;
    if (sel) n_cb = *fa0->n_sel;                            // (wave-uniform; a workgroup beyond the count leaves at once)
  }
  if constexpr (PERS) {      // progress words start at zero (the one workgroup barrier of this mode: every wave is here)
    if ((z & 63) == 0) *(volatile uint32_t*)((char*)Praw + sb_own) = 0u;
    __syncthreads();
  }
  for (int cb0 = blockIdx.x * NS; PERS || cb0 < n_cb; cb0 += gridDim.x * NS) {
    int cbi = cb0 + slot;                                   // position in the work list
    if constexpr (PERS) {      // this slot's next block: one lane draws, the slot's barrier hands the position to its six waves
      if (z == 0) {
        const int got = aborted ? n_cb : atomicAdd(fuse_args()->queue, 1);
        *(volatile uint32_t*)((char*)Praw + SLOTQ_BASE + 4u * (uint32_t)slot) = (uint32_t)got;
      }
      BAR();
      cbi = (int)*(volatile uint32_t*)((char*)Praw + SLOTQ_BASE + 4u * (uint32_t)slot);
      cbi = __builtin_amdgcn_readfirstlane(cbi);
      if (cbi >= n_cb || aborted) break;
    }
    int one = 1;
    asm volatile("" : "+s"(one));                          // keeps the per-layer `if (live)` a real branch (see dec2)
    const bool live = cbi < n_cb && one != 0;
    bool last_zero = false;                                 // SKIPZ: the last layer's extension LLRs are all zero in this wave
    const int cbl = live ? cbi : n_cb - 1;                  // (a wave without a code block loads an existing one)
    const int cbq = sel ? sel[cbl] : cbl;                   // the code block itself
    const int cb = cbq;
    const double* in = FUSED ? llr : llr + (size_t)cbq * N;
    int fE = 0, foff = 0;                                  // FUSED: E_r and the offset of the block in the LLR stream
    int rows_live = B::ROWS;
    if constexpr (!FUSED || SKIPR) rows_live = fuse_args()->g.rows_live;
    FuseGeom fg{};
    if constexpr (FUSED) {
      const fargs_t fa = fuse_args();
      fg.C = fa->g.C; fg.e_small = fa->g.e_small; fg.n_small = fa->g.n_small; fg.f = fa->g.f; fg.qm = fa->g.qm;
      fg.sys_len = fa->g.sys_len; fg.F = fa->g.F; fg.llr_len = fa->g.llr_len;
      const int t = cbq / fg.C, r = cbq - t * fg.C;
      if (r < fg.n_small) { fE = fg.e_small; foff = r * fg.e_small; }
      else { fE = fg.e_small + fg.f; foff = fg.n_small * fg.e_small + (r - fg.n_small) * fE; }
      in = llr + (size_t)t * fg.llr_len;
    }
    // Element z of the column that starts at code-word position p0 (rate-recovered, punctured code word), clipped like
    // ldpc.py:1536; + 0.0 turns -0.0 into +0.0.  FUSED: the block's E_r LLRs arrive DE-INTERLEAVED (buffer order, written
    // that way by nrx_qam_demap_cb_*: position e = q*(E_r/Qm) + s of ldpc.py:1390-1397), so consecutive lanes read
    // consecutive addresses.  Branch-free: every lane issues its load from a clamped address (all 35 loads of a lane are in
    // flight together) and the value is selected afterwards.
    // (zl: the lane index as a value the optimiser cannot see through, redefined for every code block: otherwise the
    //  per-lane, per-column addresses and flags below -- invariant over the code-block loop -- are hoisted in front of it,
    //  35 of them, spilled to scratch there, and come back one dependent round trip at a time: ~100 serial memory round
    //  trips per code block)
    int zl0 = z;
    asm volatile("" : "+v"(zl0));
    const int zl = zl0;
    // rot: the element wanted is (z + rot) mod Zc (a layer's extension column under the layer's row rotation, Lay::sigma)
    // Two phases: every load of the fill first (35 per lane, all in flight together), the selection afterwards.  Written as
    // one expression per column the compiler put a branch around each column's clip ("no lane of the wave received this
    // column"), and with it a full wait behind each load: 35 dependent round trips per code block, 0.6 ms per 256 slots.
    auto elem = [&](int rot) __attribute__((always_inline)) -> int {
      int zr = zl0 + rot;
      zr -= zr >= ZC ? ZC : 0;
      return zr;
    };
    auto addr = [&](int p0, int rot) __attribute__((always_inline)) -> int {
      const int zr = elem(rot);
      if constexpr (!FUSED) {
        return p0 + zr;
      } else {
        // lanes [0, a): before the fillers (buffer position p0 + z); [a, b): fillers; [b, Zc): behind them (p0 - F + z);
        // transmitted iff the buffer position is < E_r, i.e. z < t1 resp. z < t2 -- four wave-uniform thresholds per column
        const int a = fg.sys_len - p0, b = a + fg.F, t1 = fE - p0, t2 = t1 + fg.F;
        const bool use1 = zr < a && zr < t1, use2 = zr >= b && zr < t2;
        return foff + (use1 ? p0 + zr : (use2 ? p0 - fg.F + zr : 0));     // (a lane without a value loads the block's first one)
      }
    };
    auto value = [&](double x, int p0, int rot) __attribute__((always_inline)) -> double {
      const double v = clip10(x) + 0.0;
      if constexpr (!FUSED) {
        return v;
      } else {
        // (the selection is recomputed from an opaque copy of the lane index: otherwise the flags of all 35 columns are kept
        //  alive across the loads)
        int zz = elem(rot);
        asm("" : "+v"(zz));
        const int a = fg.sys_len - p0, b = a + fg.F, t1 = fE - p0, t2 = t1 + fg.F;
        const bool sent = (zz < a && zz < t1) || (zz >= b && zz < t2);
        const bool filler = zz >= a && zz < b;                             // LARGE_LLR 1e20, clipped (ldpc.py:1414-1418)
        const double w = sent ? v : 0.0;
        return filler ? 1e10 : w;
      }
    };
    // ---- load: prepend the two punctured columns as zeros (ldpc.py:1536-1538)
    if constexpr (FUSED && (MODE & 2)) {
      using SL = StateLay<BG, RA>;
      const double* st = fuse_args()->state + (size_t)cbq * SL::NF * ZC + zl;
      static_for<B::CORE - 1>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value + 1;
        Ps[c * ZS + zl] = *(&st[(size_t)(SL::COL + c - 1) * ZC]);
      });
      c0 = *(&st[(size_t)SL::C0 * ZC]);
      f1 = *(&st[(size_t)SL::F1 * ZC]);
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        m1[L] = *(&st[(size_t)(SL::M1 + L) * ZC]);
        m2[L] = *(&st[(size_t)(SL::M2 + L) * ZC]);
        if constexpr (Y::has_ext(L)) rext[Y::ext_idx(L)] = *(&st[(size_t)(SL::REXT + Y::ext_idx(L)) * ZC]);
      });
      static_for<SL::NW>([&](auto i) __attribute__((always_inline)) {
        sgw[decltype(i)::value] = (uint32_t)__double_as_longlong(*(&st[(size_t)(SL::WORDS + decltype(i)::value) * ZC]));
      });
      static_for<SL::NN>([&](auto i) __attribute__((always_inline)) {
        sgn[decltype(i)::value] = (uint32_t)__double_as_longlong(*(&st[(size_t)(SL::WORDS + SL::NW + decltype(i)::value) * ZC]));
      });
    } else {
      double xs[B::CORE - 2 + NEXT];      // (the hybrid only touches the entries of its on-chip rows)
      static_for<B::CORE - 2>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        xs[c] = *(&in[addr(c * ZC, 0)]);
      });
      static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        if constexpr (Y::has_ext(L) && L < RC) xs[B::CORE - 2 + Y::ext_idx(L)] = *(&in[addr((Y::ext_col(L) - 2) * ZC, Y::sigma(ILS, ZC, L))]);   // element of row (z + sigma_L)
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (HYB) {       // streamed rows: zero minima, the extension LLR (or 0 beyond the rows that run) straight to the workspace
        const uint32_t zo8 = 8u * (uint32_t)zl;
        static_for<RA - RC>([&](auto lc) __attribute__((always_inline)) {
          constexpr int L = RC + decltype(lc)::value;
          *wsL(wsb, ws_off(L, 0), zo8) = 0.0;
          *wsL(wsb, ws_off(L, 1), zo8) = 0.0;
          if constexpr (Y::has_ext(L)) {
            const double e = clip10(in[addr((Y::ext_col(L) - 2) * ZC, Y::sigma(ILS, ZC, L))]) + 0.0;
            *wsL(wsb, ws_off(L, 2), zo8) = L < rows_live ? e : 0.0;
          }
        });
      }
      Ps[1 * ZS + zl] = 0.0;
      double lmax_all = 0.0, lmax_pe = 0.0;      // MODE bit 2: largest |LLR| that is not a filler (all columns / parity + extension columns)
      static_for<B::CORE - 2>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        const double v = value(xs[c], c * ZC, 0);
        Ps[(c + 2) * ZS + zl] = v;
        if constexpr (FUSED && (MODE & 4) != 0) {
          const bool filler = elem(0) >= fg.sys_len - c * ZC && elem(0) < fg.sys_len - c * ZC + fg.F;
          const double av = filler ? 0.0 : __builtin_fabs(v);
          lmax_all = __builtin_fmax(lmax_all, av);
          if constexpr (c + 2 >= B::KB) lmax_pe = __builtin_fmax(lmax_pe, av);
        }
      });
      static_for<RC>([&](auto lc) __attribute__((always_inline)) {
        constexpr int L = decltype(lc)::value;
        m1[L] = 0.0;
        m2[L] = 0.0;
        if constexpr (Y::has_ext(L)) {
          rext[Y::ext_idx(L)] = value(xs[B::CORE - 2 + Y::ext_idx(L)], (Y::ext_col(L) - 2) * ZC, Y::sigma(ILS, ZC, L));
          if constexpr (!FUSED) rext[Y::ext_idx(L)] = L < rows_live ? rext[Y::ext_idx(L)] : 0.0;      // (wave-uniform)
          if constexpr (SKIPZ && L == LAST) last_zero = __builtin_amdgcn_ballot_w64(rext[Y::ext_idx(L)] != 0.0) == 0;
          if constexpr (FUSED && (MODE & 4) != 0) {
            lmax_all = __builtin_fmax(lmax_all, __builtin_fabs(rext[Y::ext_idx(L)]));
            lmax_pe = __builtin_fmax(lmax_pe, __builtin_fabs(rext[Y::ext_idx(L)]));
          }
        }
      });
      if constexpr (FUSED && (MODE & 4) != 0) {      // block maxima: shuffles inside the wave, the waves joined behind the fill's barrier
        for (int k = 1; k < 64; k <<= 1) {
          lmax_all = __builtin_fmax(lmax_all, __shfl_xor(lmax_all, k, 64));
          lmax_pe = __builtin_fmax(lmax_pe, __shfl_xor(lmax_pe, k, 64));
        }
        double* redd = Praw + 16 + slot * 2 * (ZC / 64);      // (the padding in front of the columns is never addressed by a layer)
        if ((zl & 63) == 0) {
          redd[2 * (zl >> 6)] = lmax_all;
          redd[2 * (zl >> 6) + 1] = lmax_pe;
        }
      }
    }
    if constexpr (!(FUSED && (MODE & 2))) {
      c0 = 0.0;                                            // punctured column (ldpc.py:1536-1538)
      static_for<(Y::n_wide() > 0 ? Y::n_wide() : 1)>([&](auto i) __attribute__((always_inline)) { sgw[decltype(i)::value] = 0u; });
      static_for<(Y::n_narrow() + 1) / 2>([&](auto i) __attribute__((always_inline)) { sgn[decltype(i)::value] = 0u; });
    }
    BAR();
    if constexpr (FUSED && (MODE & 4) != 0 && (MODE & 2) == 0) {
      if (z == 0 && live) {
        const double* redd = Praw + 16 + slot * 2 * (ZC / 64);
        double a = 0.0, b = 0.0;
        for (int w = 0; w < ZC / 64; ++w) {
          a = __builtin_fmax(a, redd[2 * w]);
          b = __builtin_fmax(b, redd[2 * w + 1]);
        }
        double* lam = fuse_args()->lam;
        lam[2 * (size_t)cb] = a;
        lam[2 * (size_t)cb + 1] = b;
      }
    }

    if constexpr (HYB) {      // (the barrier above waited for the fill's stores)
      int zq = z;
      asm volatile("" : "+v"(zq));
      static_for<PF>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, Lk = RC + k;
        pf_m1[k] = *wsL(wsb, ws_off(Lk, 0), 8u * (uint32_t)zq);
        pf_m2[k] = *wsL(wsb, ws_off(Lk, 1), 8u * (uint32_t)zq);
        if constexpr (Y::has_ext(Lk)) pf_rx[k] = *wsL(wsb, ws_off(Lk, 2), 8u * (uint32_t)zq);
      });
    }
    // wrap masks of the layer about to run (SGPR pairs)
    uint64_t wcur[19];
    static_for<(Y::has_ext(0) ? Y::deg(0) - 1 : Y::deg(0))>([&](auto jc) __attribute__((always_inline)) {
      wcur[decltype(jc)::value] = wm[B::row_start(0) + decltype(jc)::value];
    });

    // SL: the copy of the iteration loop without the last layer's body (see SKIPZ)
    auto iter_loop = [&](auto slc, int n_it) __attribute__((always_inline)) {
    constexpr bool SL = decltype(slc)::value;
    for (int it = 0; it < n_it; ++it) {
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
            pf_m1[k] = *wsL(wsb, ws_off(Lp, 0), zo8);
            pf_m2[k] = *wsL(wsb, ws_off(Lp, 1), zo8);
            if constexpr (Y::has_ext(Lp)) pf_rx[k] = *wsL(wsb, ws_off(Lp, 2), zo8);
          }
        }
        // hybrid: a layer beyond the caller's row count has all-zero extension LLRs -- every row of it the exact no-op of DESIGN 4.2a --
        // and is left out (kernel-uniform test; its barrier and the next layer's mask loads stay)
        bool runs = live;
        if constexpr (SKIPR) runs = live && runs_at(L, rows_live);
        // byte addresses of element z of column 0 of this slot: plain, wrapped (- Zc), and both + HI
        const uint32_t zb = zbo, zbw = zbo - zc8, zbh = zbo + HI, zbwh = zbo - zc8 + HI;
        if constexpr (SL && L == LAST) {
          if (live) {      // the layer is left out; the next layer's masks are still wanted
            constexpr int Ln = (L + 1) % B::ROWS;
            constexpr int DCn = Y::has_ext(Ln) ? Y::deg(Ln) - 1 : Y::deg(Ln);
            static_for<DCn>([&](auto jc) __attribute__((always_inline)) {
              wcur[decltype(jc)::value] = wml[B::row_start(Ln) + decltype(jc)::value];
            });
          }
        } else if (__builtin_expect(runs, 1)) {
          double t[D];
          // priority steps at about a quarter, a half and three quarters of the layer's VALU work (5 + 4 + 5 per edge)
          constexpr int PQ1 = (7 * D) / 10 < D - 1 ? (7 * D) / 10 : D - 1, PQ2 = D / 2 < 2 ? 2 : D / 2, PQ3 = (3 * D) / 10 < 1 ? 1 : (3 * D) / 10;
          LAYER_PRIO(Y::prio_q(L, 0));
          uint32_t ad[DC > 0 ? DC : 1];      // LDS byte address (without the immediate) of each core edge's element
          // ---- pass 1a: issue every LDS read of the layer, first edge first (the order pass 1b consumes them in).  Column 0
          // is the lane's own register; a handed-over column 1 comes from the predecessor's register (Lay::sigma, Lay::fwd1).
          static_for<DC>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            constexpr int col = B::col(E0 + j);
            if constexpr (col == 0) {
              t[j] = c0;
            } else if constexpr (col == 1 && Y::fwd1(L)) {
              t[j] = f1;
            } else {
              constexpr uint32_t off = 8u * (uint32_t)(col * ZS + Y::eff_shift(ILS, ZC, L, E0 + j));
              const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);            // z + shift >= Zc
              if constexpr (off < 65536) { ad[j] = wraps ? zbw : zb; t[j] = *(const double*)((const char*)Praw + ad[j] + off); }
              else { ad[j] = wraps ? zbwh : zbh; t[j] = *(const double*)((const char*)Praw + ad[j] + (off - HI)); }
            }
          });
          __builtin_amdgcn_sched_barrier(0);
          // ---- old state: pm1 / pm2 (scaled minima carrying the row parity), and the sign / argmin word: argmin in the low
          // bits of its field, above it the signs of the t_j of the previous iteration, edge 0 highest
          double om1, om2;
          if constexpr (L < RC) { om1 = m1[L]; om2 = m2[L]; } else { om1 = cm1; om2 = cm2; }
          uint32_t word, oidx;
          int top;   // left shift that brings the highest sign bit of the field to bit 31
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
          // ---- pass 1b: t_j = r_j - msg_old_j = fma(-u_j, argmin ? pm2 : pm1, r_j)  (ldpc.py:1550-1553); the extension
          // column's r comes from its register
          if constexpr (EXT) t[D - 1] = L < RC ? rext[Y::ext_idx(L)] : crx;
          uint32_t wrun = word << (top - (D - 1));           // sign of edge 0 at bit 31; doubled per edge
          static_for<D>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const double u = unit_of(uv, wrun);
            if constexpr (j < D - 1) wrun = dbl(wrun);
            if constexpr (j == PQ1 && Y::prio_q(L, 1) != Y::prio_q(L, 0)) LAYER_PRIO(Y::prio_q(L, 1));
            t[j] = sel_fnma_x<j>(t[j], oidx, u, om1, om2);
          });
          // ---- min-sum (ldpc.py:1556-1564): two smallest magnitudes by min/max; the signs of the t_j are collected (edge 0
          // ends highest) and their parity is a population count
          static_assert(D >= 2, "a check row has at least two edges");
          double a1 = vmin_abs2(t[0], t[1]);
          double a2 = vmax_abs2(t[0], t[1]);
          uint32_t nsg = __builtin_amdgcn_alignbit(hi32(t[0]) >> 31, hi32(t[1]), 31);
          // a2 = min(a2, max(a1, |t_j|)); a1 = min(a1, |t_j|); nsg = (nsg << 1) | sign(t_j)  for j = 2 .. D - 1, in blocks of 4 / 2 / 1
          static_for<D - 2>([&](auto jc) __attribute__((always_inline)) {
            constexpr int i = decltype(jc)::value, j = i + 2, left = D - 2 - (i / 4) * 4;      // edges left at the start of this group of four
            if constexpr (j == PQ2 && Y::prio_q(L, 2) != Y::prio_q(L, 1)) LAYER_PRIO(Y::prio_q(L, 2));
            if constexpr (i % 4 == 0 && left >= 4) minsum_step4(a1, a2, nsg, t[j], t[j + 1], t[j + 2], t[j + 3]);
            else if constexpr (left < 4 && i % 4 == 0 && left >= 2) minsum_step2(a1, a2, nsg, t[j], t[j + 1]);
            else if constexpr (left < 4 && ((left == 1 && i % 4 == 0) || (left == 3 && i % 4 == 2))) minsum_step(a1, a2, nsg, t[j]);
          });
          // QUIRK ldpc.py:1563: min2 = min(min2, |v_argmin + 1e5|) with the SIGNED argmin entry; it can only win where
          // min2 > 5e4 (filler / saturated LLRs): wave-uniform cold path.
          bool tie_quirk = false;     // wave-uniform: some lane has min2 moved UNDER a tied minimum (see below)
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(a2 > 5.0e4) != 0, 0)) {
            double v = t[D - 1];
            static_for<D - 1>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 2 - decltype(jc)::value;
              // (compared as bit patterns: the same test as pass 2's |t_j| == min1 for these finite values, but a different
              //  instruction -- written as the same float compare, the compiler hoists all D of them above this branch to
              //  share them with pass 2 and then cannot keep that many SGPR pairs alive)
              v = (__double_as_longlong(t[j]) & 0x7fffffffffffffffll) == __double_as_longlong(a1) ? t[j] : v;   // ends at the first index holding the minimum
            });
            const double q = __builtin_fabs(v + 100000.0);
            // With several entries AT the minimum (min2 == min1) and a negative first one, q < min1: the reference then gives
            // min2 = q to np.argmin's entry ALONE and min1 to the other tied ones (ldpc.py:1566-1570).  The hot write pass gives
            // pm2 to every tied entry (identical while min2 == min1), so such a layer takes the first-only write pass below.
            tie_quirk = __builtin_amdgcn_ballot_w64(q < a2 && a2 == a1) != 0;
            a2 = q < a2 ? q : a2;
          }
          // 0.75 * min (ldpc.py:1573; the scale commutes with the sign) times 2^7 (see unit_of), with the row parity as its
          // sign: +-96 is built from the population count of the collected signs
          const double c96 = c96_of(uv, nsg);
          const double nm1 = a1 * c96, nm2 = a2 * c96;
          if constexpr (L < RC) {
            m1[L] = nm1;
            m2[L] = nm2;
          } else {
            *wsL(wsb, ws_off(L, 0), zo8) = nm1;
            *wsL(wsb, ws_off(L, 1), zo8) = nm2;
          }
          // ---- pass 2: r_j = t_j + msg_new_j = fma(u_j, first argmin ? pm2 : pm1, t_j), written back to the element it was
          // read from.  The FIRST entry equal to min1 gets min2 (np.argmin, ldpc.py:1558-1570).
          uint32_t idx = 0;
          auto put = [&](auto jc2) __attribute__((always_inline)) {      // r_j back to its column element / register
            constexpr int j = decltype(jc2)::value;
            constexpr int col = B::col(E0 + j);
            if constexpr (col == 0) {
              c0 = t[j];
            } else if constexpr (col == 1 && Y::give1(L)) {
              f1 = t[j];                                    // the next layer takes it from here and writes the column itself
            } else if constexpr (col < B::CORE) {
              constexpr uint32_t off = 8u * (uint32_t)(col * ZS + Y::eff_shift(ILS, ZC, L, E0 + j));
              const bool wraps = __builtin_amdgcn_inverse_ballot_w64(wcur[j]);
              if constexpr (off < 65536) *(double*)((char*)Praw + (wraps ? zbw : zb) + off) = t[j];
              else *(double*)((char*)Praw + (wraps ? zbwh : zbh) + (off - HI)) = t[j];
            } else {
              if constexpr (L < RC) rext[Y::ext_idx(L)] = t[j];
              else *wsL(wsb, ws_off(L, 2), zo8) = t[j];
            }
          };
          if (__builtin_expect(!tie_quirk, 1)) {
            static_for<D>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = D - 1 - decltype(jc)::value;      // last edge first: idx ends at the first index holding the minimum
              if constexpr (decltype(jc)::value == PQ3 && Y::prio_q(L, 3) != Y::prio_q(L, 2)) LAYER_PRIO(Y::prio_q(L, 3));
              const double u = unit_of_ordered(uv, hi32(t[j]));
              if constexpr (j < D - 1) put(std::integral_constant<int, j + 1>{});      // (the write of the edge before: between the unit and its use)
              t[j] = sel_fma_x<j>(t[j], idx, a1, u, nm1, nm2);
            });
            put(std::integral_constant<int, 0>{});
          } else {
            uint64_t seen = 0;
            static_for<D>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              const uint64_t is_min = cmp_abs_eq(t[j], a1);
              const double u = unit_of(uv, hi32(t[j]));
              sel_fma_first<j>(t[j], idx, seen, is_min, u, nm1, nm2);
              put(std::integral_constant<int, j>{});
            });
          }
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
        } else if constexpr (SKIPR) {
          if (live) {      // the layer is left out; the next layer's masks are still wanted
            constexpr int Ln = (L + 1) % B::ROWS;
            constexpr int DCn = Y::has_ext(Ln) ? Y::deg(Ln) - 1 : Y::deg(Ln);
            static_for<DCn>([&](auto jc) __attribute__((always_inline)) {
              wcur[decltype(jc)::value] = wml[B::row_start(Ln) + decltype(jc)::value];
            });
          }
        }
        if constexpr (Y::plan_rot.need[(L + 1) % B::ROWS]) {
          BAR();
        }
        __builtin_amdgcn_sched_barrier(0);   // nothing migrates between layers (register pressure)
      });
    }
    };
    // PERS: the block's stages, one after the other (the certificate behind every stage but the last decides whether it goes on)
    int iters_done = 0;
    for (int stage = 0;; ++stage) {
    int n_it = n_iter;
    bool last_stage = true;
    if constexpr (PERS) {
      const fargs_t fq = fuse_args();
      n_it = fq->stage_iters[stage];
      last_stage = stage + 1 >= fq->n_stages;
    }
    iters_done += n_it;
    if constexpr (SKIPZ) {
      if (last_zero) iter_loop(std::true_type{}, n_it);
      else iter_loop(std::false_type{}, n_it);
    } else {
      iter_loop(std::false_type{}, n_it);
    }
    BAR();

    // ---- hard decisions of the information columns (ldpc.py:1578-1581)
    if constexpr (!FUSED) {
      int zh = z;
      asm volatile("" : "+v"(zh));
      if (live) {
        static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          if constexpr (c == 0) hard[(size_t)cb * K + zh] = c0 < 0.0 ? 1 : 0;
          else hard[(size_t)cb * K + c * ZC + zh] = Ps[c * ZS + zh] < 0.0 ? 1 : 0;
        });
      }
    } else {
      // ... merged into the transport block (fillers and the code-block CRC stripped, ldpc.py:1600-1619), and the CRC24B
      // check of the block: bit n = c*Zc + z contributes x^((KB-1-c)*Zc + (Zc-1-z)) mod g, i.e. lane z xors the
      // compile-time constants x^((KB-1-c)*Zc) of its set bits and the lanes are then joined as a polynomial in x
      // (lane z weighs x^(Zc-1-z)).  The common factor x^(...) against the reference's long division (chancodebase.py:
      // 119-128) is invertible mod g, so the remainder is zero for exactly the same bit strings.
      uint32_t v = 0;
      int zt = z;
      asm volatile("" : "+v"(zt));                          // (as zl above: nothing of the tail is computed in front of the loop)
      const int cbm = live ? cb : 0;
      const fargs_t fa = fuse_args();
      const int payload = fa->g.payload, cb_len = fa->g.cb_len;
      uint8_t* dst = fa->tb_out + (size_t)cbm * payload;   // (t*C + r) * payload
      static_for<B::KB>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value;
        constexpr uint32_t w = gf2_xpow24((uint32_t)((B::KB - 1 - c) * ZC));
        const uint32_t bit = (c == 0 ? c0 : Ps[c * ZS + zt]) < 0.0 ? 1u : 0u;
        if (live && zt < payload - c * ZC) dst[c * ZC + zt] = (uint8_t)bit;    // (thresholds are wave-uniform)
        v ^= (zt < cb_len - c * ZC && bit) ? w : 0u;
      });
      // join inside the wave: level k pairs blocks of 2^k lanes, left * x^(2^k) + right
      static_for<6>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const uint32_t other = (uint32_t)__shfl_xor((int)v, 1 << k, 64);
        const bool upper = (zt >> k) & 1;
        v = gf2_mulc24<gf2_xpow24(1u << k)>(upper ? other : v) ^ (upper ? v : other);
      });
      uint32_t* red = (uint32_t*)Praw;                      // the padding in front of the columns is never addressed
      if ((zt & 63) == 0) red[slot * (ZC / 64) + (zt >> 6)] = v;
      BAR();
      if (zt == 0 && live) {
        uint32_t tot = 0;
        for (int w = 0; w < ZC / 64; ++w) tot = gf2_mulc24<gf2_xpow24(64)>(tot) ^ red[slot * (ZC / 64) + w];
        fa->cb_ok[cb] = tot == 0 ? 1 : 0;
        if constexpr ((MODE & 1) != 0) red[NS * (ZC / 64) + slot] = tot == 0 ? 1u : 0u;
      }
      if constexpr ((MODE & 1) != 0) {
        BAR();
        // MODE bit 2: every block parks (whatever its CRC says); where asked for (cert_w != NULL) the stability certificate is evaluated
        // on the spot in between -- posterior columns still in LDS, the check-node state read back from what this lane just parked --
        // and a block that holds it gets its exit iteration (the caller's work list of the next stage leaves it out) and keeps its
        // columns to itself.
        // (the check-node state first: it is all the certificate reads back; the posterior columns follow behind the certificate, and
        //  only for the blocks that have to go on)
        const bool parks = live && !PERS && ((MODE & 4) != 0 || red[NS * (ZC / 64) + slot] == 0u);
        const bool cert_here = (MODE & 4) != 0 && fa->cert_w != nullptr;      // (kernel-uniform)
        auto park_columns = [&]() __attribute__((always_inline)) {
          using SL = StateLay<BG, RA>;
          double* st = fa->state + (size_t)cb * SL::NF * ZC + zt;
          static_for<B::CORE - 1>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value + 1;
            st[(size_t)(SL::COL + c - 1) * ZC] = Ps[c * ZS + zt];
          });
          st[(size_t)SL::F1 * ZC] = f1;
        };
        // the check-node state of this lane to / from a state record (StateLay): where the certificate reads it back
        auto park_rows = [&](double* st) __attribute__((always_inline)) {
          using SL = StateLay<BG, RA>;
          st[(size_t)SL::C0 * ZC] = c0;
          static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
            constexpr int L = decltype(lc)::value;
            st[(size_t)(SL::M1 + L) * ZC] = m1[L];
            st[(size_t)(SL::M2 + L) * ZC] = m2[L];
            if constexpr (Y::has_ext(L)) st[(size_t)(SL::REXT + Y::ext_idx(L)) * ZC] = rext[Y::ext_idx(L)];
          });
          static_for<SL::NW>([&](auto i) __attribute__((always_inline)) {
            st[(size_t)(SL::WORDS + decltype(i)::value) * ZC] = __longlong_as_double((long long)sgw[decltype(i)::value]);
          });
          static_for<SL::NN>([&](auto i) __attribute__((always_inline)) {
            st[(size_t)(SL::WORDS + SL::NW + decltype(i)::value) * ZC] = __longlong_as_double((long long)sgn[decltype(i)::value]);
          });
        };
        if (parks) {      // park the state for the continuation launch
          using SL = StateLay<BG, RA>;
          park_rows(fa->state + (size_t)cb * SL::NF * ZC + zt);
          if (!cert_here) park_columns();
        }
        if constexpr ((MODE & 4) != 0 && !PERS) {
          if (fa->cert_w != nullptr) {             // (kernel-uniform)
            using SL = StateLay<BG, RA>;
            nrx_certcore::Params cp;
            cp.gamma = fa->cp.gamma; cp.gamma1 = fa->cp.gamma1; cp.dmax = fa->cp.dmax; cp.n_iter_total = fa->cp.n_iter_total;
            cp.max_sweeps = fa->cp.max_sweeps; cp.flags = fa->cp.flags; cp.iter_now = fa->cp.iter_now;
            const bool crc = red[NS * (ZC / 64) + slot] != 0u;
            const double* lam = fa->lam;
            const double la = __hip_atomic_load(lam + 2 * (size_t)cbm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double lp = __hip_atomic_load(lam + 2 * (size_t)cbm + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float* Wg = fa->cert_w + ((size_t)blockIdx.x * NS + slot) * (size_t)((B::CORE + 2 * B::ROWS) * ZC);
            const double* stc = fa->state + (size_t)cbm * SL::NF * ZC + zt;
            const double c0c = stc[(size_t)SL::C0 * ZC];
            __syncthreads();                       // (the CRC flags have been read: the padding words are free again)
            const bool want = live && (crc || (cp.flags & 4) != 0);
            const nrx_certcore::lds_cd Pc = (nrx_certcore::lds_cd)(Praw + ZS + slot * BUF);
            auto wgbar = []() __attribute__((always_inline)) { __syncthreads(); };
            // (with filler bits the row pass also has to recognise their posteriors: 7 more instructions per edge)
            const bool settled = fa->g.F > 0 ? nrx_certcore::certify_on_chip<BG, ZI, RA, true, NS>(cp, la, lp, stc, c0c, Pc, Wg, zt, slot, want, (uint32_t*)(Praw + 64), wgbar)
                                             : nrx_certcore::certify_on_chip<BG, ZI, RA, false, NS>(cp, la, lp, stc, c0c, Pc, Wg, zt, slot, want, (uint32_t*)(Praw + 64), wgbar);
            if (settled && zt == 0) fa->exit_iter[cb] = (uint8_t)(cp.iter_now < 255 ? cp.iter_now : 255);
            if (parks && !settled) park_columns();
          }
        }
        if constexpr (PERS) {
          // The block stops here if this was its last stage, or if its CRC passes and its frozen state holds the certificate; else it
          // goes on with the next stage's iterations.  The certificate reads the check-node state back from this SLOT's record (with the
          // state live in registers across it, the certificate's own registers spill into the layer loop), and a block that goes on takes
          // it back from there.  Nothing else leaves the chip: no parked columns, no resume.
          using SL = StateLay<BG, RA>;
          const bool crc = red[NS * (ZC / 64) + slot] != 0u;
          const int32_t cflags = fa->cp.flags;
          bool stop = last_stage;
          if (!last_stage) {
            // (EVERY block that is not at its last stage leaves its check-node state in the slot's record and takes it back below: one
            //  definition of the state in front of the next stage's loop, whichever way the block got there, and no state live across the
            //  certificate)
            double* stc = fa->state + ((size_t)blockIdx.x * NS + slot) * SL::NF * ZC + zt;
            park_rows(stc);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (the record is written before anything reads it back)
            bool settled = false;
            BAR();                                 // (the CRC flags have been read: the padding words are free again)
            if (crc || (cflags & 4) != 0) {        // (slot-uniform)
              nrx_certcore::Params cp;
              cp.gamma = fa->cp.gamma; cp.gamma1 = fa->cp.gamma1; cp.dmax = fa->cp.dmax; cp.n_iter_total = fa->cp.n_iter_total;
              cp.max_sweeps = fa->cp.max_sweeps; cp.flags = cflags; cp.iter_now = iters_done;
              const double* lam = fa->lam;
              const double la = __hip_atomic_load(lam + 2 * (size_t)cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const double lp = __hip_atomic_load(lam + 2 * (size_t)cb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              float* Wg = fa->cert_w + ((size_t)blockIdx.x * NS + slot) * (size_t)((B::CORE + 2 * B::ROWS) * ZC);
              const double c0c = stc[(size_t)SL::C0 * ZC];
              const nrx_certcore::lds_cd Pc = (nrx_certcore::lds_cd)(Praw + ZS + slot * BUF);
              settled = fa->g.F > 0 ? nrx_certcore::certify_on_chip<BG, ZI, RA, true, 1>(cp, la, lp, stc, c0c, Pc, Wg, zt, slot, true, (uint32_t*)(Praw + 64), BAR)
                                    : nrx_certcore::certify_on_chip<BG, ZI, RA, false, 1>(cp, la, lp, stc, c0c, Pc, Wg, zt, slot, true, (uint32_t*)(Praw + 64), BAR);
            }
            if (settled) {
              if (zt == 0) fa->exit_iter[cb] = (uint8_t)(iters_done < 255 ? iters_done : 255);
              stop = true;
            } else {      // take the check-node state back (what this lane wrote itself)
              const double* sq = stc;
              asm volatile("" : "+v"(sq));
              c0 = sq[(size_t)SL::C0 * ZC];
              static_for<B::ROWS>([&](auto lc) __attribute__((always_inline)) {
                constexpr int L = decltype(lc)::value;
                m1[L] = sq[(size_t)(SL::M1 + L) * ZC];
                m2[L] = sq[(size_t)(SL::M2 + L) * ZC];
                if constexpr (Y::has_ext(L)) rext[Y::ext_idx(L)] = sq[(size_t)(SL::REXT + Y::ext_idx(L)) * ZC];
              });
              static_for<SL::NW>([&](auto i) __attribute__((always_inline)) {
                sgw[decltype(i)::value] = (uint32_t)__double_as_longlong(sq[(size_t)(SL::WORDS + decltype(i)::value) * ZC]);
              });
              static_for<SL::NN>([&](auto i) __attribute__((always_inline)) {
                sgn[decltype(i)::value] = (uint32_t)__double_as_longlong(sq[(size_t)(SL::WORDS + SL::NW + decltype(i)::value) * ZC]);
              });
            }
          }
          if (stop) break;
          BAR();                                   // (the padding words of the tail are free before the next stage's tail writes them)
          continue;
        }
      }
    }
    break;      // (every other mode: one stage)
    }
    BAR();
  }
  if constexpr (PERS) {
    if (aborted && (z & 63) == 0) atomicOr(fuse_args()->queue + 1, 1);
  }
}

__constant__ WrapTab kWrap1_384_r13 = make_wrap<1, zindex_c(384), 13>();
__constant__ WrapTab kWrap1_384_r15 = make_wrap<1, zindex_c(384), 15>();
__constant__ WrapTab kWrap1_384_r31 = make_wrap<1, zindex_c(384), 31>();
__constant__ WrapTab kWrap1_384_r46 = make_wrap<1, zindex_c(384), 46>();
// rows of the hybrid instantiations whose state stays in registers (RA - RC a multiple of the prefetch depth)
constexpr int HYB_RC46 = 12, HYB_RC31 = 11;      // resident rows of the 46- / 31-row hybrids

struct DevTab { const uint64_t* p[4]; bool ok; };

// wrap-mask table of the instantiation that serves n_rows (13 or 15 rows), resolved per device
int32_t wrap_table(int n_rows, const uint64_t** out) {
  static DevTab tabs[16] = {};
  static std::mutex mu;
  int dev = 0;
  NRX_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16, NRX_E_HIP, "nrx_ldpc_decode_f64: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(mu);
  DevTab& dt = tabs[dev];
  if (!dt.ok) {
    void* p[4] = {};
    const hipError_t e[4] = {hipGetSymbolAddress(&p[0], HIP_SYMBOL(kWrap1_384_r13)), hipGetSymbolAddress(&p[1], HIP_SYMBOL(kWrap1_384_r15)),
                             hipGetSymbolAddress(&p[2], HIP_SYMBOL(kWrap1_384_r31)), hipGetSymbolAddress(&p[3], HIP_SYMBOL(kWrap1_384_r46))};
    for (int i = 0; i < 4; ++i)
      NRX_REQUIRE(e[i] == hipSuccess && p[i], NRX_E_HIP, "nrx_ldpc_decode_f64: hipGetSymbolAddress(wrap masks) failed");
    for (int i = 0; i < 4; ++i) dt.p[i] = (const uint64_t*)p[i];
    dt.ok = true;
  }
  *out = dt.p[n_rows <= 13 ? 0 : (n_rows <= 15 ? 1 : (n_rows <= 31 ? 2 : 3))];
  return NRX_OK;
}

bool chip64_covers(const nrx_ldpc_cfg* cfg, int n_rows, int max_rows = 15) {
  // developer switches (read at every call): always the workspace kernel / never the Zc = 384 specialisation (nrx_ldpc_dec4.hip then)
  const bool off = getenv("NRX_LDPC_NOCHIP64") != nullptr || getenv("NRX_LDPC_NOCHIP384") != nullptr;
  return !off && cfg->bg == 1 && cfg->Zc == 384 && cfg->iLS == 1 && n_rows <= max_rows;
}

}  // namespace nrx_dec3

// Called by nrx_ldpc_decode_rows_f64 (nrx_ldpc_dec.hip) for hard decisions of the K information bits.
// Returns 1 when no on-chip instantiation covers (bg, Zc, n_rows): the caller then runs the workspace kernel.
int32_t nrx_ldpc_decode_chip64_launch(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                                      int32_t n_rows, uint8_t* hard, hipStream_t st, void* ws, size_t ws_bytes,
                                      const int32_t* sel, const int32_t* n_sel) {
  using namespace nrx_dec3;
  if (chip64_covers(cfg, n_rows, 46) && n_rows > 15 && getenv("NRX_LDPC_NOHYBRID") == nullptr) {
    // more rows than fit on chip: the hybrid -- the first 31 or all 46 rows of the graph run (the ones beyond n_rows as exact
    // no-ops on zeroed extension LLRs), the state of the sparse rows streams through the caller's workspace
    constexpr int ZI384 = zindex_c(384);
    const int ra = n_rows <= 31 ? 31 : 46, rc_rows = n_rows <= 31 ? HYB_RC31 : HYB_RC46;
    const size_t per_slot = sizeof(double) * (size_t)(ra - rc_rows) * 3 * 384;
    const int n_wg = (n_cb + 1) / 2;
    int grid = n_wg < 256 ? n_wg : 256;
    if (ws == nullptr || ws_bytes < 2 * per_slot) return 1;          // no workspace: the workspace kernel reports it
    if ((size_t)grid > ws_bytes / (2 * per_slot)) grid = (int)(ws_bytes / (2 * per_slot));
    const uint64_t* wt = nullptr;
    const int32_t rc = wrap_table(ra, &wt);
    if (rc) return rc;
    FuseArgs fa{};
    fa.g.rows_live = n_rows;
    fa.ws = (double*)ws;
    fa.sel = sel;
    fa.n_sel = n_sel;
    // (fewer rows than the instantiation has: the copy that leaves the layers beyond n_rows out, MODE bit 3)
    if (ra == 31 && n_rows == 31)
      hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 31, false, 2, 0, HYB_RC31>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard,
                         (mtab_t)wt, fa);
    else if (ra == 31)
      hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 31, false, 2, 8, HYB_RC31>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard,
                         (mtab_t)wt, fa);
    else if (n_rows == 46)
      hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 46, false, 2, 0, HYB_RC46>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard,
                         (mtab_t)wt, fa);
    else
      hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 46, false, 2, 8, HYB_RC46>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard,
                         (mtab_t)wt, fa);
    NRX_CHECK_LAUNCH("nrx_ldpc_decode_f64(hybrid)");
    return NRX_OK;
  }
  if (!chip64_covers(cfg, n_rows)) return 1;
  const uint64_t* wt = nullptr;
  const int32_t rc = wrap_table(n_rows, &wt);
  if (rc) return rc;
  constexpr int ZI384 = zindex_c(384);
  FuseArgs fa{};
  fa.g.rows_live = n_rows;
  fa.sel = sel;
  fa.n_sel = n_sel;
  const int n_wg = (n_cb + 1) / 2;
  const int grid = n_wg < 1024 ? n_wg : 1024;
  if (n_rows <= 13)
    hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 13, false>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard, (mtab_t)wt, fa);
  else
    hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, 15, false>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, hard, (mtab_t)wt, fa);
  NRX_CHECK_LAUNCH("nrx_ldpc_decode_f64(on-chip)");
  return NRX_OK;
}

// ldpc.py:1330-1418 recoverRate (first transmission) + :1495-1581 decode + :1584-1619 checkCrcAndMerge in ONE launch.
// NRX_E_UNSUPPORTED when the configuration has no fused instantiation: the caller runs the three separate entries.
static int32_t recover_decode_merge_impl(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                         int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                         uint8_t* cb_ok, const int32_t* sel, const int32_t* n_sel, void* stream,
                                         int mode = 0, double* state = nullptr, size_t* state_bytes_per_cb = nullptr,
                                         double* lam = nullptr, int* rows_run = nullptr, float* cert_w = nullptr, size_t cert_w_bytes = 0,
                                         uint8_t* exit_iter = nullptr, const nrx_certcore::Params* cp = nullptr,
                                         int32_t* queue = nullptr, const int32_t* stage_iters = nullptr, int n_stages = 0) {
  using namespace nrx_dec3;
  NRX_REQUIRE(llr && cfg && tb_out && cb_ok, NRX_E_ARG, "nrx_ldpc_recover_decode_merge: NULL buffer");
  NRX_REQUIRE(nl >= 1 && qm >= 1 && llr_len > 0 && n_tb >= 0 && n_iter >= 0, NRX_E_ARG, "nrx_ldpc_recover_decode_merge: bad argument");
  const int f = nl * qm, gb = (llr_len + f - 1) / f;
  NRX_REQUIRE(cfg->C >= 1 && gb / cfg->C > 0, NRX_E_SHAPE, "nrx_ldpc_recover_decode_merge: G=%d too small for %d code blocks", llr_len, cfg->C);
  FuseArgs fa;
  fa.tb_out = tb_out;
  fa.cb_ok = cb_ok;
  fa.sel = sel;
  fa.n_sel = n_sel;
  fa.state = state;
  fa.ws = nullptr;
  fa.lam = lam;
  fa.cert_w = nullptr;
  fa.exit_iter = exit_iter;
  fa.cp = nrx_certcore::Params{};
  fa.queue = queue;
  fa.n_stages = n_stages;
  for (int i = 0; i < 4; ++i) fa.stage_iters[i] = (stage_iters && i < n_stages) ? stage_iters[i] : 0;
  FuseGeom& fg = fa.g;
  fg.C = cfg->C; fg.f = f; fg.qm = qm; fg.F = cfg->F; fg.llr_len = llr_len; fg.cb_len = cfg->cb_len;
  fg.e_small = (gb / cfg->C) * f;
  fg.n_small = cfg->C - gb % cfg->C;
  fg.sys_len = cfg->K - 2 * cfg->Zc - cfg->F;
  fg.payload = cfg->cb_len - 24;
  fg.rows_live = 46;
  const int e_max = fg.e_small + (fg.n_small < cfg->C ? f : 0);
  // the rows that can matter for e_max received bits (as ops.ldpc_active_rows): never fewer than the caller asks for
  const int last = e_max - 1 + (e_max > fg.sys_len ? cfg->F : 0);
  int need = last / cfg->Zc + 2 - 26 + 1 + 4;
  if (need < 4) need = 4;
  if (n_rows < need) n_rows = need;
  if (cfg->C < 2 || cfg->cb_len <= 24 || e_max > cfg->N - cfg->F || fg.n_small * fg.e_small + (cfg->C - fg.n_small) * (fg.e_small + f) != llr_len || fg.e_small % qm || (fg.e_small + f) % qm || !chip64_covers(cfg, n_rows)) {
    ::nrx::set_error("nrx_ldpc_recover_decode_merge: no fused instantiation for bg %d Zc %d C %d rows %d", cfg->bg, cfg->Zc, cfg->C, n_rows);
    return NRX_E_UNSUPPORTED;
  }
  if (rows_run) *rows_run = n_rows <= 13 ? 13 : 15;          // (the rows of the instantiation that runs)
  if (state_bytes_per_cb) {          // (query: the size of a parked state for this configuration's instantiation)
    *state_bytes_per_cb = sizeof(double) * 384 * (size_t)(n_rows <= 13 ? StateLay<1, 13>::NF : StateLay<1, 15>::NF);
    return NRX_OK;
  }
  if (n_tb == 0) return NRX_OK;
  const uint64_t* wt = nullptr;
  const int32_t rc = wrap_table(n_rows, &wt);
  if (rc) return rc;
  const int n_cb = n_tb * cfg->C;
  const int n_wg = (n_cb + 1) / 2;
  int grid = n_wg < 1024 ? n_wg : 1024;
  if (cert_w) {            // the in-kernel certificate: its slack sums need 26 * 384 floats per code-block slot of the launch
    NRX_REQUIRE(cp && exit_iter && lam && (mode & 4), NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: missing certificate argument");
    const size_t per_wg = 2 * sizeof(float) * (26 + 2 * 15) * 384;      // slack sums of 26 columns + two slacks per row and lane
    NRX_REQUIRE(cert_w_bytes >= per_wg, NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: scratch of %zu bytes (one workgroup needs %zu)", cert_w_bytes, per_wg);
    if ((size_t)grid > cert_w_bytes / per_wg) grid = (int)(cert_w_bytes / per_wg);
    fa.cert_w = cert_w;
    fa.cp = *cp;
  }
  constexpr int ZI384 = zindex_c(384);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 21) {        // the persistent certified schedule: one workgroup per CU, blocks drawn from the queue
    NRX_REQUIRE(queue && cert_w && n_stages >= 1 && n_stages <= 4, NRX_E_ARG, "nrx_ldpc_certified_persistent: missing queue / scratch / stage plan");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (grid > cus) grid = cus;
    NRX_REQUIRE(hipMemsetAsync(queue, 0, 2 * sizeof(int32_t), st) == hipSuccess, NRX_E_HIP, "nrx_ldpc_certified_persistent: hipMemsetAsync failed");
    NRX_REQUIRE(hipMemsetAsync(exit_iter, 0, (size_t)n_cb, st) == hipSuccess, NRX_E_HIP, "nrx_ldpc_certified_persistent: hipMemsetAsync failed");
  }
#define NRX_FUSED_LAUNCH(RA_, MODE_) \
  hipLaunchKernelGGL((ldpc_dec_chip64_kernel<1, ZI384, RA_, true, 2, MODE_>), dim3(grid), dim3(768), 0, st, llr, n_cb, n_iter, nullptr, (mtab_t)wt, fa)
  // a whole decode that needs two or more rows fewer than its instantiation has: the copy that leaves them out (one row fewer: the waves
  // see the all-zero last layer themselves, NRX_DEC3_SKIPZ)
  fg.rows_live = n_rows;
  if (mode == 0 && n_rows <= 11) NRX_FUSED_LAUNCH(13, 8);
  else if (n_rows <= 13) {
    if (mode == 0) NRX_FUSED_LAUNCH(13, 0); else if (mode == 1) NRX_FUSED_LAUNCH(13, 1); else if (mode == 2) NRX_FUSED_LAUNCH(13, 2); else if (mode == 3) NRX_FUSED_LAUNCH(13, 3);
    else if (mode == 5) NRX_FUSED_LAUNCH(13, 5); else if (mode == 21) NRX_FUSED_LAUNCH(13, 21); else NRX_FUSED_LAUNCH(13, 7);
  } else {
    if (mode == 0) NRX_FUSED_LAUNCH(15, 0); else if (mode == 1) NRX_FUSED_LAUNCH(15, 1); else if (mode == 2) NRX_FUSED_LAUNCH(15, 2); else if (mode == 3) NRX_FUSED_LAUNCH(15, 3);
    else if (mode == 5) NRX_FUSED_LAUNCH(15, 5); else if (mode == 21) NRX_FUSED_LAUNCH(15, 21); else NRX_FUSED_LAUNCH(15, 7);
  }
#undef NRX_FUSED_LAUNCH
  NRX_CHECK_LAUNCH("nrx_ldpc_recover_decode_merge_f64");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_recover_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                                     int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                                     uint8_t* cb_ok, void* stream) {
  return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, nullptr, nullptr, stream);
}

// The same for a selection of the code blocks: sel[0 .. *n_sel) are indices into the n_tb * C blocks, list and count both on
// the device (the launch covers the worst case; no host read).  Only the selected blocks' payload bits in tb_out and their
// cb_ok entries are written.  With nrx_select_failed this is the second pass of the two-pass schedule: every block decoded with
// few iterations first, the ones whose CRC fails decoded again from scratch with all of them.
extern "C" int32_t nrx_ldpc_recover_decode_merge_sel_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                                         int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                                         uint8_t* cb_ok, const int32_t* sel, const int32_t* n_sel, void* stream) {
  NRX_REQUIRE(sel && n_sel, NRX_E_ARG, "nrx_ldpc_recover_decode_merge_sel: NULL selection");
  return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, sel, n_sel, stream);
}

// The continuation form of the two-pass schedule.  _park: nrx_ldpc_recover_decode_merge_f64, and every block whose CRC24B
// fails leaves its decoder state in `state` (nrx_ldpc_fused_state_bytes per code block, indexed by code block).  _resume: the
// blocks of the selection continue from their parked state for n_iter MORE iterations (llr is not read): park(n1) followed by
// resume(n2) computes exactly what one run of n1 + n2 iterations computes for those blocks.
extern "C" int64_t nrx_ldpc_fused_state_bytes(const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm, int32_t llr_len, int32_t n_rows) {
  if (!cfg) return -1;
  size_t b = 0;
  uint8_t dummy = 0;
  const double d0 = 0;
  const int32_t rc = recover_decode_merge_impl(&d0, 1, llr_len, cfg, nl, qm, 1, n_rows, &dummy, &dummy, nullptr, nullptr, nullptr, 0, nullptr, &b);
  return rc == NRX_OK ? (int64_t)b : (int64_t)rc;
}
extern "C" int32_t nrx_ldpc_recover_decode_merge_park_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                                          int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                                          uint8_t* cb_ok, void* state, void* stream) {
  NRX_REQUIRE(state, NRX_E_ARG, "nrx_ldpc_recover_decode_merge_park: NULL state");
  return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, nullptr, nullptr, stream, 1, (double*)state);
}
extern "C" int32_t nrx_ldpc_resume_decode_merge_sel_f64(int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm,
                                                        int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                                        const int32_t* sel, const int32_t* n_sel, void* state, int32_t park_again,
                                                        void* stream) {
  NRX_REQUIRE(state && sel && n_sel, NRX_E_ARG, "nrx_ldpc_resume_decode_merge_sel: NULL state / selection");
  return recover_decode_merge_impl((const double*)state, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, sel, n_sel, stream,
                                   park_again ? 3 : 2, (double*)state);
}

// (for nrx_ldpc_cert.hip) the rows of the fused instantiation that serves this configuration; NRX_E_UNSUPPORTED when there is none
int32_t nrx_ldpc_fused_rows_run(const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm, int32_t llr_len, int32_t n_rows, int* rows_run) {
  size_t per = 0;
  uint8_t dummy = 0;
  const double d0 = 0;
  return recover_decode_merge_impl(&d0, 1, llr_len, cfg, nl, qm, 1, n_rows, &dummy, &dummy, nullptr, nullptr, nullptr, 0, nullptr, &per, nullptr, rows_run);
}

// A stage with the certificate evaluated in the kernel's tail (nrx_ldpc_certcore.h): nrx_ldpc_stage_decode_merge_f64, and a block
// whose CRC passes and whose state holds the stability certificate gets exit_iter[cb] = min(iter_now, 255) and does not park.
extern "C" int32_t nrx_ldpc_stage_certify_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                                           int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                                           const int32_t* sel, const int32_t* n_sel, void* state, double* lam,
                                                           uint8_t* exit_iter, void* scratch, size_t scratch_bytes, int32_t iter_now,
                                                           int32_t n_iter_total, int32_t max_sweeps, int32_t flags, void* stream) {
  NRX_REQUIRE(state && lam && exit_iter && scratch, NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: NULL state / maxima / exit_iter / scratch");
  NRX_REQUIRE((sel == nullptr) == (n_sel == nullptr), NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: selection list without its count");
  NRX_REQUIRE(iter_now >= 1 && n_iter_total >= iter_now && max_sweeps >= 1, NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: bad iteration counts");
  int rows_run = 0;
  const int32_t rr = nrx_ldpc_fused_rows_run(cfg, nl, qm, llr_len, n_rows, &rows_run);
  if (rr) return rr;
  double b3[3];
  const int32_t rb = nrx_ldpc_cert_bounds(cfg, rows_run, b3);
  if (rb) return rb;
  nrx_certcore::Params cp;
  cp.gamma = b3[0]; cp.gamma1 = b3[1]; cp.dmax = (int32_t)b3[2]; cp.n_iter_total = n_iter_total; cp.max_sweeps = max_sweeps < 16 ? max_sweeps : 16; cp.flags = flags;
  cp.iter_now = iter_now;
  if (sel == nullptr) {
    NRX_REQUIRE(llr, NRX_E_ARG, "nrx_ldpc_stage_certify_decode_merge: NULL llr");
    return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, nullptr, nullptr, stream, 5, (double*)state,
                                     nullptr, lam, nullptr, (float*)scratch, scratch_bytes, exit_iter, &cp);
  }
  return recover_decode_merge_impl((const double*)state, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, sel, n_sel, stream, 7,
                                   (double*)state, nullptr, lam, nullptr, (float*)scratch, scratch_bytes, exit_iter, &cp);
}

// The certified schedule as ONE launch (round 6): every code-block slot of the grid (one workgroup of two slots per CU) draws blocks from a
// device queue and takes each through its stages -- stages[0] iterations from the LLRs, CRC + certificate, stages[1] more, ... the last
// stage without a certificate -- stopping at the first stage whose CRC passes and whose frozen state holds the stability certificate
// (exit_iter[cb] = the iterations it had then; 0 = it ran all of them).  The two slots of a workgroup never wait for each other, nothing is
// parked between stages and nothing reloaded: same bits and verdicts as the staged launches (nrx_ldpc_stage_certify_decode_merge_f64 ->
// nrx_select_failed -> ...) and as the fixed schedule of sum(stages) iterations.  state: one record of nrx_ldpc_fused_state_bytes(...) per
// slot of the grid (2 x CUs; where a slot's certificate reads the check-node state back), scratch / scratch_bytes: the slack sums (as the stage
// entry), lam[2 n_cb], queue: int32[2] on the device (zeroed here; queue[1] != 0 afterwards = a slot barrier gave up: results invalid).
extern "C" int32_t nrx_ldpc_certified_persistent_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                                     int32_t qm, const int32_t* stages, int32_t n_stages, int32_t n_rows, uint8_t* tb_out,
                                                     uint8_t* cb_ok, void* state, double* lam, uint8_t* exit_iter, void* scratch,
                                                     size_t scratch_bytes, int32_t* queue, int32_t max_sweeps, int32_t flags, void* stream) {
  NRX_REQUIRE(llr && state && lam && exit_iter && scratch && queue && stages, NRX_E_ARG, "nrx_ldpc_certified_persistent: NULL buffer");
  NRX_REQUIRE(n_stages >= 1 && n_stages <= 4 && max_sweeps >= 1, NRX_E_ARG, "nrx_ldpc_certified_persistent: 1 ... 4 stages");
  int total = 0;
  for (int i = 0; i < n_stages; ++i) {
    NRX_REQUIRE(stages[i] >= 1, NRX_E_ARG, "nrx_ldpc_certified_persistent: a stage needs one iteration or more");
    total += stages[i];
  }
  int rows_run = 0;
  const int32_t rr = nrx_ldpc_fused_rows_run(cfg, nl, qm, llr_len, n_rows, &rows_run);
  if (rr) return rr;
  double b3[3];
  const int32_t rb = nrx_ldpc_cert_bounds(cfg, rows_run, b3);
  if (rb) return rb;
  nrx_certcore::Params cp;
  cp.gamma = b3[0]; cp.gamma1 = b3[1]; cp.dmax = (int32_t)b3[2]; cp.n_iter_total = total; cp.max_sweeps = max_sweeps < 16 ? max_sweeps : 16; cp.flags = flags;
  cp.iter_now = 0;
  return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, total, n_rows, tb_out, cb_ok, nullptr, nullptr, stream, 21, (double*)state,
                                   nullptr, lam, nullptr, (float*)scratch, scratch_bytes, exit_iter, &cp, queue, stages, n_stages);
}

// One STAGE of the certified schedule on the fused entry: n_iter iterations of every block (sel == NULL: from the LLRs, which
// also leaves the two LLR maxima of every block in lam[2 n_cb]) or of the selected blocks (continued from their parked state;
// llr is not read), and EVERY block that ran parks its complete decoder state, whatever its CRC says.
extern "C" int32_t nrx_ldpc_stage_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                                   int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                                   const int32_t* sel, const int32_t* n_sel, void* state, double* lam, void* stream) {
  NRX_REQUIRE(state && lam, NRX_E_ARG, "nrx_ldpc_stage_decode_merge: NULL state / maxima");
  NRX_REQUIRE((sel == nullptr) == (n_sel == nullptr), NRX_E_ARG, "nrx_ldpc_stage_decode_merge: selection list without its count");
  if (sel == nullptr) {
    NRX_REQUIRE(llr, NRX_E_ARG, "nrx_ldpc_stage_decode_merge: NULL llr");
    return recover_decode_merge_impl(llr, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, nullptr, nullptr, stream, 5, (double*)state,
                                     nullptr, lam);
  }
  return recover_decode_merge_impl((const double*)state, n_tb, llr_len, cfg, nl, qm, n_iter, n_rows, tb_out, cb_ok, sel, n_sel, stream, 7,
                                   (double*)state, nullptr, lam);
}

namespace {
// indices of the zero entries of flags[0 .. n), ascending, and their number: one workgroup, a ballot scan per 1024 entries
__global__ void __launch_bounds__(1024)
select_failed_kernel(const uint8_t* __restrict__ flags, int n, int32_t* __restrict__ sel, int32_t* __restrict__ n_sel) {
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool f = i < n && flags[i] == 0;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < w; ++k) off += wave_cnt[k];
    if (f) sel[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int k = 0; k < 16; ++k) t += wave_cnt[k];
      base_s += t;
    }
    __syncthreads();
  }
  if (tid == 0) *n_sel = base_s;
}
}  // namespace

// sel (>= n entries) = the indices i with flags[i] == 0 in ascending order, *n_sel = how many: all on the device.
extern "C" int32_t nrx_select_failed(const uint8_t* flags, int32_t n, int32_t* sel, int32_t* n_sel, void* stream) {
  NRX_REQUIRE(flags && sel && n_sel && n >= 0, NRX_E_ARG, "nrx_select_failed: bad argument");
  hipLaunchKernelGGL(select_failed_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flags, n, sel, n_sel);
  NRX_CHECK_LAUNCH("nrx_select_failed");
  return NRX_OK;
}

