// Polar codec for the control-channel path (gfx950): encode, rate match, rate recover, CRC-aided SCL decode.
//
// Reference: polar.py:527-564 (encode), :567-603 (rateMatch), :882-928 (recoverRate), :606-720 (SclDecoder),
// :931-982 (decode).  The code construction (frozen / message / parity-check sets, interleaver patterns) is
// host-side integer bookkeeping (neoradium_amd/polar.py) and arrives here as index tables.
//
// The SCL decoder is the only kernel with real work in it.  It is latency-bound, data-dependent byte/double work:
// one wavefront decodes one codeword, thousands of blind-decode candidates run side by side.  Layout per wave (LDS):
//   * LLR rows per tree stage s = 1..n-1: 8 rows x 2^s doubles.  A stage array is rewritten in full whenever its
//     tree node is entered, for the candidates alive at that moment, so forks never copy LLRs: each candidate
//     only keeps, per stage, the 4-bit row number it reads from (one packed 64-bit register, permuted by a
//     cross-lane shuffle when the list is re-ranked).
//   * partial sums of completed left children, bit-packed, stage s = 0..n-1, same row-number scheme.
//   * one byte per (information leaf, list slot): parent slot + decided bit -- the decoded words are recovered by
//     a trace-back at the end instead of carrying N-bit histories through every fork.
// All arithmetic is float64 and follows the reference operation for operation (f = sign*sign*min with
// sign(0)=0, g = b + (1-2x)a, cost updates, stable ranking of the 2L forked costs), so the surviving list, its
// order and the path costs are bit-identical to the reference's.
#include "nrx_common.h"
#include "nrx_crc.h"
#include <utility>

namespace {

using nrx::crc_len;
using nrx::crc_poly;

constexpr int LMAX = 8;  // list slots (rows) per stage

// ---------------------------------------------------------------------------------------------------- encode
// polar.py:527-564.  One workgroup per code block; the N-bit word lives in LDS.  u G_N is the XOR butterfly
// (G_N = F^{(x)n}: in every block of 2h bits the first half absorbs the second), not the dense N x N product.
__global__ void polar_encode_kernel(const uint8_t* __restrict__ cbs, int K, int N, const int32_t* __restrict__ in_il,
                                    const int32_t* __restrict__ msg_pos, const int32_t* __restrict__ pc_pos, int n_pc,
                                    uint8_t* __restrict__ out) {
  __shared__ uint8_t x[1024];
  const int cw = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < N; i += nt) x[i] = 0;
  __syncthreads();
  const uint8_t* cb = cbs + (int64_t)cw * K;
  for (int k = tid; k < K; k += nt) x[msg_pos[k]] = cb[in_il ? in_il[k] : k] & 1;
  __syncthreads();
  if (n_pc > 0 && tid == 0) {  // TS 38.212 5.3.1.2: 5-stage cyclic shift register (polar.py:554-560)
    uint32_t y[5] = {0, 0, 0, 0, 0};
    int head = 0;
    for (int i = 0; i < N; ++i) {
      head = (head + 1) % 5;  // np.roll(y, -1): y[0] is the next register
      bool is_pc = false;
      for (int p = 0; p < n_pc; ++p) is_pc |= (pc_pos[p] == i);
      if (is_pc)
        x[i] = (uint8_t)y[head];
      else
        y[head] ^= x[i];
    }
  }
  __syncthreads();
  for (int h = 1; h < N; h <<= 1) {
    for (int p = tid; p < N / 2; p += nt) {
      const int lo = ((p / h) * 2 * h) + (p % h);
      x[lo] ^= x[lo + h];
    }
    __syncthreads();
  }
  for (int i = tid; i < N; i += nt) out[(int64_t)cw * N + i] = x[i];
}

// polar.py:567-603: sub-block interleave + bit selection + coded-bit interleave, composed on the host to one gather.
__global__ void polar_gather_kernel(const uint8_t* __restrict__ in, int64_t n_rows, int N, int E,
                                    const int32_t* __restrict__ idx, uint8_t* __restrict__ out) {
  const int64_t total = n_rows * E;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = g / E;
    const int e = (int)(g - r * E);
    out[g] = in[r * N + idx[e]];
  }
}

// polar.py:882-928.  mode 0: repetition (sum of the repeated LLRs in transmission order, TS 38.212 5.4.1.2),
// 1: puncturing (zeros in front), 2: shortening (LARGE_LLR behind).  deil = inverse coded-bit interleaver or null.
__global__ void polar_rate_recover_kernel(const double* __restrict__ llr, int64_t n_rows, int N, int E, int mode,
                                          const int32_t* __restrict__ deil, const int32_t* __restrict__ inv_sb,
                                          double* __restrict__ out) {
  const int64_t total = n_rows * N;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = g / N;
    const int nn = (int)(g - r * N);
    const int p = inv_sb[nn];
    const double* row = llr + r * E;
    double v;
    if (mode == 0) {
      v = 0.0;
      for (int i = p; i < E; i += N) v += row[deil ? deil[i] : i];
    } else if (mode == 1) {
      const int i = p - (N - E);
      v = (i >= 0) ? row[deil ? deil[i] : i] : 0.0;
    } else {
      v = (p < E) ? row[deil ? deil[p] : p] : 1e20;
    }
    out[g] = v;
  }
}

// ------------------------------------------------------------------------------------------------ SCL decode
struct SclLayout {
  int llr_off;   // doubles: 8 * (2^n - 2)
  int xl_off;    // bytes
  int hist_off;  // bytes
  int scr_off;   // bytes
  int kind_off;  // bytes: per-leaf kind byte (copy of info_mask)
  int total;     // bytes
};
__host__ __device__ inline int xl_words(int s) { return s <= 5 ? 1 : (1 << (s - 5)); }
__host__ __device__ inline int xl_stage_off(int s) {  // in words, LMAX rows per stage
  // stages 0..5 take one word per row, stage s>5 takes 2^(s-5)
  return s <= 6 ? s * LMAX : (6 + (1 << (s - 5)) - 2) * LMAX;
}
__host__ __device__ inline SclLayout scl_layout(int n, int k_info) {
  SclLayout l;
  const int N = 1 << n;
  int llr_bytes = LMAX * (N - 2) * 8;
  const int ub_bytes = LMAX * k_info;  // trace-back bits reuse the LLR region
  if (ub_bytes > llr_bytes) llr_bytes = (ub_bytes + 7) & ~7;
  l.llr_off = 0;
  l.xl_off = llr_bytes;
  const int xl_bytes = xl_stage_off(n) * 4;
  l.hist_off = l.xl_off + xl_bytes;
  const int hist_bytes = ((LMAX * k_info) + 7) & ~7;
  l.scr_off = l.hist_off + hist_bytes;
  l.kind_off = l.scr_off + 16 * 8 + 16 * 4;
  l.total = l.kind_off + N;
  return l;
}

__device__ __forceinline__ double clip20(double v) { return fmin(fmax(v, -20.0), 20.0); }  // polar.py:958

// value of lane r (compile-time r) as a wave-uniform scalar: two v_readlane instead of two LDS-crossbar permutes
template <int R>
__device__ __forceinline__ double lane_value(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, R);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), R);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int N, class F>
__device__ __forceinline__ void unrolled(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) {
    (f(std::integral_constant<int, I>{}), ...);
  }(std::make_integer_sequence<int, N>{});
}

// One wavefront per codeword.  info_mask[i] = 1 for every non-frozen leaf (message and parity-check bits).
__global__ __launch_bounds__(64) void polar_scl_kernel(
    const double* __restrict__ llr_in, int n, int L, const uint8_t* __restrict__ info_mask, int k_info,
    const int32_t* __restrict__ msg_src, int K, int crc_id, const uint32_t* __restrict__ crc_expect,
    uint8_t* __restrict__ msg_out, uint8_t* __restrict__ crc_ok, uint8_t* __restrict__ cand_out,
    double* __restrict__ cost_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const SclLayout lay = scl_layout(n, k_info);
  double* const lds_llr = reinterpret_cast<double*>(smem + lay.llr_off);
  uint32_t* const xl = reinterpret_cast<uint32_t*>(smem + lay.xl_off);
  uint8_t* const hist = smem + lay.hist_off;
  double* const sc_cost = reinterpret_cast<double*>(smem + lay.scr_off);
  int* const sel = reinterpret_cast<int*>(smem + lay.scr_off + 16 * 8);

  const int N = 1 << n;
  const int lane = threadIdx.x;
  const int64_t cw = blockIdx.x;
  const double* chan = llr_in + cw * N;
  // leaf kinds: bit 0 = non-frozen leaf; bits 1..4 = s0 > 0 when leaves [i, i + 2^s0) form an aligned all-frozen
  // block (rate-0 node) starting here
  uint8_t* const kind = smem + lay.kind_off;
  for (int i = lane; i < N; i += 64) kind[i] = info_mask[i];
  __syncthreads();

  // per-candidate state, owned by lane c (< LMAX)
  double cost = 0.0;
  uint64_t lrow = 0, xrow = 0;  // 4-bit row numbers per stage
  int count = 1;                // wave-uniform
  int kinfo = 0;                // wave-uniform

  auto stage_ptr = [&](int s) { return lds_llr + LMAX * ((1 << s) - 2); };  // stage s >= 1

  for (int i = 0; i < N;) {
    const int t = (i == 0) ? n : __builtin_ctz(i);
    const int kd = kind[i];
    int s0 = (kd & 1) ? 0 : (kd >> 1) & 15;   // rate-0 node of 2^s0 frozen leaves starts here (0: plain leaf)
    if (s0 > t) s0 = t;                        // (cannot happen for a well-formed table; keeps the descent sane)
    if (s0 > n - 1) s0 = n - 1;
    double leaf = 0.0;  // leaf LLR of candidate `lane`
    // ---- g step at stage t (right child entered): polar.py:659-663
    if (i != 0) {
      const int packed = (int)((lrow >> (4 * (t + 1))) & 15u) | ((int)((xrow >> (4 * t)) & 15u) << 4);
      const int half = 1 << t;
      const int items = count << t;
      const double* src = (t + 1 < n) ? stage_ptr(t + 1) : nullptr;
      double* dst = (t > 0) ? stage_ptr(t) : nullptr;   // (t >= s0 >= 1 on the rate-0 path)
      const uint32_t* xs = xl + xl_stage_off(t);
      const int xw = xl_words(t);
      for (int it0 = 0; it0 < items; it0 += 64) {
        const int it = it0 + lane;
        const int c = (it >> t) & (LMAX - 1);
        const int e = it & (half - 1);
        const int pk = __shfl(packed, c, 64);
        if (it < items) {
          double a, b;
          if (src) {
            const double* r = src + (pk & 15) * (2 * half);
            a = r[e];
            b = r[e + half];
          } else {
            a = clip20(chan[e]);
            b = clip20(chan[e + half]);
          }
          const uint32_t xbit = (xs[(pk >> 4) * xw + (e >> 5)] >> (e & 31)) & 1u;
          const double v = b + (xbit ? -a : a);
          if (dst)
            dst[c * half + e] = v;
          else
            leaf = v;
        }
      }
      __syncthreads();
    }
    // ---- f steps down to the leaf: polar.py:697-705
    for (int s = ((i == 0) ? n : t) - 1; s >= s0; --s) {
      const int half = 1 << s;
      const int items = count << s;
      const double* src = (s + 1 < n) ? stage_ptr(s + 1) : nullptr;
      double* dst = (s > 0) ? stage_ptr(s) : nullptr;
      for (int it0 = 0; it0 < items; it0 += 64) {
        const int it = it0 + lane;
        if (it < items) {
          const int c = it >> s;
          const int e = it & (half - 1);
          double a, b;
          if (src) {
            const double* r = src + c * (2 * half);
            a = r[e];
            b = r[e + half];
          } else {
            a = clip20(chan[e]);
            b = clip20(chan[e + half]);
          }
          const double m = fmin(fabs(a), fabs(b));
          double v = ((a < 0.0) != (b < 0.0)) ? -m : m;
          if (a == 0.0 || b == 0.0) v = 0.0;  // np.sign(0) = 0
          if (dst)
            dst[c * half + e] = v;
          else
            leaf = v;
        }
      }
      __syncthreads();
    }
    // every stage <= min(t, n-1) now holds one row per live candidate, in slot order
    {
      const int top = (t < n) ? t : n - 1;
      const uint64_t mask = (top >= 15) ? ~0ull : ((1ull << (4 * (top + 1))) - 1ull);
      lrow = (lrow & ~mask) | ((0x1111111111111111ull * (uint64_t)(lane & 15)) & mask);
    }

    // ---- rate-0 node: every leaf below is frozen, so no decision depends on another one's.  The whole subtree is
    // expanded level by level, in place in the stage-s0 rows (left child <- f(a,b), right child <- g(a,b,u=0) = b + a:
    // the values the leaf-by-leaf recursion computes, polar.py:697-713), then the 2^s0 frozen-leaf penalties are
    // added to each path cost in leaf order (polar.py:623-628: same additions, same order).
    if (s0 > 0) {
      double* A = stage_ptr(s0);
      const int rowlen = 1 << s0;
      for (int lev = s0; lev >= 1; --lev) {
        const int h = 1 << (lev - 1);
        const int items = count << (s0 - 1);
        for (int it0 = 0; it0 < items; it0 += 64) {
          const int it = it0 + lane;
          if (it < items) {
            const int c = it >> (s0 - 1);
            const int pidx = it & ((1 << (s0 - 1)) - 1);          // pair index inside the row
            const int idx = ((pidx >> (lev - 1)) << lev) | (pidx & (h - 1));
            double* r = A + c * rowlen + idx;
            const double a = r[0], b = r[h];
            const double m = fmin(fabs(a), fabs(b));
            double v = ((a < 0.0) != (b < 0.0)) ? -m : m;
            if (a == 0.0 || b == 0.0) v = 0.0;
            r[0] = v;
            r[h] = b + a;
          }
        }
        __syncthreads();
      }
      if (lane < count) {
        const double* r = A + lane * rowlen;
        for (int k = 0; k < rowlen; ++k) cost = cost - fmin(0.0, r[k]);
      }
    }

    // ---- leaf
    int ubit = 0;
    if (s0 > 0) {
      // (handled above)
    } else if (!(kd & 1)) {
      if (lane < count) cost = cost - fmin(0.0, leaf);  // polar.py:623-628
    } else {
      // polar.py:631-656: 2*count forked costs [all 0-branches, all 1-branches], keep the L cheapest, stable.
      const int q = lane & 15, qc = q & 7, qb = q >> 3;
      const double cq = __shfl(cost, qc, 64);
      const double lq = __shfl(leaf, qc, 64);
      const double nc = qb ? (cq + fmax(0.0, lq)) : (cq - fmin(0.0, lq));
      const bool valid = qc < count;
      int rank = 0;
      unrolled<16>([&](auto rc) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const double cr = lane_value<r>(nc);
        const bool vr = (r & 7) < count;
        rank += (vr && (cr < nc || (cr == nc && r < q))) ? 1 : 0;
      });
      if (lane < 16) {
        sc_cost[lane] = nc;
        if (valid && rank < L) sel[rank] = q;
      }
      __syncthreads();
      const int two = 2 * count;
      count = two < L ? two : L;
      const int myq = (lane < count) ? sel[lane] : 0;
      const int par = myq & 7;
      ubit = myq >> 3;
      const uint32_t lr_lo = __shfl((uint32_t)lrow, par, 64), lr_hi = __shfl((uint32_t)(lrow >> 32), par, 64);
      const uint32_t xr_lo = __shfl((uint32_t)xrow, par, 64), xr_hi = __shfl((uint32_t)(xrow >> 32), par, 64);
      lrow = ((uint64_t)lr_hi << 32) | lr_lo;
      xrow = ((uint64_t)xr_hi << 32) | xr_lo;
      if (lane < count) {
        cost = sc_cost[myq];
        hist[kinfo * LMAX + lane] = (uint8_t)myq;
      }
      ++kinfo;
      __syncthreads();
    }

    // ---- partial sums: the finished leaf closes `S` right children in a row (polar.py:666-668); the
    // combined word becomes the left-child result of stage S.
    const int il = i + (1 << s0) - 1;          // last leaf of this step
    if (il != N - 1) {
      const int S = __builtin_ctz(~(unsigned)il);
      const int W = xl_words(S);
      const int items = count * W;
      for (int it0 = 0; it0 < items; it0 += 64) {
        const int it = it0 + lane;
        const int c = (it / W) & (LMAX - 1);
        const int w = it % W;
        const uint32_t ub = (uint32_t)__shfl(ubit, c, 64);
        const uint32_t x_lo = __shfl((uint32_t)xrow, c, 64), x_hi = __shfl((uint32_t)(xrow >> 32), c, 64);
        const uint64_t xr = ((uint64_t)x_hi << 32) | x_lo;
        if (it < items) {
          uint32_t cur = ub;                   // the step's own partial-sum vector (all zero for a rate-0 node)
          const int low = S < 5 ? S : 5;
          for (int j = s0; j < low; ++j) {
            const uint32_t xv = xl[xl_stage_off(j) + (int)((xr >> (4 * j)) & 15u)];
            const int sh = 1 << j;
            cur = ((xv ^ cur) & ((1u << sh) - 1u)) | (cur << sh);
          }
          for (int j = (s0 > 5 ? s0 : 5); j < S; ++j) {
            if (!((w >> (j - 5)) & 1)) {
              const int wj = xl_words(j);
              cur ^= xl[xl_stage_off(j) + (int)((xr >> (4 * j)) & 15u) * wj + (w & (wj - 1))];
            }
          }
          xl[xl_stage_off(S) + c * W + w] = cur;
        }
      }
      xrow = (xrow & ~(15ull << (4 * S))) | ((uint64_t)(lane & 15) << (4 * S));
      __syncthreads();
    }
    i = il + 1;
  }

  // ---- final ranking by path cost (polar.py:671-678), stable
  {
    int rank = 0;
    unrolled<LMAX>([&](auto rc) __attribute__((always_inline)) {
      constexpr int r = decltype(rc)::value;
      const double cr = lane_value<r>(cost);
      rank += (r < count && (cr < cost || (cr == cost && r < lane))) ? 1 : 0;
    });
    if (lane < count) sel[rank] = lane;
    __syncthreads();
  }
  // ---- trace-back: one lane per list slot walks the fork history backwards
  uint8_t* const ub = smem + lay.llr_off;  // [slot][k_info]
  if (lane < count) {
    int cand = lane;
    for (int k = k_info - 1; k >= 0; --k) {
      const int q = hist[k * LMAX + cand];
      ub[lane * k_info + k] = (uint8_t)(q >> 3);
      cand = q & 7;
    }
  }
  __syncthreads();
  // ---- CRC-aided pick (polar.py:965-977): lane p checks the p-th cheapest candidate
  bool pass = false;
  if (lane < count && crc_id >= 0) {
    const int slot = sel[lane];
    const int CL = crc_len(crc_id);
    const uint32_t low = crc_poly(crc_id) & ((1u << CL) - 1u), mask = (1u << CL) - 1u;
    uint32_t reg = 0;
    for (int m = 0; m < K; ++m) {
      const uint32_t bit = ub[slot * k_info + msg_src[m]];
      const uint32_t topb = ((reg >> (CL - 1)) ^ bit) & 1u;
      reg = ((reg << 1) & mask) ^ (topb ? low : 0u);
    }
    // plain CRC: the register over message + parity ends at zero.  A mask XORed onto the parity bits (a DCI's RNTI, the
    // 24 ones prepended to the CRC input: TS 38.212 7.3.2) shifts that end value by the register value of the mask
    // alone (the CRC is linear), which the caller passes per code word
    pass = (reg == (crc_expect ? crc_expect[cw] : 0u));
  }
  const unsigned long long ok = __ballot(pass);
  const int best = ok ? (__ffsll((long long)ok) - 1) : 0;
  const int best_slot = sel[best];
  for (int m = lane; m < K; m += 64) msg_out[cw * K + m] = ub[best_slot * k_info + msg_src[m]];
  if (lane == 0) crc_ok[cw] = (crc_id < 0) ? 1 : (ok ? 1 : 0);
  if (cand_out) {
    for (int it = lane; it < L * K; it += 64) {
      const int p = it / K, m = it - p * K;
      cand_out[(cw * L + p) * K + m] = (p < count) ? ub[sel[p] * k_info + msg_src[m]] : 0;
    }
  }
  if (cost_out) {
    // slot `lane` holds cost; sorted position p wants the cost of slot sel[p]
    const int slot = (lane < count) ? sel[lane] : 0;
    const double cs = __shfl(cost, slot, 64);
    if (lane < L) cost_out[cw * L + lane] = (lane < count) ? cs : __builtin_inf();
  }
}

}  // namespace

extern "C" {

int32_t nrx_polar_encode(const uint8_t* cbs, int32_t n_cw, int32_t K, int32_t N, const int32_t* in_il,
                         const int32_t* msg_pos, const int32_t* pc_pos, int32_t n_pc, uint8_t* coded, void* stream) {
  NRX_REQUIRE(cbs && msg_pos && coded, NRX_E_ARG, "nrx_polar_encode: null pointer");
  NRX_REQUIRE(n_cw >= 0 && K > 0, NRX_E_SHAPE, "nrx_polar_encode: bad sizes n_cw=%d K=%d", n_cw, K);
  NRX_REQUIRE(N >= 32 && N <= 1024 && (N & (N - 1)) == 0, NRX_E_SHAPE, "nrx_polar_encode: N=%d is not 2^5..2^10", N);
  NRX_REQUIRE(K + n_pc <= N && n_pc >= 0 && (n_pc == 0 || pc_pos), NRX_E_SHAPE,
              "nrx_polar_encode: K=%d + nPC=%d does not fit N=%d", K, n_pc, N);
  if (n_cw == 0) return NRX_OK;
  hipLaunchKernelGGL(polar_encode_kernel, dim3(n_cw), dim3(N >= 512 ? 256 : 64), 0, (hipStream_t)stream, cbs, K, N,
                     in_il, msg_pos, pc_pos, n_pc, coded);
  NRX_CHECK_LAUNCH("nrx_polar_encode");
  return NRX_OK;
}

int32_t nrx_polar_rate_match(const uint8_t* coded, int32_t n_cw, int32_t N, int32_t E, const int32_t* gather,
                             uint8_t* out, void* stream) {
  NRX_REQUIRE(coded && gather && out, NRX_E_ARG, "nrx_polar_rate_match: null pointer");
  NRX_REQUIRE(n_cw >= 0 && N > 0 && E > 0, NRX_E_SHAPE, "nrx_polar_rate_match: bad sizes");
  if (n_cw == 0) return NRX_OK;
  hipLaunchKernelGGL(polar_gather_kernel, dim3(nrx::stream_grid((long)n_cw * E, 256)), dim3(256), 0,
                     (hipStream_t)stream, coded, (int64_t)n_cw, N, E, gather, out);
  NRX_CHECK_LAUNCH("nrx_polar_rate_match");
  return NRX_OK;
}

int32_t nrx_polar_rate_recover_f64(const double* llr, int32_t n_cw, int32_t N, int32_t E, int32_t K,
                                   const int32_t* deinterleave, const int32_t* inv_subblock, double* out,
                                   void* stream) {
  NRX_REQUIRE(llr && inv_subblock && out, NRX_E_ARG, "nrx_polar_rate_recover_f64: null pointer");
  NRX_REQUIRE(n_cw >= 0 && N > 0 && E > 0 && K > 0, NRX_E_SHAPE, "nrx_polar_rate_recover_f64: bad sizes");
  if (n_cw == 0) return NRX_OK;
  const int mode = (E >= N) ? 0 : (((double)K / (double)E <= 7.0 / 16.0) ? 1 : 2);  // polar.py:911-923
  hipLaunchKernelGGL(polar_rate_recover_kernel, dim3(nrx::stream_grid((long)n_cw * N, 256)), dim3(256), 0,
                     (hipStream_t)stream, llr, (int64_t)n_cw, N, E, mode, deinterleave, inv_subblock, out);
  NRX_CHECK_LAUNCH("nrx_polar_rate_recover_f64");
  return NRX_OK;
}

int32_t nrx_polar_scl_decode_f64(const double* llr, int32_t n_cw, int32_t N, int32_t list_size,
                                 const uint8_t* info_mask, int32_t n_info, const int32_t* msg_src, int32_t K,
                                 int32_t crc_poly_id, const uint32_t* crc_expect, uint8_t* msg_out, uint8_t* crc_ok,
                                 uint8_t* cand_out, double* cost_out, void* stream) {
  NRX_REQUIRE(llr && info_mask && msg_src && msg_out && crc_ok, NRX_E_ARG, "nrx_polar_scl_decode_f64: null pointer");
  NRX_REQUIRE(N >= 32 && N <= 1024 && (N & (N - 1)) == 0, NRX_E_SHAPE, "nrx_polar_scl_decode_f64: N=%d is not 2^5..2^10",
              N);
  NRX_REQUIRE(n_cw >= 0 && K > 0 && K <= n_info && n_info <= N, NRX_E_SHAPE,
              "nrx_polar_scl_decode_f64: bad sizes K=%d n_info=%d N=%d", K, n_info, N);
  NRX_REQUIRE(list_size >= 1 && list_size <= LMAX, NRX_E_UNSUPPORTED,
              "nrx_polar_scl_decode_f64: list size %d not in 1..%d", list_size, LMAX);
  NRX_REQUIRE(crc_poly_id >= -1 && crc_poly_id <= NRX_CRC24C, NRX_E_ARG, "nrx_polar_scl_decode_f64: bad CRC id %d",
              crc_poly_id);
  if (n_cw == 0) return NRX_OK;
  int n = 0;
  while ((1 << n) < N) ++n;
  const SclLayout lay = scl_layout(n, n_info);
  if (lay.total > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(polar_scl_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lay.total);
    NRX_REQUIRE(e == hipSuccess, NRX_E_HIP, "nrx_polar_scl_decode_f64: cannot reserve %d bytes of LDS: %s", lay.total,
                hipGetErrorString(e));
  }
  hipLaunchKernelGGL(polar_scl_kernel, dim3(n_cw), dim3(64), lay.total, (hipStream_t)stream, llr, n, list_size,
                     info_mask, n_info, msg_src, K, crc_poly_id, crc_expect, msg_out, crc_ok, cand_out, cost_out);
  NRX_CHECK_LAUNCH("nrx_polar_scl_decode_f64");
  return NRX_OK;
}

}  // extern "C"
