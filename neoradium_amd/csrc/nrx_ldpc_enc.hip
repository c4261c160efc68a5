// CRC, code-block segmentation, LDPC encode, rate match / rate recover, CRC check + merge (gfx950).
//
// These are the integer/byte stages either side of the decoder.  They are streaming, HBM-bound kernels
// (one pass over the bits / LLRs); none of them is a contraction, so no MFMA.  Reference lines are cited per
// kernel.  Bits are one uint8 per bit (the reference's int8 arrays).
#include "gen_ldpc_bg.h"
#include "nrx_common.h"
#include "nrx_crc.h"

namespace {

constexpr int ZMAX = 384;

using nrx::crc_len;
using nrx::crc_poly;

// (a * b) mod g over GF(2), degrees < L.
__host__ __device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b, uint32_t low, int L) {
  const uint32_t mask = (1u << L) - 1u;
  uint32_t r = 0;
  for (int i = L - 1; i >= 0; --i) {
    const uint32_t top = (r >> (L - 1)) & 1u;
    r = ((r << 1) & mask) ^ (top ? low : 0u);
    if ((b >> i) & 1u) r ^= a;
  }
  return r;
}
// x^e mod g
__host__ __device__ __forceinline__ uint32_t gf2_xpow(uint64_t e, uint32_t low, int L) {
  uint32_t result = 1u;          // x^0
  uint32_t base = (L > 1) ? 2u : low;  // x^1 (L>=6 always)
  while (e) {
    if (e & 1) result = gf2_mulmod(result, base, low, L);
    base = gf2_mulmod(base, base, low, L);
    e >>= 1;
  }
  return result;
}

// Workgroup-parallel CRC of one row of n bits (one byte per bit).  Thread t runs the bit-serial register over the
// `chunk` bits that END at n - (T-1-t)*chunk (the front of the row is virtually padded with zero bits, which do not
// change a CRC), reading 16 bytes at a time where the address allows it; the T chunk remainders are then combined in
// a binary tree: at level k a pair is joined as  left * x^(chunk * 2^k) + right  (mod g).  The level multipliers
// depend only on (chunk, polynomial) and are computed on the host (CrcPlan), so no thread ever exponentiates.
// The result equals the reference's long division (chancodebase.py:119-128); every thread gets it.
struct CrcPlan {
  int32_t poly_id, L;
  uint32_t low;
  int32_t threads, chunk;  // threads: power of two, 64..1024
  uint32_t mk[10];         // x^(chunk * 2^k) mod g
};

CrcPlan make_crc_plan(int64_t n_bits, int poly_id, int max_threads = 1024) {
  CrcPlan p;
  p.poly_id = poly_id;
  p.L = crc_len(poly_id);
  p.low = crc_poly(poly_id) & ((1u << p.L) - 1u);
  int t = 64;
  while (t < max_threads && (int64_t)t * 256 < n_bits) t *= 2;   // >= ~256 bits per thread before widening
  p.threads = t;
  int64_t c = (n_bits + t - 1) / t;
  c = (c + 15) / 16 * 16;
  if (c < 16) c = 16;
  p.chunk = (int32_t)c;
  uint32_t m = gf2_xpow((uint64_t)c, p.low, p.L);
  for (int k = 0; k < 10; ++k) {
    p.mk[k] = m;
    m = gf2_mulmod(m, m, p.low, p.L);
  }
  return p;
}

__device__ __forceinline__ uint32_t crc_step(uint32_t reg, uint32_t bit, uint32_t low, uint32_t mask, int L) {
  const uint32_t top = ((reg >> (L - 1)) ^ bit) & 1u;
  return ((reg << 1) & mask) ^ (top ? low : 0u);
}

// `red`: >= 16 words of LDS.  Requires blockDim.x == plan.threads.
__device__ uint32_t block_crc(const uint8_t* __restrict__ row, int64_t n, const CrcPlan& pl, uint32_t* red) {
  const int L = pl.L;
  const uint32_t low = pl.low, mask = (1u << L) - 1u;
  const int nt = blockDim.x, tid = threadIdx.x;
  const int64_t e = n - (int64_t)(nt - 1 - tid) * pl.chunk;   // end of this thread's chunk (exclusive)
  int64_t i = e - pl.chunk;
  if (i < 0) i = 0;
  uint32_t reg = 0;
  for (; i < e && ((uintptr_t)(row + i) & 15u); ++i) reg = crc_step(reg, row[i] & 1u, low, mask, L);
  for (; i + 64 <= e; i += 64) {     // four loads in flight (the register recurrence would otherwise serialise them: load, 16 steps, load)
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(row + i + 16 * u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) reg = crc_step(reg, (w[q] >> (8 * b)) & 1u, low, mask, L);
    }
  }
  for (; i + 16 <= e; i += 16) {
    const uint4 v = *reinterpret_cast<const uint4*>(row + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) reg = crc_step(reg, (w[q] >> (8 * b)) & 1u, low, mask, L);
  }
  for (; i < e; ++i) reg = crc_step(reg, row[i] & 1u, low, mask, L);
  // tree inside the wave
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const uint32_t other = __shfl_xor(reg, 1 << k, 64);
    const bool upper = (tid >> k) & 1;
    reg = gf2_mulmod(upper ? other : reg, pl.mk[k], low, L) ^ (upper ? reg : other);
  }
  const int nw = nt >> 6;
  if (nw == 1) return reg;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = reg;
  __syncthreads();
  uint32_t tot = red[tid & (nw - 1)];   // every wave finishes the tree over the nw wave remainders
  for (int k = 0; (1 << k) < nw; ++k) {
    const uint32_t other = __shfl_xor(tot, 1 << k, 64);
    const bool upper = (tid >> k) & 1;
    tot = gf2_mulmod(upper ? other : tot, pl.mk[6 + k], low, L) ^ (upper ? tot : other);
  }
  __syncthreads();
  return tot;
}

// ---------------------------------------------------------------------------------------------- nrx_crc
__global__ void __launch_bounds__(1024)
crc_rows_kernel(const uint8_t* __restrict__ bits, int64_t row_len, int64_t row_stride, const CrcPlan pl,
                uint8_t* __restrict__ out) {
  __shared__ uint32_t red[16];
  const uint8_t* row = bits + (size_t)blockIdx.x * row_stride;
  const uint32_t r = block_crc(row, row_len, pl, red);
  const int L = pl.L;
  if ((int)threadIdx.x < L) out[(size_t)blockIdx.x * L + threadIdx.x] = (r >> (L - 1 - threadIdx.x)) & 1u;
}

// ------------------------------------------------------------------------------------- nrx_ldpc_segment
// ldpc.py:981-1030 in two launches:
//  1. tb_crc_scatter_kernel: one workgroup (up to 1024 threads) per transport block computes CRC24A
//     (chancodebase.py:161-189 appendCrc('24A')) and writes its 24 bits straight to their final place inside the
//     code-block buffer (bits A..A+23 of the segmented stream: tail of the last block, possibly straddling two);
//  2. segment_kernel: one small workgroup per (tb, code block) copies the payload from the TB, leaves the TB-CRC
//     bits that are already in place, zero-pads, appends CRC24B (C>1) and the zero filler bits.
__global__ void __launch_bounds__(1024)
tb_crc_scatter_kernel(const uint8_t* __restrict__ tb, int A, int C, int K, int per, uint8_t* __restrict__ cbs,
                      const CrcPlan pl) {
  __shared__ uint32_t red[16];
  const int t = blockIdx.x;
  const uint8_t* src = tb + (size_t)t * A;
  const uint32_t crc = block_crc(src, A, pl, red);
  if (threadIdx.x < 24) {
    const int64_t g = (int64_t)A + threadIdx.x;
    const int c = (int)(g / per), o = (int)(g - (int64_t)c * per);
    cbs[((size_t)t * C + c) * K + o] = (crc >> (23 - threadIdx.x)) & 1u;
  }
}

__global__ void __launch_bounds__(256)
segment_kernel(const uint8_t* __restrict__ tb, int A, int B, int C, int K, int cb_len, uint8_t* __restrict__ cbs,
               const CrcPlan pl) {
  __shared__ uint32_t red[16];
  __shared__ __attribute__((aligned(16))) uint8_t stage[8448];   // the block's payload: the CRC runs on this copy, not on a re-read of dst
  const int t = blockIdx.x / C, c = blockIdx.x % C;
  const int per = (B + C - 1) / C;  // payload bits per block before its CRC
  const uint8_t* src = tb + (size_t)t * A;
  uint8_t* dst = cbs + (size_t)blockIdx.x * K;
  const bool staged = per <= (int)sizeof(stage);
  auto copy1 = [&](int i) {
    const int64_t g = (int64_t)c * per + i;
    uint8_t v;
    if (g < A) { v = src[g] & 1; dst[i] = v; }
    else if (g >= B) { v = 0; dst[i] = 0; }  // zero padding at the end of the last block (ldpc.py:1014-1016)
    else v = dst[i];   // A <= g < B: TB-CRC bit, already written by tb_crc_scatter_kernel (or part of the caller's TB when B == A)
    if (staged) stage[i] = v;
  };
  const uint8_t* s0 = src + (int64_t)c * per;
  if ((per & 7) == 0 && (((uintptr_t)s0 | (uintptr_t)dst) & 7u) == 0) {
    // eight bits per load / store; the word that straddles the end of the payload goes bit by bit
    for (int w = threadIdx.x; w < (per >> 3); w += blockDim.x) {
      const int64_t g0 = (int64_t)c * per + 8 * w;
      if (g0 + 8 <= A) {
        const uint64_t v = *(const uint64_t*)(s0 + 8 * w) & 0x0101010101010101ull;
        *(uint64_t*)(dst + 8 * w) = v;
        if (staged) *(uint64_t*)(stage + 8 * w) = v;
      } else {
        for (int k = 0; k < 8; ++k) copy1(8 * w + k);
      }
    }
  } else {
    for (int i = threadIdx.x; i < per; i += blockDim.x) copy1(i);
  }
  __syncthreads();
  if (C > 1) {
    const uint32_t r = block_crc(staged ? (const uint8_t*)stage : (const uint8_t*)dst, per, pl, red);
    if (threadIdx.x < 24) dst[per + threadIdx.x] = (r >> (23 - threadIdx.x)) & 1u;
  }
  for (int i = cb_len + threadIdx.x; i < K; i += blockDim.x) dst[i] = 0;  // fillers are ZERO bits (ldpc.py:1025-1028)
}

// -------------------------------------------------------------------------------------- nrx_ldpc_encode
// ldpc.py:1033-1090.  One workgroup per code block, lane z = bit z of every Zc-column; the 22(10) information
// columns plus the 4 core parity columns sit in LDS; circulant products are LDS reads at (z+shift) mod Zc.
struct EncTab {
  int16_t s[320];
};

template <int BG>
__global__ void __launch_bounds__(ZMAX)
encode_kernel(const uint8_t* __restrict__ cbs, int zc, int puncture, uint8_t* __restrict__ coded, const EncTab tab, int n_rows) {
  constexpr int ROWS = BG == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  constexpr int KB = BG == 1 ? 22 : 10;
  const int16_t* rs = BG == 1 ? kBg1RowStart : kBg2RowStart;
  const int16_t* cl = BG == 1 ? kBg1Col : kBg2Col;
  __shared__ uint8_t w[(KB + 4) * ZMAX];
  __shared__ uint8_t tmp[ZMAX];
  const int z = threadIdx.x;
  const bool act = z < zc;
  const int K = KB * zc;
  const int ncols_out = (BG == 1 ? 68 : 52) - (puncture ? 2 : 0);
  const uint8_t* in = cbs + (size_t)blockIdx.x * K;
  uint8_t* out = coded + (size_t)blockIdx.x * ncols_out * zc;
  const int skip = puncture ? 2 : 0;
  auto rot = [&](int s) { int a = z + s; return a >= zc ? a - zc : a; };

  if (act)
    for (int c = 0; c < KB; ++c) {
      const uint8_t b = in[c * zc + z] & 1;
      w[c * ZMAX + z] = b;
      if (c >= skip) out[(c - skip) * zc + z] = b;
    }
  __syncthreads();
  // information part of rows 0..3
  uint8_t core[4] = {0, 0, 0, 0};
  int sh_p[4][4];  // shift of parity column KB+q in core row i (-1 = none)
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 4; ++q) sh_p[i][q] = -1;
  for (int i = 0; i < 4; ++i)
    for (int e = rs[i]; e < rs[i + 1]; ++e) {
      const int c = cl[e];
      if (c < KB) { if (act) core[i] ^= w[c * ZMAX + rot(tab.s[e])]; }
      else if (c < KB + 4) sh_p[i][c - KB] = tab.s[e];
    }
  // p0 = rot(sum of the four rows, Zc - shift) with shift = bg[1][KB] (or bg[2][KB] when that is -1): ldpc.py:1068-1074
  const int s0 = sh_p[1][0] >= 0 ? sh_p[1][0] : sh_p[2][0];
  if (act) tmp[z] = core[0] ^ core[1] ^ core[2] ^ core[3];
  __syncthreads();
  if (act) w[KB * ZMAX + z] = tmp[rot((zc - s0) % zc)];
  __syncthreads();
  // p1..p3: ldpc.py:1077-1080
  for (int i = 0; i < 3; ++i) {
    if (act) {
      uint8_t v = core[i];
      for (int q = 0; q <= i; ++q)
        if (sh_p[i][q] >= 0) v ^= w[(KB + q) * ZMAX + rot(sh_p[i][q])];
      w[(KB + i + 1) * ZMAX + z] = v;
    }
    __syncthreads();
  }
  if (act)
    for (int q = 0; q < 4; ++q) out[(KB + q - skip) * zc + z] = w[(KB + q) * ZMAX + z];
  // extension parities: ldpc.py:1083-1084
  if (act)
    for (int r = 4; r < n_rows; ++r) {
      uint8_t v = 0;
      for (int e = rs[r]; e < rs[r + 1]; ++e) {
        const int c = cl[e];
        if (c < KB + 4) v ^= w[c * ZMAX + rot(tab.s[e])];
      }
      out[(KB + r - skip) * zc + z] = v;
    }
}

// Bit-packed encoder for lifting sizes that are multiples of 32: one wavefront per code block, a Zc-column is
// NW = Zc/32 words, a circulant product is a word rotation + funnel shift (v_alignbit) instead of Zc byte reads.
// Same equations as encode_kernel (ldpc.py:1033-1090); packing / unpacking the one-byte-per-bit interface is the
// only part that touches memory: bytes -> words with wave ballots, words -> bytes with a 4-bits-per-store expansion.
template <int BG>
__global__ void __launch_bounds__(64)
encode_packed_kernel(const uint8_t* __restrict__ cbs, int zc, int puncture, uint8_t* __restrict__ coded, const EncTab tab,
                     int n_rows) {   // parity of base-graph rows < n_rows only (the others are not transmitted at this rate)
  constexpr int ROWS = BG == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  constexpr int KB = BG == 1 ? 22 : 10;
  constexpr int NWMAX = ZMAX / 32;
  const int16_t* rs = BG == 1 ? kBg1RowStart : kBg2RowStart;
  const int16_t* cl = BG == 1 ? kBg1Col : kBg2Col;
  __shared__ uint32_t W[(KB + 4 + ROWS - 4) * NWMAX];   // info + core parity + extension parity columns
  __shared__ uint32_t acc[4 * NWMAX];                   // information part of the four core rows
  const int lane = threadIdx.x;
  const int nw = zc >> 5;
  const int K = KB * zc;
  const int ncols = BG == 1 ? 68 : 52;
  const int skip = puncture ? 2 : 0;
  const uint8_t* in = cbs + (size_t)blockIdx.x * K;
  uint8_t* out = coded + (size_t)blockIdx.x * (ncols - skip) * zc;
  // word j of the column rotated by s:  out[z] = in[(z + s) mod Zc]
  auto rotw = [&](int col, int s, int j) -> uint32_t {
    const int q = s >> 5, r = s & 31;
    int a = j + q;
    if (a >= nw) a -= nw;
    int b = a + 1;
    if (b >= nw) b -= nw;
    return __builtin_amdgcn_alignbit(W[col * NWMAX + b], W[col * NWMAX + a], r);
  };

  // ---- pack the information columns.  A lane takes 16 bytes (one load), folds each 4 of them into a nibble -- bytes b0..b3
  // of a word are 0/1, (w * 0x10204080) >> 28 = b0 | b1<<1 | b2<<2 | b3<<3: the sixteen partial products land on distinct bits
  // -- and stores its 16 bits as one half word of W; all loads of the lane are issued before the first is used.  (The first
  // version went column by column with a wave ballot per 64 bytes: 132 dependent load -> ballot round trips per code block.)
  if ((zc & 15) == 0 && (((uintptr_t)in) & 15u) == 0) {
    const int cpc = zc >> 4;                      // 16-byte chunks per column
    const int n_chunk = KB * cpc;
    constexpr int MAXR = (KB * (ZMAX / 16) + 63) / 64;
    uint4 ld[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int t = lane + 64 * r;
      ld[r] = t < n_chunk ? ((const uint4*)in)[t] : uint4{0u, 0u, 0u, 0u};
    }
    __builtin_amdgcn_sched_barrier(0);
    uint16_t* W16 = (uint16_t*)W;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int t = lane + 64 * r;
      if (t < n_chunk) {
        const int c = t / cpc, p = t - c * cpc;
        auto nib = [](uint32_t w) -> uint32_t { return ((w & 0x01010101u) * 0x10204080u) >> 28; };
        W16[c * NWMAX * 2 + p] = (uint16_t)(nib(ld[r].x) | (nib(ld[r].y) << 4) | (nib(ld[r].z) << 8) | (nib(ld[r].w) << 12));
      }
    }
  } else {
    for (int c = 0; c < KB; ++c)      // 64 bits per ballot
      for (int k = 0; k < zc; k += 64) {
        const int z = k + lane;
        const unsigned long long m = __ballot(z < zc && (in[c * zc + (z < zc ? z : 0)] & 1));
        if (lane == 0) {
          W[c * NWMAX + (k >> 5)] = (uint32_t)m;
          if (k + 32 < zc) W[c * NWMAX + (k >> 5) + 1] = (uint32_t)(m >> 32);
        }
      }
  }
  __syncthreads();
  // ---- information part of core rows 0..3 (lane = (row, word))
  int sh_p[4][4];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 4; ++q) sh_p[i][q] = -1;
  for (int i = 0; i < 4; ++i)
    for (int e = rs[i]; e < rs[i + 1]; ++e) {
      const int c = cl[e];
      if (c >= KB && c < KB + 4) sh_p[i][c - KB] = tab.s[e];
    }
  if (lane < 4 * nw) {
    const int i = lane / nw, j = lane - i * nw;
    uint32_t v = 0;
    for (int e = rs[i]; e < rs[i + 1]; ++e)
      if (cl[e] < KB) v ^= rotw(cl[e], tab.s[e], j);
    acc[i * NWMAX + j] = v;
  }
  __syncthreads();
  // p0 = rot(sum of the four rows, Zc - shift): ldpc.py:1068-1074
  const int s0 = sh_p[1][0] >= 0 ? sh_p[1][0] : sh_p[2][0];
  if (lane < nw) W[(KB + 4) * NWMAX + lane] = acc[lane] ^ acc[NWMAX + lane] ^ acc[2 * NWMAX + lane] ^ acc[3 * NWMAX + lane];
  __syncthreads();
  if (lane < nw) W[KB * NWMAX + lane] = rotw(KB + 4, (zc - s0) % zc, lane);   // column KB+4 is scratch here
  __syncthreads();
  // p1..p3: ldpc.py:1077-1080
  for (int i = 0; i < 3; ++i) {
    if (lane < nw) {
      uint32_t v = acc[i * NWMAX + lane];
      for (int q = 0; q <= i; ++q)
        if (sh_p[i][q] >= 0) v ^= rotw(KB + q, sh_p[i][q], lane);
      W[(KB + i + 1) * NWMAX + lane] = v;
    }
    __syncthreads();
  }
  // ---- extension parities (ldpc.py:1083-1084): items (row, word) over the wave
  for (int it = lane; it < (n_rows - 4) * nw; it += 64) {
    const int r = 4 + it / nw, j = it - (r - 4) * nw;
    uint32_t v = 0;
    for (int e = rs[r]; e < rs[r + 1]; ++e) {
      const int c = cl[e];
      if (c < KB + 4) v ^= rotw(c, tab.s[e], j);
    }
    W[(KB + r) * NWMAX + j] = v;
  }
  __syncthreads();
  // ---- unpack: 4 bits -> 4 bytes per store (nibble * 0x00204081 puts bit k at bit 8k)
  const int quads = zc >> 2;                                // 4-byte stores per column
  for (int it = lane; it < (KB + n_rows - skip) * quads; it += 64) {
    const int c = it / quads + skip, qd = it - (c - skip) * quads;
    const uint32_t nib = (W[c * NWMAX + (qd >> 3)] >> ((qd & 7) * 4)) & 15u;
    *reinterpret_cast<uint32_t*>(out + (size_t)(c - skip) * zc + 4 * qd) = (nib * 0x00204081u) & 0x01010101u;
  }
}

// ---------------------------------------------------------------------------------- nrx_ldpc_rate_match
// ldpc.py:1093-1159.  One thread per output bit: locate the code block r and position within E_r, undo the
// bit interleaver (out[i*qm+q] = sel[q*(E/qm)+i]), map through the filler-free circular buffer at k0.
struct RmGeom {
  int C, N, K, F, zc, ncb, k0, sys_len, cs;  // cs = ncb - F (circular buffer without fillers)
  int e_small, n_small, f;                   // first n_small blocks have e_small bits, the rest e_small+f
  int G, qm;
  int k0_tab[4];                             // k0 of every redundancy version (per-transport-block rv arrays)
};

__device__ __forceinline__ void rm_locate(const RmGeom& g, int pos, int& r, int& j, int& E) {
  const int split = g.n_small * g.e_small;
  if (pos < split) {
    r = pos / g.e_small;
    j = pos - r * g.e_small;
    E = g.e_small;
  } else {
    const int e2 = g.e_small + g.f;
    r = g.n_small + (pos - split) / e2;
    j = (pos - split) - (r - g.n_small) * e2;
    E = e2;
  }
}

__global__ void __launch_bounds__(256)
rate_match_kernel(const uint8_t* __restrict__ coded, int n_tb, RmGeom g, uint8_t* __restrict__ out,
                  const int32_t* __restrict__ rvs) {
  const int64_t total = (int64_t)n_tb * g.G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / g.G), pos = (int)(i - (int64_t)t * g.G);
    int r, j, E;
    rm_locate(g, pos, r, j, E);
    const int eq = E / g.qm;
    const int e = (j % g.qm) * eq + j / g.qm;        // position in the selected (pre-interleaver) sequence
    const int k0 = rvs ? g.k0_tab[rvs[t] & 3] : g.k0;
    const int ci = (e + k0) % g.cs;                  // circular buffer without fillers
    const int src = ci < g.sys_len ? ci : ci + g.F;  // index in the N-bit coded block (fillers skipped)
    out[i] = coded[((size_t)t * g.C + r) * g.N + src];
  }
}

// One (x-strip, code block) per workgroup and the modulation order as a template parameter: no 64-bit divisions,
// a thread handles the QM bits of one modulation symbol (consecutive output bytes), and for a fixed bit plane the
// lanes read consecutive coded bits.
template <int QM>
__global__ void __launch_bounds__(256)
rate_match_cb_kernel(const uint8_t* __restrict__ coded, RmGeom g, uint8_t* __restrict__ out,
                     const int32_t* __restrict__ rvs) {
  const int cbi = blockIdx.y;
  const int t = cbi / g.C, r = cbi - t * g.C;
  const int k0 = rvs ? g.k0_tab[rvs[t] & 3] : g.k0;
  int E, off;
  if (r < g.n_small) { E = g.e_small; off = r * g.e_small; }
  else { E = g.e_small + g.f; off = g.n_small * g.e_small + (r - g.n_small) * E; }
  const int eq = E / QM;
  const uint8_t* src = coded + (size_t)cbi * g.N;
  uint8_t* dst = out + (size_t)t * g.G + off;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  // Four symbols per thread where the geometry allows 32-bit accesses (bit plane q of four consecutive symbols = four
  // consecutive coded bits: one load when they neither wrap nor straddle the fillers and sit on a word boundary; the 4 x QM
  // output bytes are consecutive: QM word stores), one modulo per bit plane and thread instead of one per bit.
  int n_vec = 0;
  if ((((uintptr_t)src | (uintptr_t)dst) & 3u) == 0) {
    n_vec = eq >> 2;
    int bq[QM];     // (q * eq + k0) mod cs, by additions: a 64-bit modulo is ~130 instructions
    const int eqm = (int)((uint32_t)eq % (uint32_t)g.cs);
    bq[0] = (int)((uint32_t)k0 % (uint32_t)g.cs);
#pragma unroll
    for (int q = 1; q < QM; ++q) {
      bq[q] = bq[q - 1] + eqm;
      bq[q] -= bq[q] >= g.cs ? g.cs : 0;
    }
    for (int v = tid; v < n_vec; v += nth) {
      const int sidx0 = 4 * v;
      uint32_t w[QM];
      int ci[QM], pp[QM];
      bool fast = true;
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        ci[q] = bq[q] + sidx0;
        while (ci[q] >= g.cs) ci[q] -= g.cs;
        pp[q] = ci[q] < g.sys_len ? ci[q] : ci[q] + g.F;
        fast = fast && ci[q] + 3 < g.cs && (ci[q] >= g.sys_len || ci[q] + 3 < g.sys_len) && (pp[q] & 3) == 0;
      }
      if (fast) {     // (one branch for the thread, so that its QM loads are issued together)
#pragma unroll
        for (int q = 0; q < QM; ++q) w[q] = *(const uint32_t*)(src + pp[q]);
      } else {
#pragma unroll
        for (int q = 0; q < QM; ++q) {
          w[q] = 0;
          for (int k = 0; k < 4; ++k) {
            int c2 = ci[q] + k;
            if (c2 >= g.cs) c2 -= g.cs;
            w[q] |= (uint32_t)src[c2 < g.sys_len ? c2 : c2 + g.F] << (8 * k);
          }
        }
      }
      // byte k of w[q] = bit plane q of symbol sidx0 + k  ->  output byte (sidx0 + k) * QM + q
      uint32_t o[QM];
#pragma unroll
      for (int j = 0; j < QM; ++j) {
        uint32_t x = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int n = 4 * j + b, k = n / QM, q = n - k * QM;
          x |= ((w[q] >> (8 * k)) & 0xffu) << (8 * b);
        }
        o[j] = x;
      }
#pragma unroll
      for (int j = 0; j < QM; ++j) *(uint32_t*)(dst + (size_t)sidx0 * QM + 4 * j) = o[j];
    }
  }
  for (int sidx = 4 * n_vec + tid; sidx < eq; sidx += nth) {
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      const int e = q * eq + sidx;
      const int ci = (e + k0) % g.cs;
      dst[sidx * QM + q] = src[ci < g.sys_len ? ci : ci + g.F];
    }
  }
}

// Rate recovery without wrap-around repetition (E_r <= circular buffer): every buffer position receives at most one
// LLR, so the de-interleaver can run as a coalesced scatter; untransmitted positions, fillers and the part of the
// code block beyond a limited buffer are filled by the same workgroup.  Same sums as rate_recover_kernel.
template <typename T, int QM>
__global__ void __launch_bounds__(256)
rate_recover_cb_kernel(const T* __restrict__ llr, int llr_len, RmGeom g, T* __restrict__ circ, T* __restrict__ out,
                       const int32_t* __restrict__ rvs, const uint8_t* __restrict__ reset) {
  const int cbi = blockIdx.y;
  const int t = cbi / g.C, r = cbi - t * g.C;
  const int k0 = rvs ? g.k0_tab[rvs[t] & 3] : g.k0;
  const bool fresh = reset && reset[t];     // new transport block in this HARQ process: the soft buffer starts from 0
  int E, off;
  if (r < g.n_small) { E = g.e_small; off = r * g.e_small; }
  else { E = g.e_small + g.f; off = g.n_small * g.e_small + (r - g.n_small) * E; }
  const int eq = E / QM;
  const T* src = llr + (size_t)t * llr_len;
  T* dst = out + (size_t)cbi * g.N;
  T* cb = circ ? circ + (size_t)cbi * g.cs : nullptr;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int sidx = tid; sidx < eq; sidx += nth) {
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      const int e = q * eq + sidx;
      const int ci = (e + k0) % g.cs;
      const int j = off + sidx * QM + q;
      T v = j < llr_len ? src[j] : (T)0;   // short input is zero padded (ldpc.py:1401-1402)
      if (cb) { v = (fresh ? (T)0 : cb[ci]) + v; cb[ci] = v; }
      dst[ci < g.sys_len ? ci : ci + g.F] = v;
    }
  }
  for (int e = E + tid; e < g.cs; e += nth) {             // buffer positions this transmission did not reach
    const int ci = (e + k0) % g.cs;
    T v = (T)0;
    if (cb) {
      if (fresh) cb[ci] = (T)0; else v = cb[ci];
    }
    dst[ci < g.sys_len ? ci : ci + g.F] = v;
  }
  for (int i = tid; i < g.F; i += nth) dst[g.sys_len + i] = (T)1e20;   // LARGE_LLR fillers (ldpc.py:1414-1418)
  for (int n = g.cs + g.F + tid; n < g.N; n += nth) dst[n] = (T)0;      // beyond a limited (LBRM) buffer
}

// -------------------------------------------------------------------------------- nrx_ldpc_rate_recover
// ldpc.py:1330-1418 as a gather: one thread per (code block, coded position).  Accumulation order over the
// wrap-around repetitions follows the reference (increasing e), so float sums are bit-identical.
template <typename T>
__global__ void __launch_bounds__(256)
rate_recover_kernel(const T* __restrict__ llr, int n_tb, int llr_len, RmGeom g, T* __restrict__ circ, T* __restrict__ out,
                    const int32_t* __restrict__ rvs, const uint8_t* __restrict__ reset) {
  const int64_t total = (int64_t)n_tb * g.C * g.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % g.N);
    const int64_t cbi = i / g.N;
    const int r = (int)(cbi % g.C), t = (int)(cbi / g.C);
    T v;
    if (n >= g.sys_len && n < g.sys_len + g.F) {
      v = (T)1e20;  // LARGE_LLR for filler positions (chancodebase.py:52, ldpc.py:1414-1418)
    } else {
      const int ci = n < g.sys_len ? n : n - g.F;
      v = (T)0;
      if (ci < g.cs) {  // positions beyond the (LBRM-limited) buffer do not exist; out is N wide only when ncb==N
        T acc = (circ && !(reset && reset[t])) ? circ[cbi * g.cs + ci] : (T)0;
        // E_r and the offset of block r inside the G-long LLR stream
        int E, off;
        if (r < g.n_small) { E = g.e_small; off = r * g.e_small; }
        else { E = g.e_small + g.f; off = g.n_small * g.e_small + (r - g.n_small) * E; }
        const int eq = E / g.qm;
        int e = ci - (rvs ? g.k0_tab[rvs[t] & 3] : g.k0);
        e %= g.cs;
        if (e < 0) e += g.cs;
        for (; e < E; e += g.cs) {
          const int j = (e % eq) * g.qm + e / eq;  // de-interleave: x[e] = rx[(e mod E/qm)*qm + e div (E/qm)]
          const int src = off + j;
          acc += (src < llr_len) ? llr[(size_t)t * llr_len + src] : (T)0;  // short input is zero padded
        }
        if (circ) circ[cbi * g.cs + ci] = acc;
        v = acc;
      }
    }
    out[i] = v;
  }
}

// ----------------------------------------------------------------------------------- nrx_ldpc_crc_merge
// ldpc.py:1584-1619.  One workgroup per code block.
__global__ void __launch_bounds__(1024)
crc_merge_kernel(const uint8_t* __restrict__ dec, int C, int K, int cb_len, int B, uint8_t* __restrict__ tb_out,
                 uint8_t* __restrict__ cb_ok, const CrcPlan pl) {
  __shared__ uint32_t red[16];
  const int t = blockIdx.x / C, c = blockIdx.x % C;
  const uint8_t* row = dec + (size_t)blockIdx.x * K;
  const int payload = C > 1 ? cb_len - 24 : cb_len;
  if (tb_out) {
    // merged stream: C*payload >= B bits (the zero padding of the last block is kept, like the reference)
    uint8_t* dst = tb_out + (size_t)t * C * payload + (size_t)c * payload;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) dst[i] = row[i] & 1;
  }
  const uint32_t r = block_crc(row, cb_len, pl, red);
  if (threadIdx.x == 0) cb_ok[blockIdx.x] = r == 0 ? 1 : 0;
}

__global__ void __launch_bounds__(1024)
crc_ok_rows_kernel(const uint8_t* __restrict__ bits, int64_t row_len, int64_t row_stride, const CrcPlan pl,
                   uint8_t* __restrict__ ok) {
  __shared__ uint32_t red[16];
  const uint8_t* row = bits + (size_t)blockIdx.x * row_stride;
  const uint32_t r = block_crc(row, row_len, pl, red);
  if (threadIdx.x == 0) ok[blockIdx.x] = r == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------- nrx_count_errors
__global__ void __launch_bounds__(256)
count_errors_kernel(const uint8_t* __restrict__ cb_ok, int n_ok, const uint8_t* __restrict__ tb_out,
                    const uint8_t* __restrict__ tb_ref, int n_tb, int A, int stride, unsigned long long* counters) {
  // grid: (x-blocks, n_tb).  Row blockIdx.y of the transport blocks is compared 8 bytes (= 8 bits) at a time.
  unsigned long long be = 0, bits = 0;
  const int t = blockIdx.y;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  if (t == 0)
    for (int64_t i = gtid; i < n_ok; i += gsz) be += cb_ok[i] ? 0 : 1;
  if (tb_out && tb_ref) {
    const uint8_t* a = tb_out + (size_t)t * stride;
    const uint8_t* r = tb_ref + (size_t)t * A;
    if ((((uintptr_t)a | (uintptr_t)r) & 7u) == 0) {
      const int n8 = A >> 3;
      for (int64_t i = gtid; i < n8; i += gsz) {
        const unsigned long long x = *reinterpret_cast<const unsigned long long*>(a + 8 * i) ^
                                     *reinterpret_cast<const unsigned long long*>(r + 8 * i);
        bits += __popcll(x & 0x0101010101010101ull);
      }
      for (int64_t k = (int64_t)n8 * 8 + gtid; k < A; k += gsz) bits += ((a[k] ^ r[k]) & 1) ? 1 : 0;
    } else {
      for (int64_t k = gtid; k < A; k += gsz) bits += ((a[k] ^ r[k]) & 1) ? 1 : 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    be += __shfl_xor(be, o, 64);
    bits += __shfl_xor(bits, o, 64);
  }
  __shared__ unsigned long long red[2][4];   // one atomic per workgroup (they all hit the same two words)
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = be;
    red[1][threadIdx.x >> 6] = bits;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    be = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    bits = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (be) atomicAdd(&counters[0], be);
    if (bits) atomicAdd(&counters[2], bits);
  }
  if (gtid == 0 && t == 0) {
    atomicAdd(&counters[1], (unsigned long long)n_ok);
    if (tb_out && tb_ref) atomicAdd(&counters[3], (unsigned long long)n_tb * A);
  }
}

int fill_geom(const nrx_ldpc_cfg* cfg, int G, int nl, int qm, int rv, int n_ref, RmGeom* g) {
  static const int k0num[2][4] = {{0, 17, 33, 56}, {0, 13, 25, 43}};
  g->C = cfg->C; g->N = cfg->N; g->K = cfg->K; g->F = cfg->F; g->zc = cfg->Zc;
  g->ncb = n_ref == 0 ? cfg->N : (cfg->N < n_ref ? cfg->N : n_ref);       // ldpc.py:1138
  g->sys_len = cfg->K - 2 * cfg->Zc - cfg->F;                             // systematic bits w/o fillers
  g->cs = g->ncb - cfg->F;
  g->k0 = (int)(((long)k0num[cfg->bg - 1][rv] * g->ncb / cfg->N) * cfg->Zc);  // ldpc.py:1145
  for (int v = 0; v < 4; ++v) g->k0_tab[v] = (int)(((long)k0num[cfg->bg - 1][v] * g->ncb / cfg->N) * cfg->Zc);
  g->f = nl * qm;
  const int gb = (G + g->f - 1) / g->f;
  g->e_small = (gb / cfg->C) * g->f;
  g->n_small = cfg->C - gb % cfg->C;
  g->G = G; g->qm = qm;
  return 0;
}

}  // namespace

extern "C" int32_t nrx_crc(const uint8_t* bits, int32_t n_rows, int64_t row_len, int64_t row_stride, int32_t poly_id,
                           uint8_t* crc_out, void* stream) {
  NRX_REQUIRE(bits && crc_out, NRX_E_ARG, "nrx_crc: NULL buffer");
  NRX_REQUIRE(poly_id >= NRX_CRC6 && poly_id <= NRX_CRC24C, NRX_E_ARG, "nrx_crc: unknown polynomial id %d", poly_id);
  NRX_REQUIRE(n_rows >= 0 && row_len >= 0 && row_stride >= row_len, NRX_E_SHAPE, "nrx_crc: bad row geometry");
  if (n_rows == 0) return NRX_OK;
  const CrcPlan pl = make_crc_plan(row_len, poly_id);
  hipLaunchKernelGGL(crc_rows_kernel, dim3(n_rows), dim3(pl.threads), 0, (hipStream_t)stream, bits, row_len, row_stride,
                     pl, crc_out);
  NRX_CHECK_LAUNCH("nrx_crc");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_segment(const uint8_t* tb, int32_t n_tb, int32_t A, int32_t add_tb_crc,
                                    const nrx_ldpc_cfg* cfg, uint8_t* cbs, void* stream) {
  NRX_REQUIRE(tb && cfg && cbs, NRX_E_ARG, "nrx_ldpc_segment: NULL buffer");
  NRX_REQUIRE(n_tb >= 0 && A > 0, NRX_E_ARG, "nrx_ldpc_segment: bad sizes");
  NRX_REQUIRE(cfg->B == A + (add_tb_crc ? 24 : 0), NRX_E_SHAPE, "nrx_ldpc_segment: cfg->B (%d) != A (%d) + TB CRC",
              cfg->B, A);
  if (n_tb == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int per = (cfg->B + cfg->C - 1) / cfg->C;
  if (add_tb_crc) {
    const CrcPlan pa = make_crc_plan(A, NRX_CRC24A);
    hipLaunchKernelGGL(tb_crc_scatter_kernel, dim3(n_tb), dim3(pa.threads), 0, st, tb, A, cfg->C, cfg->K, per, cbs, pa);
  }
  const CrcPlan pb = make_crc_plan(per, NRX_CRC24B, 256);
  hipLaunchKernelGGL(segment_kernel, dim3(n_tb * cfg->C), dim3(pb.threads), 0, st, tb, A, cfg->B, cfg->C, cfg->K,
                     cfg->cb_len, cbs, pb);
  NRX_CHECK_LAUNCH("nrx_ldpc_segment");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_encode(const uint8_t* cbs, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t puncture,
                                   int32_t n_rows, uint8_t* coded, void* stream) {
  NRX_REQUIRE(cbs && cfg && coded, NRX_E_ARG, "nrx_ldpc_encode: NULL buffer");
  NRX_REQUIRE(cfg->bg == 1 || cfg->bg == 2, NRX_E_ARG, "nrx_ldpc_encode: bg must be 1|2");
  NRX_REQUIRE(cfg->Zc >= 2 && cfg->Zc <= ZMAX && cfg->iLS >= 0 && cfg->iLS < 8, NRX_E_ARG, "nrx_ldpc_encode: bad Zc/iLS");
  NRX_REQUIRE(n_cb >= 0, NRX_E_ARG, "nrx_ldpc_encode: negative count");
  const int rows_all = cfg->bg == 1 ? NRX_BG1_ROWS : NRX_BG2_ROWS;
  NRX_REQUIRE(n_rows == 0 || (n_rows >= 4 && n_rows <= rows_all), NRX_E_ARG, "nrx_ldpc_encode: n_rows must be 0 (all) or 4..%d", rows_all);
  if (n_rows == 0) n_rows = rows_all;
  if (n_cb == 0) return NRX_OK;
  EncTab tab;
  memset(&tab, 0, sizeof(tab));
  const int E = cfg->bg == 1 ? NRX_BG1_EDGES : NRX_BG2_EDGES;
  for (int e = 0; e < E; ++e)
    tab.s[e] = (int16_t)((cfg->bg == 1 ? kBg1Shift[cfg->iLS][e] : kBg2Shift[cfg->iLS][e]) % cfg->Zc);
  const int threads = ((cfg->Zc + 63) / 64) * 64;
  // output rows are (N or N+2Zc) bytes apart: 4-byte stores need that, and the buffer itself, 4-byte aligned
  if (cfg->Zc % 32 == 0 && ((uintptr_t)coded & 3u) == 0) {
    if (cfg->bg == 1)
      hipLaunchKernelGGL(encode_packed_kernel<1>, dim3(n_cb), dim3(64), 0, (hipStream_t)stream, cbs, cfg->Zc, puncture, coded, tab, n_rows);
    else
      hipLaunchKernelGGL(encode_packed_kernel<2>, dim3(n_cb), dim3(64), 0, (hipStream_t)stream, cbs, cfg->Zc, puncture, coded, tab, n_rows);
    NRX_CHECK_LAUNCH("nrx_ldpc_encode");
    return NRX_OK;
  }
  if (cfg->bg == 1)
    hipLaunchKernelGGL(encode_kernel<1>, dim3(n_cb), dim3(threads), 0, (hipStream_t)stream, cbs, cfg->Zc, puncture, coded, tab, n_rows);
  else
    hipLaunchKernelGGL(encode_kernel<2>, dim3(n_cb), dim3(threads), 0, (hipStream_t)stream, cbs, cfg->Zc, puncture, coded, tab, n_rows);
  NRX_CHECK_LAUNCH("nrx_ldpc_encode");
  return NRX_OK;
}

static int32_t check_rm_args(const char* who, const nrx_ldpc_cfg* cfg, int G, int nl, int qm, int rv) {
  NRX_REQUIRE(cfg, NRX_E_ARG, "%s: NULL cfg", who);
  NRX_REQUIRE(rv >= 0 && rv <= 3, NRX_E_ARG, "%s: Invalid 'rv' value! It must be one of 0, 1, 2, or 3.", who);
  NRX_REQUIRE(nl >= 1 && qm >= 1 && G > 0, NRX_E_ARG, "%s: bad G/nl/qm", who);
  const int f = nl * qm;
  const int gb = (G + f - 1) / f;
  NRX_REQUIRE(gb / cfg->C > 0, NRX_E_SHAPE, "%s: G=%d too small for %d code blocks", who, G, cfg->C);
  return NRX_OK;
}

static int32_t rate_match_impl(const uint8_t* coded, int32_t n_tb, const nrx_ldpc_cfg* cfg, int32_t G, int32_t nl,
                               int32_t qm, int32_t rv, const int32_t* rvs, int32_t n_ref, uint8_t* out, void* stream) {
  NRX_REQUIRE(coded && out, NRX_E_ARG, "nrx_ldpc_rate_match: NULL buffer");
  int32_t rc = check_rm_args("nrx_ldpc_rate_match", cfg, G, nl, qm, rv);
  if (rc) return rc;
  if (n_tb == 0) return NRX_OK;
  RmGeom g;
  fill_geom(cfg, G, nl, qm, rv, n_ref, &g);
  g.G = ((G + nl * qm - 1) / (nl * qm)) * (nl * qm);  // = sum of E_r (ldpc.py:852-855); equals G for PDSCH
  {
    const int eq_max = (g.e_small + g.f) / qm;
    // x-strips per code block: a thread takes four symbols (vector path), and a workgroup should have a few microseconds of
    // work -- 150 k workgroups of one symbol per thread were bound by their own dispatch
    const int strips = (eq_max / 4 + 1023) / 1024;
    const dim3 grid2(strips > 8 ? 8 : (strips < 1 ? 1 : strips), n_tb * cfg->C);
    bool done = true;
    switch (qm) {
      case 1: hipLaunchKernelGGL(rate_match_cb_kernel<1>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      case 2: hipLaunchKernelGGL(rate_match_cb_kernel<2>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      case 4: hipLaunchKernelGGL(rate_match_cb_kernel<4>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      case 6: hipLaunchKernelGGL(rate_match_cb_kernel<6>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      case 8: hipLaunchKernelGGL(rate_match_cb_kernel<8>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      case 10: hipLaunchKernelGGL(rate_match_cb_kernel<10>, grid2, dim3(256), 0, (hipStream_t)stream, coded, g, out, rvs); break;
      default: done = false;
    }
    if (done) {
      NRX_CHECK_LAUNCH("nrx_ldpc_rate_match");
      return NRX_OK;
    }
  }
  hipLaunchKernelGGL(rate_match_kernel, dim3(nrx::stream_grid((long)n_tb * g.G, 256)), dim3(256), 0, (hipStream_t)stream,
                     coded, n_tb, g, out, rvs);
  NRX_CHECK_LAUNCH("nrx_ldpc_rate_match");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_rate_match(const uint8_t* coded, int32_t n_tb, const nrx_ldpc_cfg* cfg, int32_t G,
                                       int32_t nl, int32_t qm, int32_t rv, int32_t n_ref, uint8_t* out, void* stream) {
  return rate_match_impl(coded, n_tb, cfg, G, nl, qm, rv, nullptr, n_ref, out, stream);
}
extern "C" int32_t nrx_ldpc_rate_match_harq(const uint8_t* coded, int32_t n_tb, const nrx_ldpc_cfg* cfg, int32_t G,
                                            int32_t nl, int32_t qm, const int32_t* rv_per_tb, int32_t n_ref,
                                            uint8_t* out, void* stream) {
  NRX_REQUIRE(rv_per_tb, NRX_E_ARG, "nrx_ldpc_rate_match_harq: NULL rv array");
  return rate_match_impl(coded, n_tb, cfg, G, nl, qm, 0, rv_per_tb, n_ref, out, stream);
}

template <typename T>
static int32_t rate_recover_entry(const T* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                  int32_t qm, int32_t rv, int32_t n_ref, T* circ, T* out, void* stream,
                                  const int32_t* rvs = nullptr, const uint8_t* reset = nullptr) {
  NRX_REQUIRE(llr && out, NRX_E_ARG, "nrx_ldpc_rate_recover: NULL buffer");
  int32_t rc = check_rm_args("nrx_ldpc_rate_recover", cfg, llr_len, nl, qm, rv);
  if (rc) return rc;
  if (n_tb == 0) return NRX_OK;
  RmGeom g;
  fill_geom(cfg, llr_len, nl, qm, rv, n_ref, &g);
  if (g.e_small + g.f <= g.cs) {   // no wrap-around repetition: coalesced scatter form
    const dim3 grid2(8, n_tb * cfg->C);
    hipStream_t st = (hipStream_t)stream;
    bool done = true;
    switch (qm) {
      case 1: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 1>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      case 2: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 2>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      case 4: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 4>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      case 6: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 6>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      case 8: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 8>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      case 10: hipLaunchKernelGGL((rate_recover_cb_kernel<T, 10>), grid2, dim3(256), 0, st, llr, llr_len, g, circ, out, rvs, reset); break;
      default: done = false;
    }
    if (done) {
      NRX_CHECK_LAUNCH("nrx_ldpc_rate_recover");
      return NRX_OK;
    }
  }
  hipLaunchKernelGGL(rate_recover_kernel<T>, dim3(nrx::stream_grid((long)n_tb * cfg->C * cfg->N, 256)), dim3(256), 0,
                     (hipStream_t)stream, llr, n_tb, llr_len, g, circ, out, rvs, reset);
  NRX_CHECK_LAUNCH("nrx_ldpc_rate_recover");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_rate_recover_f32(const float* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                             int32_t nl, int32_t qm, int32_t rv, int32_t n_ref, float* circ, float* out,
                                             void* stream) {
  return rate_recover_entry<float>(llr, n_tb, llr_len, cfg, nl, qm, rv, n_ref, circ, out, stream);
}
extern "C" int32_t nrx_ldpc_rate_recover_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                             int32_t nl, int32_t qm, int32_t rv, int32_t n_ref, double* circ,
                                             double* out, void* stream) {
  return rate_recover_entry<double>(llr, n_tb, llr_len, cfg, nl, qm, rv, n_ref, circ, out, stream);
}

extern "C" int32_t nrx_ldpc_rate_recover_harq_f32(const float* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                                  int32_t nl, int32_t qm, const int32_t* rv_per_tb,
                                                  const uint8_t* reset_per_tb, int32_t n_ref, float* circ, float* out,
                                                  void* stream) {
  NRX_REQUIRE(rv_per_tb && circ, NRX_E_ARG, "nrx_ldpc_rate_recover_harq: NULL rv array / soft buffer");
  return rate_recover_entry<float>(llr, n_tb, llr_len, cfg, nl, qm, 0, n_ref, circ, out, stream, rv_per_tb, reset_per_tb);
}
extern "C" int32_t nrx_ldpc_rate_recover_harq_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                                  int32_t nl, int32_t qm, const int32_t* rv_per_tb,
                                                  const uint8_t* reset_per_tb, int32_t n_ref, double* circ, double* out,
                                                  void* stream) {
  NRX_REQUIRE(rv_per_tb && circ, NRX_E_ARG, "nrx_ldpc_rate_recover_harq: NULL rv array / soft buffer");
  return rate_recover_entry<double>(llr, n_tb, llr_len, cfg, nl, qm, 0, n_ref, circ, out, stream, rv_per_tb, reset_per_tb);
}

extern "C" int32_t nrx_ldpc_crc_merge(const uint8_t* dec, int32_t n_tb, const nrx_ldpc_cfg* cfg, uint8_t* tb_out,
                                      uint8_t* cb_ok, uint8_t* tb_ok, void* stream) {
  NRX_REQUIRE(dec && cfg && cb_ok, NRX_E_ARG, "nrx_ldpc_crc_merge: NULL buffer");
  NRX_REQUIRE(!tb_ok || tb_out, NRX_E_ARG, "nrx_ldpc_crc_merge: tb_ok needs tb_out");
  if (n_tb == 0) return NRX_OK;
  hipStream_t st = (hipStream_t)stream;
  const CrcPlan pc = make_crc_plan(cfg->cb_len, cfg->C > 1 ? NRX_CRC24B : NRX_CRC24A, 256);
  hipLaunchKernelGGL(crc_merge_kernel, dim3(n_tb * cfg->C), dim3(pc.threads), 0, st, dec, cfg->C, cfg->K, cfg->cb_len,
                     cfg->B, tb_out, cb_ok, pc);
  if (tb_ok) {
    const CrcPlan pt = make_crc_plan(cfg->B, NRX_CRC24A);
    hipLaunchKernelGGL(crc_ok_rows_kernel, dim3(n_tb), dim3(pt.threads), 0, st, tb_out, (int64_t)cfg->B,
                       (int64_t)cfg->C * (cfg->cb_len - (cfg->C > 1 ? 24 : 0)), pt, tb_ok);
  }
  NRX_CHECK_LAUNCH("nrx_ldpc_crc_merge");
  return NRX_OK;
}

extern "C" int32_t nrx_count_errors(const uint8_t* cb_ok, int32_t n_ok, const uint8_t* tb_out, const uint8_t* tb_ref,
                                    int32_t n_tb, int32_t A, int32_t tb_out_stride, int64_t* counters, void* stream) {
  NRX_REQUIRE(cb_ok && counters, NRX_E_ARG, "nrx_count_errors: NULL buffer");
  NRX_REQUIRE((tb_out == nullptr) == (tb_ref == nullptr), NRX_E_ARG, "nrx_count_errors: tb_out and tb_ref go together");
  NRX_REQUIRE(!tb_out || tb_out_stride >= A, NRX_E_SHAPE, "nrx_count_errors: stride < A");
  const int rows = (tb_out && n_tb > 0) ? n_tb : 1;
  long per_row = tb_out ? ((long)A + 7) / 8 : 0;
  if (per_row < n_ok) per_row = n_ok;
  int gx = nrx::stream_grid(per_row, 256);
  if (gx > 16) gx = 16;
  hipLaunchKernelGGL(count_errors_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, cb_ok, n_ok, tb_out,
                     tb_ref, n_tb, A, tb_out_stride, (unsigned long long*)counters);
  NRX_CHECK_LAUNCH("nrx_count_errors");
  return NRX_OK;
}
