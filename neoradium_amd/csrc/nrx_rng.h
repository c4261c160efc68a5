// Counter-based generator of the throughput mode (gfx950): Philox4x32-10 keyed by (seed), counter = (element index,
// item, stream id); Box-Muller (float32 transcendentals, see normal_pair).  Results do not depend on grid size, batch split or GPU count.
#pragma once
#include "nrx_cplx.h"

namespace nrx {

__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// The standard complex normal pair of element e of item `item`: Box-Muller on the two uniforms of ONE Philox block.  The 53-bit
// uniforms are formed in float64; the transform itself -- log2, square root, sine and cosine of a revolution -- runs on the float32
// transcendental unit (v_log_f32, v_sqrt_f32, v_sin_f32, v_cos_f32: one instruction each, where the float64 library calls were most of
// the demodulator's time): the normals carry float32 precision (relative 1e-7) in a float64 container.  This is the throughput
// mode's SYNTHETIC input, defined by this function alone (parity mode takes the host's PCG64 draws as data, random.py:203); its
// distribution is what tests/test_gpu_phy.py::test_noise_level_and_noise checks (moments, tails, independence of the batch split).
// u1 in (0, 1]: the radius reaches sqrt(-2 ln 2^-53) = 8.6 sigma (small u1 keep their full precision as float32).
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t stream_id, uint64_t item, int64_t e, double& zr, double& zi) {
  uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)item, (uint32_t)(item >> 32) ^ (uint32_t)stream_id};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)(((uint64_t)c[0] << 21) ^ (c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
  const float u2 = (float)(c[2] >> 8) * (1.0f / 16777216.0f);           // 24 bits: exact in float32, in [0, 1)
  const float l2 = __builtin_amdgcn_logf((float)u1);                    // log2 u1 <= 0
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2);  // sqrt(-2 ln u1)
  zr = (double)(rad * __builtin_amdgcn_cosf(u2));                       // v_cos_f32 / v_sin_f32 take revolutions
  zi = (double)(rad * __builtin_amdgcn_sinf(u2));
}

// x + complex normal noise of standard deviation sigma for element e of batch item `item` (random.py:203 awgn =
// normal(0, sigma/sqrt(2)) per component)
template <typename T>
__device__ __forceinline__ cx<T> awgn_add(cx<T> v, double sigma, uint64_t seed, uint64_t stream_id, uint64_t item,
                                          int64_t e) {
  double zr, zi;
  normal_pair(seed, stream_id, item, e, zr, zi);
  const double s = sigma / 1.4142135623730951;
  return cx<T>((T)((double)v.re + s * zr), (T)((double)v.im + s * zi));
}

// The noise term of awgn_add on its own (what is added to the sample): lets a kernel compute it while the sample's load is
// still in flight.  v + awgn_noise(...) (component-wise, in double, then rounded to T) IS awgn_add(v, ...).
__device__ __forceinline__ cx<double> awgn_noise(double sigma, uint64_t seed, uint64_t stream_id, uint64_t item, int64_t e) {
  double zr, zi;
  normal_pair(seed, stream_id, item, e, zr, zi);
  const double s = sigma / 1.4142135623730951;
  return cx<double>(s * zr, s * zi);
}

}  // namespace nrx
