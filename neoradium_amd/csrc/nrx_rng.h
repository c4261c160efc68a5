// Counter-based generator of the throughput mode (gfx950): Philox4x32-10 keyed by (seed), counter = (element index,
// item, stream id); Box-Muller (float32 transcendentals by default, float64 on request: see normal_pair).  Results do not depend on grid size, batch split or GPU count.
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

// The standard complex normal pair of element e of item `item`: Box-Muller on the uniforms of ONE Philox block.
//   F64N = false (default of the throughput mode): u1 is the 53-bit uniform formed in float64, u2 a 24-bit uniform; the transform
//     itself -- log2, square root, sine and cosine of a revolution -- runs on the float32 transcendental unit (v_log_f32, v_sqrt_f32,
//     v_sin_f32, v_cos_f32: one instruction each, where the float64 library calls were most of the demodulator's time).  The normals
//     carry float32 precision in a float64 container: the angle has 24 bits, v_sin_f32 / v_cos_f32 have an ABSOLUTE error of about
//     1e-7 (not a relative one), and (float)u1 quantises the radius near u1 -> 1 to steps of about 3e-4 sigma; c[3] is unused.
//   F64N = true (nrx_set_noise_precision(1) / NRX_RNG_F64=1): both uniforms 53-bit, log / sqrt / sincospi in float64 -- float64 normals
//     like the reference's (random.py:203 draws float64 normals); what rounds 1-4 used, and what BLER-regression runs against those
//     rounds' counters need.
// Either way this is the throughput mode's SYNTHETIC input, defined by this function alone (parity mode takes the host's PCG64 draws
// as data, random.py:203); its distribution is what tests/test_gpu_phy.py::test_noise_level_and_noise checks for both settings.
// u1 in (0, 1]: the radius reaches sqrt(-2 ln 2^-53) = 8.6 sigma (small u1 keep their full precision as float32).
template <bool F64N>
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t stream_id, uint64_t item, int64_t e, double& zr, double& zi) {
  uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)item, (uint32_t)(item >> 32) ^ (uint32_t)stream_id};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)(((uint64_t)c[0] << 21) ^ (c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
  if constexpr (F64N) {
    const double u2 = ((double)(((uint64_t)c[2] << 21) ^ (c[3] >> 11))) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    zr = rad * cs;
    zi = rad * sn;
  } else {
    const float u2 = (float)(c[2] >> 8) * (1.0f / 16777216.0f);           // 24 bits: exact in float32, in [0, 1)
    const float l2 = __builtin_amdgcn_logf((float)u1);                    // log2 u1 <= 0
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * l2);  // sqrt(-2 ln u1)
    zr = (double)(rad * __builtin_amdgcn_cosf(u2));                       // v_cos_f32 / v_sin_f32 take revolutions
    zi = (double)(rad * __builtin_amdgcn_sinf(u2));
  }
}

// x + complex normal noise of standard deviation sigma for element e of batch item `item` (random.py:203 awgn =
// normal(0, sigma/sqrt(2)) per component)
template <typename T, bool F64N>
__device__ __forceinline__ cx<T> awgn_add(cx<T> v, double sigma, uint64_t seed, uint64_t stream_id, uint64_t item,
                                          int64_t e) {
  double zr, zi;
  normal_pair<F64N>(seed, stream_id, item, e, zr, zi);
  const double s = sigma / 1.4142135623730951;
  return cx<T>((T)((double)v.re + s * zr), (T)((double)v.im + s * zi));
}

// The noise term of awgn_add on its own (what is added to the sample): lets a kernel compute it while the sample's load is
// still in flight.  v + awgn_noise(...) (component-wise, in double, then rounded to T) IS awgn_add(v, ...).
template <bool F64N>
__device__ __forceinline__ cx<double> awgn_noise(double sigma, uint64_t seed, uint64_t stream_id, uint64_t item, int64_t e) {
  double zr, zi;
  normal_pair<F64N>(seed, stream_id, item, e, zr, zi);
  const double s = sigma / 1.4142135623730951;
  return cx<double>(s * zr, s * zi);
}

// host side: which transform the next launches use (nrx_api.hip; default from NRX_RNG_F64, changed by nrx_set_noise_precision)
bool noise_f64();

}  // namespace nrx
