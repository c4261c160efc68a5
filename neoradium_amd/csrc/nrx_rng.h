// Counter-based generator of the throughput mode (gfx950): Philox4x32-10 keyed by (seed), counter = (element index,
// item, stream id); Box-Muller in float64.  Results do not depend on grid size, batch split or GPU count.
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

// x + complex normal noise of standard deviation sigma for element e of batch item `item` (random.py:203 awgn =
// normal(0, sigma/sqrt(2)) per component)
template <typename T>
__device__ __forceinline__ cx<T> awgn_add(cx<T> v, double sigma, uint64_t seed, uint64_t stream_id, uint64_t item,
                                          int64_t e) {
  uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)item, (uint32_t)(item >> 32) ^ (uint32_t)stream_id};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  // two 53-bit-ish uniforms in (0,1]
  const double u1 = ((double)(((uint64_t)c[0] << 21) ^ (c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(((uint64_t)c[2] << 21) ^ (c[3] >> 11))) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  const double s = sigma / 1.4142135623730951;
  return cx<T>((T)((double)v.re + s * rad * cs), (T)((double)v.im + s * rad * sn));
}

// The noise term of awgn_add on its own (what is added to the sample): lets a kernel compute it while the sample's load is
// still in flight.  v + awgn_noise(...) (component-wise, in double, then rounded to T) IS awgn_add(v, ...).
__device__ __forceinline__ cx<double> awgn_noise(double sigma, uint64_t seed, uint64_t stream_id, uint64_t item, int64_t e) {
  uint32_t c[4] = {(uint32_t)e, (uint32_t)((uint64_t)e >> 32), (uint32_t)item, (uint32_t)(item >> 32) ^ (uint32_t)stream_id};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)(((uint64_t)c[0] << 21) ^ (c[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(((uint64_t)c[2] << 21) ^ (c[3] >> 11))) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  const double s = sigma / 1.4142135623730951;
  return cx<double>(s * rad * cs, s * rad * sn);
}

}  // namespace nrx
