// Host-only entry points of libnrx.so: version, error text, LDPC size derivation.
#include <stdarg.h>
#include <math.h>
#include "nrx_common.h"

namespace nrx {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace nrx

extern "C" int32_t nrx_version(void) { return 100; }  // 0.1.0

// Precision of the device noise generator's Box-Muller transform (nrx_rng.h): 0 = float32 transcendental unit (default), 1 = float64
// (random.py:203 draws float64 normals).  Process-wide; read by the launches that follow.
namespace nrx {
static int g_noise_f64 = -1;
bool noise_f64() {
  if (g_noise_f64 < 0) {
    const char* e = getenv("NRX_RNG_F64");
    g_noise_f64 = (e && e[0] && e[0] != '0') ? 1 : 0;
  }
  return g_noise_f64 == 1;
}
}  // namespace nrx
extern "C" int32_t nrx_set_noise_precision(int32_t f64) {
  NRX_REQUIRE(f64 == 0 || f64 == 1, NRX_E_ARG, "nrx_set_noise_precision: 0 (float32 transform) or 1 (float64 transform)");
  nrx::g_noise_f64 = f64;
  return NRX_OK;
}
extern "C" int32_t nrx_get_noise_precision(void) { return nrx::noise_f64() ? 1 : 0; }

extern "C" int32_t nrx_last_error(char* buf, int32_t buf_len) {
  if (!buf || buf_len <= 0) return NRX_E_ARG;
  strncpy(buf, nrx::g_err, (size_t)buf_len - 1);
  buf[buf_len - 1] = 0;
  return NRX_OK;
}

// Lifting sizes a * 2^j <= 384, a in {2,3,5,7,9,11,13,15}  (TS 38.212 Table 5.3.2-1; reference ldpc.py:657-666).
extern "C" int32_t nrx_ldpc_config(int32_t bg, int32_t B, nrx_ldpc_cfg* cfg) {
  NRX_REQUIRE(cfg, NRX_E_ARG, "nrx_ldpc_config: NULL cfg");
  NRX_REQUIRE(bg == 1 || bg == 2, NRX_E_ARG, "nrx_ldpc_config: bg must be 1|2 (got %d)", bg);
  NRX_REQUIRE(B > 0, NRX_E_ARG, "nrx_ldpc_config: B must be positive (got %d)", B);
  const int kcb = bg == 1 ? 8448 : 3840;
  int C = 1;
  long tot = B;
  if (B > kcb) {  // ldpc.py:864-871
    C = (B + (kcb - 24) - 1) / (kcb - 24);
    tot = (long)B + (long)C * 24;
  }
  int kb;  // ldpc.py:875-879 (BG2 thresholds are keyed on B)
  if (bg == 1) kb = 22;
  else if (B > 640) kb = 10;
  else if (B > 560) kb = 9;
  else if (B > 192) kb = 8;
  else kb = 6;
  // smallest Z over all sets with kb*Z >= K' = tot/C (K' may be fractional): kb*Z*C >= tot
  static const int base[8] = {2, 3, 5, 7, 9, 11, 13, 15};
  int best = 10000, ils = -1;
  for (int i = 0; i < 8; ++i)
    for (int z = base[i]; z <= 384; z *= 2)
      if ((long)kb * z * C >= tot && z < best) {
        best = z;
        ils = i;
      }
  NRX_REQUIRE(ils >= 0, NRX_E_UNSUPPORTED, "nrx_ldpc_config: no lifting size for B=%d", B);
  cfg->bg = bg;
  cfg->B = B;
  cfg->C = C;
  cfg->Zc = best;
  cfg->iLS = ils;
  cfg->K = (bg == 1 ? 22 : 10) * best;
  cfg->N = (bg == 1 ? 66 : 50) * best;
  cfg->cb_len = (B + C - 1) / C + (C > 1 ? 24 : 0);  // ldpc.py:1014,1367-1368
  cfg->F = cfg->K - cfg->cb_len;
  NRX_REQUIRE(cfg->F >= 0, NRX_E_UNSUPPORTED, "nrx_ldpc_config: negative filler count");
  return NRX_OK;
}

extern "C" int32_t nrx_ldpc_cb_lens(int32_t G, int32_t C, int32_t nl, int32_t qm, int32_t* e_out) {
  NRX_REQUIRE(e_out && C > 0 && nl > 0 && qm > 0 && G >= 0, NRX_E_ARG, "nrx_ldpc_cb_lens: bad argument");
  const int f = nl * qm;
  const int gb = (G + f - 1) / f;  // ldpc.py:852
  for (int r = 0; r < C; ++r) e_out[r] = (gb / C) * f + ((gb % C) && r >= C - gb % C ? f : 0);
  return NRX_OK;
}

// TS 38.211 5.2.1 length-31 Gold sequence c(n), Nc = 1600 (reference utils.py:70-94 goldSequence).  Host-only:
// scrambling / DMRS sequences are per-configuration constants that the wrappers upload once.
extern "C" int32_t nrx_gold_sequence(uint32_t c_init, int64_t n, uint8_t* out_host) {
  NRX_REQUIRE(out_host && n >= 0, NRX_E_ARG, "nrx_gold_sequence: bad argument");
  uint32_t x1 = 1u, x2 = c_init & 0x7FFFFFFFu;   // bit i of the word = x(n+i)
  auto step1 = [](uint32_t x) { return (x >> 1) | ((((x >> 3) ^ x) & 1u) << 30); };
  auto step2 = [](uint32_t x) { return (x >> 1) | ((((x >> 3) ^ (x >> 2) ^ (x >> 1) ^ x) & 1u) << 30); };
  for (int i = 0; i < 1600; ++i) { x1 = step1(x1); x2 = step2(x2); }
  for (int64_t i = 0; i < n; ++i) {
    out_host[i] = (uint8_t)((x1 ^ x2) & 1u);
    x1 = step1(x1);
    x2 = step2(x2);
  }
  return NRX_OK;
}

// Developer hook for bench.py's roofline: the shader clock under a float64 load on every CU, read on the device -- s_memtime (shader
// clock counter) against s_memrealtime (constant 100 MHz) around a spin of dependent v_fma_f64 (three waves per SIMD, like the
// decoder).  out2[0] += s_memtime ticks, out2[1] += s_memrealtime ticks of wave 0 of workgroup 0; clock = 1e8 * out2[0] / out2[1].
namespace {
__global__ void __launch_bounds__(768, 1) clock_probe_kernel(unsigned long long* out2, int spin, double* sink) {
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-12;
  for (int i = 0; i < spin; ++i) {
    a = __builtin_fma(a, b, c);
    a = __builtin_fma(a, b, c);
    a = __builtin_fma(a, b, c);
    a = __builtin_fma(a, b, c);
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  if (a == 12345.678) sink[0] = a;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out2[0] = t1 - t0;
    out2[1] = r1 - r0;
  }
}
}  // namespace
extern "C" int32_t nrx_debug_clock_probe(unsigned long long* out2_dev, double* sink_dev, int32_t spin, void* stream) {
  NRX_REQUIRE(out2_dev && sink_dev && spin > 0, NRX_E_ARG, "nrx_debug_clock_probe: bad argument");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(256), dim3(768), 0, (hipStream_t)stream, out2_dev, spin, sink_dev);
  NRX_CHECK_LAUNCH("nrx_debug_clock_probe");
  return NRX_OK;
}
