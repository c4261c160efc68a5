// Shared helpers for the nrx HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nrx.h"

namespace nrx {

void set_error(const char* fmt, ...);

// Every C-ABI entry: validate -> launch -> check the launch (never synchronises).
#define NRX_REQUIRE(cond, code, ...)                                     \
  do {                                                                   \
    if (!(cond)) {                                                       \
      ::nrx::set_error(__VA_ARGS__);                                     \
      return (code);                                                     \
    }                                                                    \
  } while (0)

#define NRX_CHECK_LAUNCH(name)                                           \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      ::nrx::set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__)); \
      return NRX_E_HIP;                                                  \
    }                                                                    \
  } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Grid size for a streaming (HBM-bound) kernel: enough workgroups to fill 256 CUs several times over,
// grid-stride the rest (cdna_hip_programming.md Guideline 11).
static inline int stream_grid(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 256 * 16) g = 256 * 16;
  return (int)g;
}

}  // namespace nrx
