// CRC generator polynomials shared by the LDPC and polar translation units.
#pragma once
#include "nrx_common.h"

namespace nrx {

// chancodebase.py:37-44 -- generator polynomials, MSB first incl. the leading one.
__host__ __device__ inline uint32_t crc_poly(int id) {
  switch (id) {
    case NRX_CRC6: return 0x61u;
    case NRX_CRC11: return 0xE21u;
    case NRX_CRC16: return 0x11021u;
    case NRX_CRC24A: return 0x1864CFBu;
    case NRX_CRC24B: return 0x1800063u;
    default: return 0x1B2B117u;  // 24C
  }
}
__host__ __device__ inline int crc_len(int id) {
  switch (id) {
    case NRX_CRC6: return 6;
    case NRX_CRC11: return 11;
    case NRX_CRC16: return 16;
    default: return 24;
  }
}

}  // namespace nrx
