/* nrx.h -- C ABI of libnrx.so: the MI355X (gfx950) PDSCH link-level hot path.
 *
 * The reference (InterDigitalInc/NeoRadium v0.4.0) is pure Python/NumPy and has NO plugin/FFI boundary; the
 * drop-in boundary is its Python class surface (neoradium/__init__.py lines 4-21).  This header is the C ABI that
 * sits directly beneath that surface: one entry point per reference method on the hot path, each citing the
 * reference lines it replaces.  `neoradium_amd/*.py` binds it with ctypes (see INTEGRATION.md for the stub a
 * reference maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no C++/torch types.
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()), row-major,
 *     densely packed unless a stride argument says otherwise.
 *   - bits are one uint8 (0/1) per bit, like the reference's int8 arrays.
 *   - `_f32` / `_f64` suffix = the floating type of LLR / sample buffers of that entry.
 *     complex buffers are interleaved (re,im) pairs of that type.
 *   - every entry returns 0 (NRX_OK) or a negative NRX_E_* code, never throws, never allocates or frees
 *     caller-visible memory, never synchronises; work is enqueued on `stream` (a hipStream_t, NULL = default).
 *   - a batch is `n_tb` transport blocks of identical configuration (Monte-Carlo slots); code blocks of the
 *     batch are laid out (n_tb*C, ...) contiguously.
 */
#ifndef NRX_H_
#define NRX_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRX_OK 0
#define NRX_E_ARG (-1)         /* NULL pointer / out-of-range scalar            */
#define NRX_E_SHAPE (-2)       /* sizes inconsistent with each other            */
#define NRX_E_UNSUPPORTED (-3) /* valid 5G parameter this build does not handle */
#define NRX_E_HIP (-4)         /* HIP runtime reported a launch error           */

/* CRC polynomial ids -- reference chancodebase.py:37-44 ('6','11','16','24A','24B','24C') */
#define NRX_CRC6 0
#define NRX_CRC11 1
#define NRX_CRC16 2
#define NRX_CRC24A 3
#define NRX_CRC24B 4
#define NRX_CRC24C 5

/* Library version / last error text (thread-local). */
int32_t nrx_version(void);
int32_t nrx_last_error(char* buf, int32_t buf_len);

/* ------------------------------------------------------------------------------------------------ CRC
 * chancodebase.py:83-128 getCrc (and :132-157 checkCrc, :161-189 appendCrc on top of it).
 * bits: n_rows rows of row_len bits, consecutive rows row_stride bytes apart.  crc_out: n_rows x L bits. */
int32_t nrx_crc(const uint8_t* bits, int32_t n_rows, int64_t row_len, int64_t row_stride, int32_t poly_id,
                uint8_t* crc_out, void* stream);

/* ------------------------------------------------------------------------------------------------ LDPC
 * Derived sizes of one transport block -- ldpc.py:859-892 initialize, :1014-1026 doSegmentation. */
typedef struct nrx_ldpc_cfg {
  int32_t bg;     /* base graph 1|2                                        */
  int32_t B;      /* TB size incl. 24-bit TB CRC                           */
  int32_t C;      /* code blocks                                            */
  int32_t Zc;     /* lifting size                                           */
  int32_t iLS;    /* lifting set index 0..7                                 */
  int32_t K;      /* 22Zc | 10Zc                                            */
  int32_t N;      /* 66Zc | 50Zc (punctured coded length)                   */
  int32_t F;      /* filler bits per code block                             */
  int32_t cb_len; /* payload bits per code block incl. CB CRC (= K - F)     */
} nrx_ldpc_cfg;

/* ldpc.py:859-892: fill `cfg` for a TB of B bits (B includes the TB CRC).  Host-only, no GPU work. */
int32_t nrx_ldpc_config(int32_t bg, int32_t B, nrx_ldpc_cfg* cfg);

/* ldpc.py:846-856 getRateMatchedCbLens: E_r for r in [0,C).  Host-only. */
int32_t nrx_ldpc_cb_lens(int32_t G, int32_t C, int32_t nl, int32_t qm, int32_t* e_out);

/* chancodebase.py:161-189 appendCrc('24A') + ldpc.py:981-1030 doSegmentation.
 * tb: n_tb x A bits.  add_tb_crc!=0: cfg->B == A+24 and CRC24A is attached; else cfg->B == A.
 * cbs out: (n_tb*C) x K bits (CB CRC24B when C>1, F zero filler bits). */
int32_t nrx_ldpc_segment(const uint8_t* tb, int32_t n_tb, int32_t A, int32_t add_tb_crc, const nrx_ldpc_cfg* cfg,
                         uint8_t* cbs, void* stream);

/* ldpc.py:1033-1090 encode.  cbs: n_cb x K.  coded: n_cb x N (puncture!=0) or n_cb x (N+2Zc). */
int32_t nrx_ldpc_encode(const uint8_t* cbs, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t puncture,
                        uint8_t* coded, void* stream);

/* ldpc.py:1093-1159 rateMatch (bit selection from the filler-free circular buffer at k0(rv), bit interleave,
 * code-block concatenation).  coded: (n_tb*C) x N.  out: n_tb x G', G' = ceil(G/(nl*qm))*nl*qm = sum of E_r
 * (= G whenever G is a multiple of nl*qm, as for every PDSCH allocation). */
int32_t nrx_ldpc_rate_match(const uint8_t* coded, int32_t n_tb, const nrx_ldpc_cfg* cfg, int32_t G, int32_t nl,
                            int32_t qm, int32_t rv, int32_t n_ref, uint8_t* out, void* stream);

/* ldpc.py:1330-1418 recoverRate.  llr: n_tb x llr_len (llr_len <= G... G = llr_len defines E_r as in the
 * reference).  circ: (n_tb*C) x (Ncb-F) HARQ soft buffer, accumulated IN PLACE (NULL = no HARQ state, start
 * from zeros).  out: (n_tb*C) x N with fillers = 1e20. */
int32_t nrx_ldpc_rate_recover_f32(const float* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                  int32_t nl, int32_t qm, int32_t rv, int32_t n_ref, float* circ, float* out,
                                  void* stream);
int32_t nrx_ldpc_rate_recover_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                  int32_t nl, int32_t qm, int32_t rv, int32_t n_ref, double* circ, double* out,
                                  void* stream);

/* ldpc.py:1495-1581 decode: layered normalised min-sum, fixed n_iter.
 * llr: n_cb x N.  hard_out (nullable): n_cb x out_cols bits (r<0).  belief_out (nullable): n_cb x out_cols.
 * out_cols = K (onlyInfoBits) or N+2Zc (all columns incl. the two punctured ones).
 * _f64 is bit-exact with the reference's float64 arithmetic (same operation order, no FMA contraction) and
 * needs a workspace of nrx_ldpc_decode_ws_bytes(); _f32 is the single-precision throughput variant
 * (workspace may be NULL). */
size_t nrx_ldpc_decode_ws_bytes(const nrx_ldpc_cfg* cfg, int32_t is_f64);
int32_t nrx_ldpc_decode_f32(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                            int32_t out_cols, uint8_t* hard_out, float* belief_out, void* ws, size_t ws_bytes,
                            void* stream);
int32_t nrx_ldpc_decode_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter,
                            int32_t out_cols, uint8_t* hard_out, double* belief_out, void* ws, size_t ws_bytes,
                            void* stream);

/* ldpc.py:1584-1619 checkCrcAndMerge (+ the TB-level checkCrc('24A') the harness applies).
 * dec: (n_tb*C) x K hard bits.  tb_out (nullable): n_tb x B bits (TB incl. its CRC24A).
 * cb_ok: n_tb x C (C>1: CRC24B per block; C==1: the TB CRC24A).  tb_ok (nullable): n_tb (CRC24A of tb_out). */
int32_t nrx_ldpc_crc_merge(const uint8_t* dec, int32_t n_tb, const nrx_ldpc_cfg* cfg, uint8_t* tb_out,
                           uint8_t* cb_ok, uint8_t* tb_ok, void* stream);

/* Harness counters (PDSCH-BLER.ipynb cell 2): counters[0] += #(cb_ok==0), [1] += n_ok entries,
 * [2] += #(tb_out[:, :A] != tb_ref), [3] += n_tb*A.  counters: int64[4] on device, accumulated atomically. */
int32_t nrx_count_errors(const uint8_t* cb_ok, int32_t n_ok, const uint8_t* tb_out, const uint8_t* tb_ref,
                         int32_t n_tb, int32_t A, int32_t tb_out_stride, int64_t* counters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NRX_H_ */
