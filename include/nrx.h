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

/* ldpc.py:1033-1090 encode.  cbs: n_cb x K.  coded: n_cb x N (puncture!=0) or n_cb x (N+2Zc).
 * n_rows: 0 = all; otherwise only the parity of the first n_rows (>= 4) base-graph rows is computed and written -- the
 * columns from 22 + n_rows (BG1) / 10 + n_rows (BG2) on stay untouched: for a caller whose rate matching (rv 0) ends before them. */
int32_t nrx_ldpc_encode(const uint8_t* cbs, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t puncture, int32_t n_rows,
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

/* HARQ batches (harq.py:145-202 HarqCW.getRateMatchedCodeBlocks / decodeLLRs for many HARQ processes at once): the
 * same two operations with one redundancy version per transport block, rv_per_tb[n_tb] (device, values 0..3), and,
 * for rate recovery, reset_per_tb[n_tb] (device, may be NULL): non-zero = this process starts a new transport block,
 * its soft buffer is restarted from zero instead of being accumulated into.  The soft buffers stay resident in HBM. */
int32_t nrx_ldpc_rate_match_harq(const uint8_t* coded, int32_t n_tb, const nrx_ldpc_cfg* cfg, int32_t G, int32_t nl,
                                 int32_t qm, const int32_t* rv_per_tb, int32_t n_ref, uint8_t* out, void* stream);
int32_t nrx_ldpc_rate_recover_harq_f32(const float* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                       int32_t nl, int32_t qm, const int32_t* rv_per_tb, const uint8_t* reset_per_tb,
                                       int32_t n_ref, float* circ, float* out, void* stream);
int32_t nrx_ldpc_rate_recover_harq_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                       int32_t nl, int32_t qm, const int32_t* rv_per_tb, const uint8_t* reset_per_tb,
                                       int32_t n_ref, double* circ, double* out, void* stream);

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

/* The same decoder restricted to the first n_rows rows of the base graph (4 <= n_rows <= 46 | 42), hard decisions of
 * the K information bits.  PRECONDITION (caller): the extension (degree-1 parity) column of every dropped row carries
 * all-zero LLRs in every code block -- i.e. that parity was punctured by rate matching and never received (first
 * transmission: rows >= ceil((F + E_max) / Zc) - 20 for E_max > K - 2Zc - F; ldpc.py:1093-1159, 1330-1418).  Such a row
 * has min1 = 0 at its extension edge, so ldpc.py:1556-1576 sends +-0 to all its other columns: the posteriors of the
 * core columns -- and with them these outputs -- are bit-identical to running all rows, in float32 and float64
 * (tests/test_gpu_ldpc.py::test_punctured_rows_are_exact_no_ops).  NR LDPC is built for exactly this (a raptor-like
 * extension: each higher-rate code is the sub-matrix of the transmitted parity), the reference just does not use it. */
int32_t nrx_ldpc_decode_rows_f32(const float* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                 uint8_t* hard_out, void* ws, size_t ws_bytes, void* stream);
int32_t nrx_ldpc_decode_rows_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                 uint8_t* hard_out, void* ws, size_t ws_bytes, void* stream);
/* ... of a selection of the code blocks: sel[0 .. *n_sel) index the n_cb rows of llr / hard_out, list and count on the DEVICE
 * (the launch covers the worst case; nothing is read by the host, e.g. nrx_select_failed of a flag vector); the other rows of
 * hard_out are left untouched.  NRX_E_UNSUPPORTED outside BG1 / Zc 384 (the kernels of nrx_ldpc_dec3.hip).  Used by the batched
 * HARQ loop: new transport blocks (rv 0 alone in the buffer: <= 15 rows) and retransmissions (all rows) of one round in two
 * launches over the same buffers. */
int32_t nrx_ldpc_decode_rows_sel_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                     uint8_t* hard_out, void* ws, size_t ws_bytes, const int32_t* sel, const int32_t* n_sel,
                                     void* stream);

/* ldpc.py:1330-1418 recoverRate (first transmission: rv 0, no HARQ soft buffer, no LBRM, no wrap-around repetition)
 * + ldpc.py:1495-1581 decode + ldpc.py:1584-1619 checkCrcAndMerge (C > 1: CRC24B per code block) in ONE launch (SURVEY 7
 * step 4): the decoder's initial fill reads the LLRs straight from the demapper output llr (n_tb, llr_len = G = sum E_r)
 * in the per-code-block DE-INTERLEAVED layout nrx_qam_demap_cb_* writes (block r at offset_r, position e = q*(E_r/qm) + s:
 * the buffer order of ldpc.py:1390-1397; contiguous loads), fillers become the clipped LARGE_LLR in-register, and its tail writes the merged hard bits
 * tb_out (n_tb, C*(cb_len-24)) and the per-code-block CRC verdicts cb_ok (n_tb*C); the (n_tb*C, N) rate-recovered LLRs and
 * the (n_tb*C, K) hard-bit matrix never exist in HBM.  Results are bit-identical to nrx_ldpc_rate_recover_f64 (on the
 * symbol-major LLRs of nrx_qam_demap_*) -> nrx_ldpc_decode_rows_f64 -> nrx_ldpc_crc_merge.  n_rows: rows of the base graph to run (0 = as many as the received
 * bits reach; raised to that number when smaller).  Returns NRX_E_UNSUPPORTED when (bg, Zc, C, rows) has no fused
 * instantiation (today: BG1, Zc 384, C > 1, <= 15 rows): the caller then uses the three separate entries. */
int32_t nrx_ldpc_recover_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                          int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                          uint8_t* cb_ok, void* stream);
/* The same for a selection of the code blocks: sel[0 .. *n_sel) index the n_tb * C blocks, list and count on the DEVICE (the
 * launch covers the worst case, no host read); only the selected blocks' payload bits in tb_out and their cb_ok entries are
 * written.  nrx_select_failed gives the list of the blocks whose cb_ok is 0 (ascending).  Together: the opt-in two-pass
 * schedule -- every block with few iterations first, the ones whose CRC24B fails again FROM SCRATCH with all iterations (not
 * the reference's schedule, ldpc.py:1495-1581 runs a fixed count; same bits for every block that passes either way). */
int32_t nrx_ldpc_recover_decode_merge_sel_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                              int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                              uint8_t* cb_ok, const int32_t* sel, const int32_t* n_sel, void* stream);
int32_t nrx_select_failed(const uint8_t* flags, int32_t n, int32_t* sel, int32_t* n_sel, void* stream);
/* The continuation form: nrx_ldpc_recover_decode_merge_park_f64 = nrx_ldpc_recover_decode_merge_f64, and every code block whose
 * CRC24B fails leaves its complete decoder state (posterior columns, check-node minima, sign / argmin words) in `state`
 * (n_tb * C slots of nrx_ldpc_fused_state_bytes(...) bytes, indexed by code block; only failing blocks' slots are written).
 * nrx_ldpc_resume_decode_merge_sel_f64 continues the selected blocks from their parked state for n_iter MORE iterations:
 * park(n1) then resume(n2) computes for those blocks exactly what one run of n1 + n2 iterations computes (ldpc.py:1495-1581
 * with numIter = n1 + n2), so the second pass of the two-pass schedule does not repeat the first pass's iterations. */
int64_t nrx_ldpc_fused_state_bytes(const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm, int32_t llr_len, int32_t n_rows);
int32_t nrx_ldpc_recover_decode_merge_park_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg,
                                               int32_t nl, int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out,
                                               uint8_t* cb_ok, void* state, void* stream);
int32_t nrx_ldpc_resume_decode_merge_sel_f64(int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm,
                                             int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                             const int32_t* sel, const int32_t* n_sel, void* state, int32_t park_again,
                                             void* stream);      /* park_again != 0: blocks that still fail park their state once more */

/* ---- Certified early exit (opt-in schedule; never the headline measurement).  The reference runs a fixed number of iterations
 * (ldpc.py:1545-1576) and has no early stop.  These entries stop a code block early ONLY where a certificate evaluated on its
 * frozen decoder state proves that every later iteration of that same float64 recursion leaves every hard decision unchanged
 * (statement and proof; oracle/certificate.py: the CPU restatement) -- the bits of a certified block ARE the
 * bits ldpc.py:1578-1581 returns after numIter iterations.
 *
 * nrx_ldpc_cert_bounds (host only): a-priori magnitude bounds of ldpc.py:1546-1576 on the first n_rows rows, per unit of the
 * LLR maxima: out3[0] bounds every |r| (prices the rounding-error budget), out3[1] the smallest |t| of a core row in units of
 * the largest |LLR| over the core-parity + extension columns (prices the +1e5 quirk of ldpc.py:1563), out3[2] = the largest
 * column degree.  Infinity = no bound (no certificate).
 * nrx_ldpc_stage_decode_merge_f64: one stage = nrx_ldpc_recover_decode_merge_f64 for n_iter iterations (sel == NULL: from the
 * LLRs, also leaving max|LLR| over the received non-filler positions and over the parity + extension columns in lam[2 cb],
 * lam[2 cb + 1]) or the continuation of the selected blocks from their parked state; EVERY block that ran parks its state.
 * nrx_ldpc_certify_f64: the certificate on the parked state of the selected blocks (all when sel == NULL) whose cb_ok is 1;
 * exit_iter[cb] = min(iter_now, 255) where it holds, untouched elsewhere.  n_iter_total = the reference's numIter (the horizon
 * of the error budget); max_sweeps = relaxation sweeps before the certificate refuses (clamped to 16: the float32 slack sums'
 * conservative inflation, nrx_ldpc_certcore.h, is priced for that many); flags -- deliberately BROKEN certificates for the tests:
 * bit 0 skips the sign / posterior / magnitude conditions (S), (Q), bit 1 the closure condition (M), bit 2 also tries blocks whose
 * CRC fails (all three honoured by the three kernels). */
int32_t nrx_ldpc_cert_bounds(const nrx_ldpc_cfg* cfg, int32_t n_rows, double* out3);
int32_t nrx_ldpc_stage_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                        int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                        const int32_t* sel, const int32_t* n_sel, void* state, double* lam, void* stream);
/* The same stage with the certificate evaluated INSIDE the kernel's tail, on the state the workgroup still holds (posterior columns in
 * LDS, check-node state in registers; only the slack sums and the rows' slacks go through `scratch`, 2 * 56 * 384 floats per workgroup of the
 * launch, which takes as many workgroups as fit): a block whose CRC passes and whose state holds the certificate gets
 * exit_iter[cb] = min(iter_now, 255) and does NOT park; every other block that ran parks.  No nrx_ldpc_certify_f64 launch is needed. */
int32_t nrx_ldpc_stage_certify_decode_merge_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                                int32_t qm, int32_t n_iter, int32_t n_rows, uint8_t* tb_out, uint8_t* cb_ok,
                                                const int32_t* sel, const int32_t* n_sel, void* state, double* lam, uint8_t* exit_iter,
                                                void* scratch, size_t scratch_bytes, int32_t iter_now, int32_t n_iter_total,
                                                int32_t max_sweeps, int32_t flags, void* stream);
int32_t nrx_ldpc_certify_f64(const void* state, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl, int32_t qm,
                             int32_t n_rows, const int32_t* sel, const int32_t* n_sel, const uint8_t* cb_ok, const double* lam,
                             int32_t iter_now, int32_t n_iter_total, int32_t max_sweeps, int32_t flags, uint8_t* exit_iter,
                             void* stream);
/* The certified schedule as ONE launch (round 6; ldpc.py:1495-1619 with the early exit of DESIGN 4.3): every code-block slot of the grid
 * (one workgroup of two slots per CU) draws blocks from a device queue and takes each through its stages -- stages[0] iterations from the
 * LLRs, CRC + certificate, stages[1] more, ... the last stage without a certificate -- stopping at the first stage whose CRC passes and
 * whose frozen state holds the stability certificate: exit_iter[cb] = the iterations it had then, 0 = it ran all sum(stages) of them.  The
 * two slots of a workgroup never wait for each other; nothing is parked between stages, nothing reloaded from HBM.  Same bits and verdicts
 * as the staged launches and as the fixed schedule of sum(stages) iterations.  stages: HOST array, 1 <= n_stages <= 4.
 * state: 2 x (CUs of the device) records of nrx_ldpc_fused_state_bytes(...) bytes (a slot's certificate reads its check-node state back
 * from there); lam: 2 n_cb doubles; scratch / scratch_bytes as nrx_ldpc_stage_certify_decode_merge_f64; queue: int32[2] on the device,
 * zeroed by the call -- queue[1] != 0 afterwards means a slot barrier gave up (a bug, never the data): the results are then invalid. */
int32_t nrx_ldpc_certified_persistent_f64(const double* llr, int32_t n_tb, int32_t llr_len, const nrx_ldpc_cfg* cfg, int32_t nl,
                                          int32_t qm, const int32_t* stages, int32_t n_stages, int32_t n_rows, uint8_t* tb_out,
                                          uint8_t* cb_ok, void* state, double* lam, uint8_t* exit_iter, void* scratch,
                                          size_t scratch_bytes, int32_t* queue, int32_t max_sweeps, int32_t flags, void* stream);

/* The certified early exit for ANY configuration (both base graphs, every lifting size, any row count): ldpc.py:1495-1581 decode of
 * rate-recovered LLRs (n_cb, N) -> hard decisions of the K information bits, with the certificate evaluated after the iterations
 * checks[0 .. n_checks) (host array, ascending, <= 8): a block that holds it stops there, exit_iter[cb] = that iteration (0 = it ran
 * all n_iter).  Plain kernel with its state in `ws` (n_cb * nrx_ldpc_decode_certified_ws_bytes bytes) -- the tuned form for the metric
 * configuration is nrx_ldpc_stage_certify_decode_merge_f64. */
int64_t nrx_ldpc_decode_certified_ws_bytes(const nrx_ldpc_cfg* cfg, int32_t n_rows);
int32_t nrx_ldpc_decode_certified_f64(const double* llr, int32_t n_cb, const nrx_ldpc_cfg* cfg, int32_t n_iter, int32_t n_rows,
                                      const int32_t* checks, int32_t n_checks, uint8_t* hard_out, uint8_t* exit_iter, void* ws,
                                      size_t ws_bytes, int32_t max_sweeps, int32_t flags, void* stream);
/* Developer hook (no reference counterpart): the shader clock under a float64 load on every CU -- out2_dev[0] / out2_dev[1] = s_memtime
 * ticks / s_memrealtime (100 MHz) ticks around `spin` x 4 dependent v_fma_f64 per lane; bench.py prices its VALU roofline with it. */
int32_t nrx_debug_clock_probe(unsigned long long* out2_dev, double* sink_dev, int32_t spin, void* stream);
/* Developer hook (no reference counterpart): out16[k] = code blocks nrx_ldpc_certify_f64 certified in relaxation sweep k (k < 15),
 * out16[15] = blocks it refused, since the last reset. */
int32_t nrx_debug_cert_sweeps(unsigned long long* out16, int32_t reset);

/* ldpc.py:1584-1619 checkCrcAndMerge (+ the TB-level checkCrc('24A') the harness applies).
 * dec: (n_tb*C) x K hard bits.  tb_out (nullable): n_tb x M bits, M = C*(cb_len - 24) for C>1 (>= B: the TB incl.
 * its CRC24A followed by the segmentation zero padding, exactly what the reference returns), M = B for C==1.
 * cb_ok: n_tb x C (C>1: CRC24B per block; C==1: the TB CRC24A).  tb_ok (nullable): n_tb (CRC24A of tb_out). */
int32_t nrx_ldpc_crc_merge(const uint8_t* dec, int32_t n_tb, const nrx_ldpc_cfg* cfg, uint8_t* tb_out,
                           uint8_t* cb_ok, uint8_t* tb_ok, void* stream);

/* Harness counters (PDSCH-BLER.ipynb cell 2): counters[0] += #(cb_ok==0), [1] += n_ok entries,
 * [2] += #(tb_out[:, :A] != tb_ref), [3] += n_tb*A.  counters: int64[4] on device, accumulated atomically. */
int32_t nrx_count_errors(const uint8_t* cb_ok, int32_t n_ok, const uint8_t* tb_out, const uint8_t* tb_ref,
                         int32_t n_tb, int32_t A, int32_t tb_out_stride, int64_t* counters, void* stream);

/* utils.py:70-94 goldSequence: c(0..n-1) of TS 38.211 5.2.1 for c_init, written to a HOST buffer (host-only). */
int32_t nrx_gold_sequence(uint32_t c_init, int64_t n, uint8_t* out_host);

/* ------------------------------------------------------------------------------------------- modem / mapping
 * modulation.py:127-156 Modem.modulate fused with pdsch.py:603-608 scrambleBits and the layer/RE scatter of
 * pdsch.py:619-639,855-932 populateGrid.
 * bits: n_batch rows (bits_stride bytes apart) of n_sym*qm bits.  scr (nullable): n_sym*qm scrambling bits.
 * re_index (nullable = identity): destination (complex-element offset inside one batch item of `out`) of
 * symbol i in layer-mapped order.  out: n_batch items of out_stride complex elements. */
int32_t nrx_qam_map_f32(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm,
                        const int32_t* re_index, int32_t n_sym, void* out, int64_t out_stride, int32_t n_batch,
                        void* stream);
int32_t nrx_qam_map_f64(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm,
                        const int32_t* re_index, int32_t n_sym, void* out, int64_t out_stride, int32_t n_batch,
                        void* stream);

/* pdsch.py:670-695 PDSCH.getGrid (DMRS-filled template of the slot) + pdsch.py:855-932 populateGrid in one pass: out
 * (n_batch, elems) is written once per element -- re_inv[e] >= 0: the scrambled + modulated symbol number re_inv[e] of the
 * item's bit stream (the INVERSE of the layer/RE map nrx_qam_map_* scatters through; requires every data RE of the
 * allocation to be covered by this one bit stream, i.e. one codeword); re_inv[e] < 0: templ[templ_sel[b]][e] (DMRS, empty
 * REs; templ (n_templ, elems), templ_sel (n_batch,) int64 = slot number in frame).  Same values as copying the template
 * and calling nrx_qam_map_*.  planes (0 or 1: nothing assumed): a promise about the map, checked by the caller -- the grid is
 * `planes` layers of elems/planes elements and wherever re_inv[e] >= 0 in the first plane, plane p holds symbol
 * re_inv[e] + p at the same position (the layer mapping of TS 38.211 7.3.1.3, pdsch.py:528-551), negative entries
 * coinciding; lets one thread fetch the contiguous bits of all layers of an RE. */
int32_t nrx_pdsch_populate_f32(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_inv,
                               const void* templ, const int64_t* templ_sel, int64_t elems, void* out, int32_t n_batch,
                               int32_t planes, void* stream);
int32_t nrx_pdsch_populate_f64(const uint8_t* bits, int64_t bits_stride, const uint8_t* scr, int32_t qm, const int32_t* re_inv,
                               const void* templ, const int64_t* templ_sel, int64_t elems, void* out, int32_t n_batch,
                               int32_t planes, void* stream);

/* modulation.py:159-204 getLLRsFromSymbols fused with pdsch.py:935-1005 getLLRsFromGrid (gather at re_index,
 * noise floor, descrambling pdsch.py:611-616, llrScale weighting).  exact=0: max-log (useMax=True), 1: log-sum-exp.
 * syms/scales: n_batch items of sym_stride elements (scales nullable, same offsets as syms).
 * noise_var: device scalar(s), element b*nv_stride is used for item b (nv_stride 0 = one value for all),
 * floored at nv_floor (pdsch.py:966 uses 1e-10; pass 0 for the bare Modem call).
 * llr: n_batch rows of llr_stride values, n_sym*qm used.  _f64o32 = complex128 in, float32 LLRs out. */
int32_t nrx_qam_demap_f32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                          int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                          void* llr, int64_t llr_stride, int32_t n_batch, int32_t exact, double nv_floor, void* stream);
int32_t nrx_qam_demap_f64(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                          int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                          void* llr, int64_t llr_stride, int32_t n_batch, int32_t exact, double nv_floor, void* stream);
int32_t nrx_qam_demap_f64o32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                             int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index,
                             int32_t n_sym, void* llr, int64_t llr_stride, int32_t n_batch, int32_t exact,
                             double nv_floor, void* stream);
/* The same max-log demapper with the rate-recovery de-interleaver of ldpc.py:1390-1397 done by its stores: the G = n_sym*qm
 * LLRs of an item are split into n_code_blocks blocks of E_r bits exactly as nrx_ldpc_cb_lens splits them (n_layers*qm
 * granularity, smaller blocks first), and LLR (symbol s, bit q) of block r is written to  offset_r + q*(E_r/qm) + s  -- the
 * position it has in the code block's circular buffer -- instead of offset_r + s*qm + q.  Stores stay coalesced (consecutive
 * symbols = consecutive lanes = consecutive addresses per q).  This is the input layout of nrx_ldpc_recover_decode_merge_f64. */
int32_t nrx_qam_demap_cb_f32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                             int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                             int32_t n_code_blocks, int32_t n_layers, void* llr, int64_t llr_stride, int32_t n_batch,
                             double nv_floor, void* stream);
int32_t nrx_qam_demap_cb_f64(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                             int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                             int32_t n_code_blocks, int32_t n_layers, void* llr, int64_t llr_stride, int32_t n_batch,
                             double nv_floor, void* stream);
int32_t nrx_qam_demap_cb_f64o32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                                int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                                int32_t n_code_blocks, int32_t n_layers, void* llr, int64_t llr_stride, int32_t n_batch,
                                double nv_floor, void* stream);

/* The demapper whose stores do the whole rate recovery of a FIRST transmission (ldpc.py:1330-1418 with rv 0 and no wrap-around
 * repetition, i.e. E_r <= N - F for every block; NRX_E_UNSUPPORTED otherwise): llr = the (n_batch * C, N) buffer
 * nrx_ldpc_rate_recover_* would write from nrx_qam_demap_*'s output -- transmitted positions from the symbols, zeros behind them,
 * LARGE_LLR on the fillers -- for the first n_cols columns (n_cols * Zc positions) of the punctured code word; behind them only
 * the transmitted positions are written, so pass at least the columns the decoder reads (20 + the rows of the instantiation that runs on BG1). */
int32_t nrx_qam_demap_rr_f32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                             int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                             const nrx_ldpc_cfg* cfg, int32_t n_layers, int32_t n_cols, void* llr, int32_t n_batch,
                             double nv_floor, void* stream);
int32_t nrx_qam_demap_rr_f64(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                             int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                             const nrx_ldpc_cfg* cfg, int32_t n_layers, int32_t n_cols, void* llr, int32_t n_batch,
                             double nv_floor, void* stream);
int32_t nrx_qam_demap_rr_f64o32(const void* syms, int64_t sym_stride, const void* scales, const void* noise_var,
                                int32_t nv_stride, const uint8_t* scr, int32_t qm, const int32_t* re_index, int32_t n_sym,
                                const nrx_ldpc_cfg* cfg, int32_t n_layers, int32_t n_cols, void* llr, int32_t n_batch,
                                double nv_floor, void* stream);

/* ------------------------------------------------------------------------------------------------ grid stages
 * grid.py:456-518 Grid.precode (wideband): grid (n_batch,nl,lk) x f (nt,nl; item b at f + b*f_stride elements,
 * f_stride 0 = shared) -> out (n_batch,nt,lk);  lk = symbols*subcarriers. */
int32_t nrx_precode_f32(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk,
                        void* out, int32_t n_batch, void* stream);
int32_t nrx_precode_f64(const void* grid, const void* f, int64_t f_stride, int32_t nl, int32_t nt, int32_t lk,
                        void* out, int32_t n_batch, void* stream);
/* grid.py:978-1018 Grid.applyChannel: grid (n_batch,nt,lk), h (lk,nr,nt per item, h_stride elements apart)
 * -> out (n_batch,nr,lk). */
int32_t nrx_apply_channel_fd_f32(const void* grid, const void* h, int64_t h_stride, int32_t nt, int32_t nr,
                                 int32_t lk, void* out, int32_t n_batch, void* stream);
int32_t nrx_apply_channel_fd_f64(const void* grid, const void* h, int64_t h_stride, int32_t nt, int32_t nr,
                                 int32_t lk, void* out, int32_t n_batch, void* stream);
/* grid.py:626-694 Grid.equalize (MMSE): rx (n_batch,nr,lk), hf (lk,nr,nl per item), noise_var as in demap
 * (floored at 1e-8, grid.py:676) -> eq (n_batch,nl,lk) and llr scale (n_batch,nl,lk) = 1/Re diag((H^H H+s2 I)^-1).
 * Built for nr in {1,2,4,8}, nl in 1..4. */
int32_t nrx_mmse_equalize_f32(const void* rx, const void* hf, int64_t h_stride, const void* noise_var,
                              int32_t nv_stride, int32_t nr, int32_t nl, int32_t lk, void* eq, void* scale,
                              int32_t n_batch, void* stream);
int32_t nrx_mmse_equalize_f64(const void* rx, const void* hf, int64_t h_stride, const void* noise_var,
                              int32_t nv_stride, int32_t nr, int32_t nl, int32_t lk, void* eq, void* scale,
                              int32_t n_batch, void* stream);

/* Noise level of addNoise(snrDb, useRxPower=True): grid.py:1040-1046 (np.var of the grid) and
 * waveform.py:107-142 (np.var of the CP-stripped samples, gathered through `gather`, / (12*numRbs) * nFFT).
 * x: n_batch items (x_stride elements apart) of n_per complex values; gather (nullable): n_gather element
 * offsets to reduce over instead of [0,n_per).  acc_ws: 192*n_batch doubles of scratch (per-workgroup partial sums,
 * reduced in a fixed order: reproducible bit for bit).
 * var_out (nullable): complex variance per item.  If snr_lin != NULL (linear SNR, element b*snr_stride):
 *   sigma_out[b] = sqrt(var*mult/snr)  and  nv_out[b] = sigma^2 * nv_mult   (either may be NULL). */
int32_t nrx_noise_level_f32(const void* x, int64_t n_per, int64_t x_stride, const int32_t* gather, int64_t n_gather,
                            int32_t n_batch, double* acc_ws, void* var_out, const double* snr_lin, int32_t snr_stride,
                            double mult, void* sigma_out, void* nv_out, double nv_mult, void* stream);
int32_t nrx_noise_level_f64(const void* x, int64_t n_per, int64_t x_stride, const int32_t* gather, int64_t n_gather,
                            int32_t n_batch, double* acc_ws, void* var_out, const double* snr_lin, int32_t snr_stride,
                            double mult, void* sigma_out, void* nv_out, double nv_mult, void* stream);
/* random.py:203 awgn + grid.py:1160-1166 / waveform.py:262-266: out = x + (sigma[b]/sqrt 2) * z, z = caller's
 * standard-normal (re,im) pairs (host PCG64 stream in parity mode).  In-place allowed. */
int32_t nrx_add_noise_f32(const void* x, const void* z, const void* sigma, int32_t sigma_stride, int64_t n_per,
                          void* out, int32_t n_batch, void* stream);
int32_t nrx_add_noise_f64(const void* x, const void* z, const void* sigma, int32_t sigma_stride, int64_t n_per,
                          void* out, int32_t n_batch, void* stream);
/* Throughput-mode AWGN: Philox4x32-10 counter RNG keyed by (seed, stream_id, item, element) + Box-Muller; independent of launch
 * geometry, batch split and GPU count.  The transform runs on the float32 transcendental unit by default (normals of float32
 * precision in the caller's type) or in float64 like the reference's normals (random.py:203): nrx_set_noise_precision(1), or
 * NRX_RNG_F64=1 in the environment at the first launch.  Process-wide; every value of the synthetic noise changes with it.  item of batch entry b = item_ids[b] when item_ids (nullable,
 * device int64[n_batch]: e.g. the absolute slot numbers of a non-contiguous slot selection) is given, else batch_offset+b. */
int32_t nrx_set_noise_precision(int32_t f64);
int32_t nrx_get_noise_precision(void);
int32_t nrx_awgn_f32(const void* x, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out,
                     int32_t n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids,
                     void* stream);
int32_t nrx_awgn_f64(const void* x, const void* sigma, int32_t sigma_stride, int64_t n_per, void* out,
                     int32_t n_batch, uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids,
                     void* stream);

/* Synthetic transport blocks (random.py:202 bits) for the throughput mode: n_batch x n_per uniform bits from the
 * same counter-based generator (independent of launch geometry / batch split / GPU count). */
int32_t nrx_random_bits(uint8_t* out, int64_t n_per, int32_t n_batch, uint64_t seed, uint64_t stream_id,
                        int64_t batch_offset, const int64_t* item_ids, void* stream);

/* ------------------------------------------------------------------------------------------------------ OFDM
 * grid.py:521-582 Grid.ofdmModulate (f0=0) + waveform.py:380-470 applyWindowing: grid rows (n_rows = items*ports,
 * each n_sym x K) -> waveform rows of wave_stride samples (slot length = sum(cp)+n_sym*nfft used).
 * cp_lens: HOST array of n_sym CP lengths (up to 112: several consecutive slots of a subframe in one call, grid.py:546-549).  window_len: raised-cosine overlap length (0 = "NONE";
 * "STD" = min over symbols of (cp+1)/2, waveform.py:99-122). */
int32_t nrx_ofdm_modulate_f32(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens,
                              int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* stream);
int32_t nrx_ofdm_modulate_f64(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens,
                              int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* stream);
/* waveform.py:317-341 sync + :473-527 ofdmDemodulate (f0=0: the down-conversion phase of f0 > 0 is one factor per symbol,
 * applied by the caller; cp_offset_ratio = cpOffsetRatio, where in the CP the FFT window starts, 0.5 by default): waveform rows (n_items*n_ant rows
 * of wave_stride samples, wave_len valid) -> grid (n_items,n_ant,n_sym,K).  t_off (nullable, device): timing
 * offset of item b at t_off[b*t_off_stride]. */
/* nrx_awgn_* followed by nrx_ofdm_demodulate_* in one pass (throughput mode: the noisy waveform is never written):
 * wave is the NOISELESS received waveform, rows contiguous (wave_stride == wave_len); sigma[item*sigma_stride] and
 * (seed, stream_id, batch_offset, item_ids) as for nrx_awgn_* over the item's n_ant*wave_len elements.  Identical output. */
int32_t nrx_ofdm_demodulate_awgn_f32(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                     int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                     const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride,
                                     uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* grid,
                                     void* stream);
int32_t nrx_ofdm_demodulate_awgn_f64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                     int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                     const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride,
                                     uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids, void* grid,
                                     void* stream);
/* float32 waveform and transform, complex128 grid out: where the float32 waveform chain (fast mode) hands over to the float64
 * estimator.  The values are those of the _f32 entries converted to double. */
int32_t nrx_ofdm_demodulate_f32o64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                   int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                   const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream);
int32_t nrx_ofdm_demodulate_awgn_f32o64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                        int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                        const int32_t* cp_lens, int32_t n_sym, const void* sigma, int32_t sigma_stride,
                                        uint64_t seed, uint64_t stream_id, int64_t batch_offset, const int64_t* item_ids,
                                        void* grid, void* stream);
/* Grid.precode (wideband F, grid.py:505-516) fused into the modulator's load: layers (n_items, n_layers, n_sym, K),
 * f: per item (f_stride = n_ports*n_layers) or shared (f_stride = 0) n_ports x n_layers; wave rows = item*n_ports+port.
 * Same arithmetic as nrx_precode_* followed by nrx_ofdm_modulate_*; the precoded grid is never materialised.
 * Symbols are transformed in parallel; tails_ws: caller-owned scratch of n_items*n_ports*n_sym*window_len complex
 * samples (may be NULL when window_len == 0) that carries the windowed symbol tails between the two launches. */
int32_t nrx_ofdm_modulate_precoded_f32(const void* layers, int32_t n_items, int32_t n_layers, int32_t n_ports,
                                       const void* f, int64_t f_stride, int32_t K, int32_t nfft, const int32_t* cp_lens,
                                       int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws,
                                       void* stream);
int32_t nrx_ofdm_modulate_precoded_f64(const void* layers, int32_t n_items, int32_t n_layers, int32_t n_ports,
                                       const void* f, int64_t f_stride, int32_t K, int32_t nfft, const int32_t* cp_lens,
                                       int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws,
                                       void* stream);
/* nrx_ofdm_modulate_* with the symbols transformed in parallel (one workgroup per (row, symbol) instead of one per row):
 * the same samples; tails_ws as above (n_rows*n_sym*window_len complex samples, may be NULL when window_len == 0). */
int32_t nrx_ofdm_modulate_sym_f32(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens,
                                  int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws,
                                  void* stream);
int32_t nrx_ofdm_modulate_sym_f64(const void* grid, int32_t n_rows, int32_t K, int32_t nfft, const int32_t* cp_lens,
                                  int32_t n_sym, int32_t window_len, void* wave, int64_t wave_stride, void* tails_ws,
                                  void* stream);
int32_t nrx_ofdm_demodulate_f32(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream);
int32_t nrx_ofdm_demodulate_f64(const void* wave, int64_t wave_stride, int64_t wave_len, const int32_t* t_off,
                                int32_t t_off_stride, int32_t n_items, int32_t n_ant, int32_t K, int32_t nfft,
                                const int32_t* cp_lens, int32_t n_sym, double cp_offset_ratio, void* grid, void* stream);

/* ------------------------------------------------------------------------------------- tapped-delay-line channel
 * cdl.py:641-645,741-811,672-738,871-887 getPathGains, time-varying part: gains (n_items,n_t,n_rx,n_tx,P) with
 * P = n_clusters (+1, LOS first, when A_los != NULL).  A: (n_rx,n_tx,n_clusters,n_rays) complex128 static
 * coefficients (fields x polarisation x location x sqrt(P_n/M) x normalisations); nu: (n_clusters,n_rays)
 * Doppler shifts in Hz; times: (n_items,n_t) seconds (channelmodel.py:328-334 chanGainSamples/sampleRate). */
int32_t nrx_cdl_gains_f64(const void* A, const double* nu, const void* A_los, double nu_los, const double* times,
                          int32_t n_items, int32_t n_t, int32_t n_rx, int32_t n_tx, int32_t n_clusters,
                          int32_t n_rays, void* gains, void* stream);
/* The same with ray coefficients of their own for every item -- A (n_items,n_rx,n_tx,n_clusters,n_rays), nu
 * (n_items,n_clusters,n_rays): the statistical TDL model that redraws its angles and phases for every slot
 * (tdl.py:1043-1067 getPathGains, sosType 'Xiao'). */
int32_t nrx_cdl_gains_items_f64(const void* A, const double* nu, const void* A_los, double nu_los, const double* times,
                                int32_t n_items, int32_t n_t, int32_t n_rx, int32_t n_tx, int32_t n_clusters,
                                int32_t n_rays, void* gains, void* stream);
/* channelmodel.py:343-346: cir (n_items,n_t,n_rx,n_tx,cl) = gains x coeff (n_paths,cl real); chan_offset
 * (nullable, n_items int32) = argmax_l sum_r |sum_{c<nc,t} cir| over the first nc instants. */
int32_t nrx_cir_f64(const void* gains, const double* coeff, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                    int32_t n_tx, int32_t n_paths, int32_t cl, void* cir, int32_t* chan_offset, void* stream);
/* channelmodel.py:362-400 getChannelMatrix: H (n_items,nc,K,n_rx,n_tx) from the first nc CIRs of each item. */
int32_t nrx_channel_matrix_f64(const void* cir, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx, int32_t n_tx,
                               int32_t cl, const int32_t* chan_offset, int32_t K, int32_t nfft, void* H, void* stream);
/* The same transform at n_k consecutive subcarriers from k0 only (direct DFT): H_sub (n_items,nc,n_k,n_rx,n_tx). */
int32_t nrx_channel_matrix_sub_f64(const void* cir, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                                   int32_t n_tx, int32_t cl, const int32_t* chan_offset, int32_t K, int32_t nfft,
                                   int32_t k0, int32_t n_k, void* H, void* stream);
/* channelmodel.py:343-346 + :362-400 in one launch, without a CIR in memory (the time-domain link filters in path form and needs the
 * CIR for nothing else): chan_offset (n_items) as nrx_cir_f64 computes it and H_sub (n_items,nc,n_k,n_rx,n_tx) as
 * nrx_channel_matrix_sub_f64 does, bit-identical to that pair.  NRX_E_UNSUPPORTED unless n_k == 12, n_paths is 13, 14, 15, 23 or 24 (the CDL / TDL profiles) and the tap
 * matrix + the (tap, subcarrier) twiddles fit 150 KB of LDS: the caller then runs the two separate entries. */
int32_t nrx_chan_setup_f64(const void* gains, const double* coeff, int32_t n_items, int32_t n_t, int32_t nc, int32_t n_rx,
                           int32_t n_tx, int32_t n_paths, int32_t cl, int32_t K, int32_t nfft, int32_t k0, int32_t n_k,
                           int32_t* chan_offset, void* H, void* stream);
/* pdsch.py:1080-1131 getPrecodingMatrix (one precoding group): mean of the n_avg (n_rx x n_tx) matrices of each item,
 * SVD, F = V[:, :n_layers]/sqrt(n_layers) -> (n_items,n_tx,n_layers).  Column phases are implementation defined. */
int32_t nrx_svd_precoder_f64(const void* H_block, int32_t n_items, int32_t n_avg, int32_t n_rx, int32_t n_tx,
                             int32_t n_layers, void* F, void* stream);
/* "Perfect" channel state of the BLER harness (PDSCH-BLER.ipynb: channelMatrix @ precoder[None,...]):
 * out (n_items,lk,n_rx,n_layers) = H (n_items,lk,n_rx,n_tx) x F (n_tx,n_layers; item b at F + b*f_stride). */
int32_t nrx_effective_channel_f64(const void* H, const void* F, int64_t f_stride, int32_t n_items, int32_t lk,
                                  int32_t n_rx, int32_t n_tx, int32_t n_layers, void* out, void* stream);
/* channelmodel.py:403-448 applyToSignal: x (n_items,n_tx,ns) -> y (n_items,n_rx,ns) with the CIR of gain set
 * sym(n) of the OUTPUT sample; cir1 (n_items,n_sets,n_rx,n_tx,cl); set_lens: HOST array of n_sets whole-symbol
 * lengths (bwp.getSymLens(), the last set also covers samples beyond their sum).  n_rx in {1,2,4,8}. */
int32_t nrx_apply_td_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* cir1, int32_t n_sets,
                         int32_t n_rx, int32_t cl, const int32_t* set_lens, void* y, void* stream);

/* The same filter in the reference's own path form (per-path fractional-delay FIR, then the per-symbol gain mix):
 * gains1 (n_items,n_sets,n_rx,n_tx,n_paths) complex128; taps (n_paths,flen) the non-zero window of each row of the
 * coefficient matrix (channelmodel.py:292-318), tap_off (n_paths) its first column; hist >= max(tap_off)+flen-1.
 * Identical result up to summation order, ~4.6x fewer FMAs at CDL-C 4x4. */
int32_t nrx_apply_td_paths_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                               int32_t n_sets, int32_t n_rx, int32_t n_paths, const double* taps,
                               const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens, void* y,
                               void* stream);
/* ... and, from the same pass, the power sums of y over its CP-stripped samples (Waveform.getRePower, waveform.py:107-117: the
 * nfft samples of every symbol from round(cpLen/2) on, all Nr rows): pow_acc (n_items, *n_part, 3) float64 = (sum re, sum im,
 * sum |y|^2) per wave, no atomics; pow_capacity = doubles available.  The last of the n_sets spans is the tail beyond the
 * slot and carries no symbol.  nrx_noise_level_finish_f64 turns the sums into the variance / sigma / noise variance of
 * nrx_noise_level_f64 with count = Nr * (n_sets-1) * nfft.  NRX_E_UNSUPPORTED when the geometry has no register-tiled
 * instantiation (filter length != 16 or Nr > 4): call nrx_apply_td_paths_f64 and nrx_noise_level_f64 then. */
int32_t nrx_apply_td_paths_pow_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                   int32_t n_sets, int32_t n_rx, int32_t n_paths, const double* taps,
                                   const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                   void* y, int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part,
                                   void* stream);
/* The float32 waveform chain's filter (opt-in fast mode; the reference computes in complex128): x, y, gains1 complex64, taps
 * float32, packed float32 arithmetic.  pow_acc may be NULL (no power sums; nfft, pow_capacity, n_part then unused).
 * NRX_E_UNSUPPORTED for filter lengths other than 16 or Nr not in {1, 2, 4}: convert and call the float64 entry. */
int32_t nrx_apply_td_paths_pow_f32(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1,
                                   int32_t n_sets, int32_t n_rx, int32_t n_paths, const float* taps,
                                   const int32_t* tap_off, int32_t flen, int32_t hist, const int32_t* set_lens,
                                   void* y, int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part,
                                   void* stream);
/* channelmodel.py:403-448 applyToSignal by OVERLAP-SAVE (nrx_chan_os.hip): the same operator as nrx_apply_td_paths_f64 -- y[r][n] =
 * sum_t sum_p gains1[set(n)][r][t][p] * (taps_p * x_t)[n - off_p] -- as 1024-point circular convolutions per gain set, the Nr x Nt
 * spectra of a set held in registers (nothing but x and y touches HBM).
 * nrx_td_path_spectra_f64: spec (n_paths,1024) complex128 = the 1024-point spectrum of every row of the coefficient matrix
 * (channelmodel.py:292-318: taps (n_paths,flen) at column tap_off[p]) in the transform's own position order, scaled by 1/1024 --
 * a constant of the channel, computed once per link.
 * nrx_apply_td_os_f64: x (n_items,n_tx,ns), gains1 (n_items,n_sets,n_rx,n_tx,n_paths), hist >= max(tap_off)+flen-1 -> y
 * (n_items,n_rx,ns); pow_acc != NULL also leaves the power sums of nrx_apply_td_paths_pow_f64 (one triple per wave: *n_part per
 * item).  Same values as the path form up to rounding (|diff| ~ 1e-15 of the largest sample).  NRX_E_UNSUPPORTED unless
 * n_rx == n_tx in {1,2,4} and hist <= 640: call nrx_apply_td_paths_* then. */
int32_t nrx_td_path_spectra_f64(const double* taps, const int32_t* tap_off, int32_t n_paths, int32_t flen, void* spec, void* stream);
int32_t nrx_apply_td_os_f64(const void* x, int32_t n_items, int32_t n_tx, int64_t ns, const void* gains1, int32_t n_sets,
                            int32_t n_rx, int32_t n_paths, const void* spec, int32_t hist, const int32_t* set_lens, void* y,
                            int32_t nfft, double* pow_acc, int64_t pow_capacity, int32_t* n_part, void* stream);
int32_t nrx_noise_level_finish_f64(const double* acc, int32_t n_part, int64_t count, int32_t n_batch, void* var_out,
                                   const double* snr_lin, int32_t snr_stride, double mult, void* sigma_out, void* nv_out,
                                   double nv_mult, void* stream);

/* grid.py:505-516 Grid.precode with ONE precoder for all subcarriers, moved behind the modulator: out
 * (n_items,n_sets,n_rx,n_layers,n_paths) = sum_t gains1[.,.,r,t,p] * F[.,t,l] (F: (n_tx,n_layers) complex128 per item,
 * f_stride elements apart; 0 = shared).  nrx_apply_td_paths_f64 on the n_layers LAYER waveforms with these gains equals the
 * filter on the n_tx precoded waveforms (precoding commutes with IFFT, cyclic prefix and windowing; channelmodel.py:431-447). */
int32_t nrx_fold_precoder_f64(const void* gains1, const void* F, int64_t f_stride, int32_t n_items, int32_t n_sets,
                              int32_t n_rx, int32_t n_tx, int32_t n_layers, int32_t n_paths, void* out, void* stream);

/* ------------------------------------------------------------------------------------- LS channel estimation
 * grid.py:874-975 estimateChannelLS(polarInt=False, kernel='linear') (channel estimate; the noise-variance
 * branch grid.py:808-851 is not on the graded path).  rx (n_batch,nr,L,K); pilots (n_sets,P,n_ds,n_k) pilot
 * values at the port's own subcarriers port_ks (P,n_k, device int32); pil_set (nullable, device): pilot set of
 * item b (slotNoInFrame-dependent DMRS); dmrs_syms: HOST array of n_ds symbol indices.
 * hest out: (n_batch,L,K,nr,P). */
int32_t nrx_chest_ls_f32(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                         const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                         int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream);
int32_t nrx_chest_ls_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                         const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                         int32_t K, int32_t nr, int32_t P, void* hest, int32_t n_batch, void* stream);

/* estimateChannelLS with the interpolator choice of the notebooks (grid.py:874-975 -> estimateChannelLsEx grid.py:740-806,
 * int2d=False, kernel='linear'): polar != 0 selects utils.py:38-42 polarInterpolate along the subcarriers (np.unwrap of
 * the angle in NumPy's operation order, angle and magnitude inter/extrapolated separately; symbols stay complex-linear,
 * grid.py:866).  pol_ws (polar only): scratch of n_batch*(n_ds/l_cdm)*nr*P*(n_k/k_cdm)*2 doubles.  hk_out (nullable):
 * the estimates at the DMRS time groups after subcarrier interpolation, (n_batch, n_ds/l_cdm, K, nr, P) complex128
 * (hEstAtPilotSyms of grid.py:806) -- the input of nrx_chest_noise_f64.  hest out: (n_batch,L,K,nr,P). */
int32_t nrx_chest_ls_ex_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                            const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                            int32_t K, int32_t nr, int32_t P, int32_t polar, void* pol_ws, void* hk_out, void* hest,
                            int32_t n_batch, void* stream);

/* Noise side output of estimateChannelLsEx, grid.py:808-837 (before scaleNoiseVar): per port the estimate hk goes to the
 * delay domain (K-point inverse DFT, any K), is cut by the raised-cosine window (2*rise non-zero taps, `win` = their
 * weights: taps 0..rise-1 then K-rise..K-1), comes back to the pilot subcarriers, and the residuals against the raw LS
 * values are written to deltas (n_batch, P*n_ds*n_k*nr) complex128; their np.var is the raw noise variance
 * (nrx_noise_level_f64 computes it).  tw: e^{2 pi i q/K}, q = 0..K-1 (complex128, caller table); cir_ws: scratch of
 * n_batch*(n_ds/l_cdm)*nr*P*2*rise complex128.  QUIRK kept (grid.py:823): the denoised estimate of EVERY port is sampled
 * at the pilot subcarriers of the LAST port -- row P-1 of port_ks, or ks_sample (device, n_k entries) when the call
 * covers only some of the ports (CSI-RS ports that sit on different symbols are estimated in separate calls). */
int32_t nrx_chest_noise_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                            const int32_t* ks_sample, const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                            int32_t K, int32_t nr, int32_t P, const void* hk, const void* tw, const double* win,
                            int32_t rise, void* cir_ws, void* deltas, int32_t n_batch, void* stream);

/* The non-default interpolators of estimateChannelLsEx (grid.py:740-871: kernel = 'nearest' | 'quadratic' |
 * 'thin_plate_spline' | 'multiquadric', int2d) in two steps.
 *   nrx_chest_pilot_means_f64: the LS estimates rx/pilot averaged over each CDM group (grid.py:775-793) as pairs of
 *     doubles, out[row][j][2], row = ((b*(n_ds/l_cdm) + tg)*nr + r)*P + p, j < n_k/k_cdm: (re, im), or with polar != 0
 *     (np.unwrap(angle), abs) as utils.py:39 forms them.
 *   nrx_interp_taps_f64: out[q] = sum_t w[q][t] * in[idx[q][t]] on both components; polar pairs are recombined as
 *     abs * e^{i angle} (utils.py:42).  idx/w (n_tabs, n_out, n_taps) hold the linear operator of the reference's
 *     interpolator for one pilot geometry (utils.py:26-35 interp1d / RBFInterpolator with nearest neighbours in one
 *     dimension, grid.py:853-861 in two) -- a host table, like the RE indices.  Rows are (outer, in) with in < inner;
 *     table = in % n_tabs; element j of a row is at in[outer*in_outer + in*in_inner + j*in_j], output q at
 *     out[outer*out_outer + in*out_inner + q*out_q] (units of complex128), so the same entry serves the subcarrier
 *     axis, the symbol axis and the 2-D case without a transpose. */
int32_t nrx_chest_pilot_means_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                  const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k,
                                  int32_t L, int32_t K, int32_t nr, int32_t P, int32_t polar, void* out, int32_t n_batch,
                                  void* stream);
int32_t nrx_interp_taps_f64(const void* in, const int32_t* idx, const double* w, int32_t n_taps, int32_t n_tabs,
                            int64_t tab_stride, int64_t n_outer, int32_t inner, int32_t n_out, int64_t in_outer,
                            int64_t in_inner, int64_t in_j, int64_t out_outer, int64_t out_inner, int64_t out_q,
                            int32_t polar, void* out, void* stream);

/* Grid.estimateTimingOffset (grid.py:592-622): xc[d] = sum over rx antennas and ports of |sum_n rx[r][n+d] conj(ref[p][n])|
 * for every lag d = 0..n_samples-1 (the second half of scipy.signal.correlate(..., 'full')); the caller takes the argmax.
 * rx (nr, n_samples), ref (P, n_ref) complex128; ref is zero outside [ref_start, ref_start+ref_len) (the CSI-RS symbols),
 * which is all the kernel reads. */
int32_t nrx_xcorr_abs_f64(const void* rx, const void* ref, int32_t n_samples, int32_t n_ref, int32_t ref_start,
                          int32_t ref_len, int32_t nr, int32_t P, double* xc, void* stream);

/* CsiReport.getSINR (csifeedback.py:419-433), the inner loop of the PMI / rank search (csifeedback.py:450-536): for every
 * codebook entry W (n_cb, nt, nl) and channel sample H (n_re, nr, nt), heff = H W and
 * sinr[cb][re][l] = 1 / (noise_var * [(heff^H heff + noise_var I)^-1]_ll) - 1  (float64).  nl <= nr <= 8. */
int32_t nrx_csi_sinr_f64(const void* h, int32_t n_re, int32_t nr, int32_t nt, const void* w, int32_t n_cb, int32_t nl,
                         double noise_var, double* sinr, void* stream);

/* ------------------------------------------------------------------------------------- per-PRG precoding
 * pdsch.py:1132-1165 (getPrecodingMatrix with prgSize 2/4, or the wideband precoder of a partial allocation: a list of
 * (rbList, F) groups) + grid.py:482-493 (Grid.precode with that list).  The group lists are host bookkeeping (the
 * reference closes a group when the first PRB of the next one arrives, so PRB 0 stands alone and trailing PRBs may be
 * left without a precoder); the device work is:
 *   nrx_group_mean_f64      Hm (n,G,E) = mean over L symbols x the group's n_k contiguous subcarriers of H (n,L,K,E)
 *   nrx_svd_precoder_f64    on the n*G means (n_avg = 1)
 *   nrx_precode_prg_*       out (n,Nt,L,K) = F[b][k2g[k]] x grid (n,Nl,L,K); k2g (device int32, K entries, -1 = no
 *                           group: zero output); f_stride = elements between items' F blocks (0 = shared)
 *   nrx_effective_channel_prg_f64   H @ F[k2g[k]] for the perfect-CSI harness path. */
int32_t nrx_group_mean_f64(const void* H, int32_t n_items, int32_t L, int32_t K, int32_t E, const int32_t* k0,
                           const int32_t* nk, int32_t G, void* Hm, void* stream);
int32_t nrx_precode_prg_f32(const void* grid, const void* f, int64_t f_stride, const int32_t* k2g, int32_t nl, int32_t nt,
                            int32_t L, int32_t K, void* out, int32_t n_batch, void* stream);
int32_t nrx_precode_prg_f64(const void* grid, const void* f, int64_t f_stride, const int32_t* k2g, int32_t nl, int32_t nt,
                            int32_t L, int32_t K, void* out, int32_t n_batch, void* stream);
int32_t nrx_effective_channel_prg_f64(const void* H, const void* F, int64_t f_stride, const int32_t* k2g, int32_t n_items,
                                      int32_t L, int32_t K, int32_t n_rx, int32_t n_tx, int32_t n_layers, void* out,
                                      void* stream);

/* Grid.equalize (grid.py:626-694) on the harness's PERFECT channel state (PDSCH-BLER.ipynb cell 2: channelMatrix @ precoder) without a
 * channel matrix in memory (nrx_mmse_paths.hip).  getChannelMatrix (channelmodel.py:362-400) is linear in the path gains, so
 *   Hest[c][k][r][l] = exp(+2 pi i k' chan_off / nfft) * sum_p gains[c][r][l][p] * S_p[k],   k' = (k - K/2) mod nfft
 * with S_p = the nfft-point spectrum of row p of the coefficient matrix at the K centred bins -- a constant of the channel:
 * nrx_td_path_spectra_bins_f64 (taps (n_paths,flen) at column tap_off[p]) -> spec (n_paths,K) complex128, once per link.
 * nrx_mmse_equalize_paths_f64: rx (n_batch,n_rx,L,K); gains (n_batch,n_sets>=L,n_rx,n_layers,n_paths) = the path gains with the wideband
 * precoder folded in (nrx_fold_precoder_f64), gain set l = symbol l; chan_off (n_batch) = chanOffset (channelmodel.py:345-346);
 * sym_mask: bit l set = symbol l is equalised (the others' outputs stay untouched) -> eq, scale (n_batch,n_layers,L,K) like
 * nrx_mmse_equalize_f64.  Same values as nrx_channel_matrix_f64 -> nrx_effective_channel_f64 -> nrx_mmse_equalize_f64 up to rounding.
 * NRX_E_UNSUPPORTED unless n_rx in {1,2,4} and n_layers <= 4. */
/* nrx_chan_setup_f64 (chanOffset + the channel matrix at n_k subcarriers from k0 on, channelmodel.py:343-346, 362-400) with the sums over the
 * cl taps taken out of the per-row work: the offset from G[r][p] = sum_{c,t} gains, the matrix from the paths' spectra (spec (n_paths,
 * spec_stride >= K) of nrx_td_path_spectra_bins_f64): H (n_items,nc,n_k,n_rx,n_tx) = exp(2 pi i k' o / nfft) sum_p gains[c][rt][p] S_p[k].  Same
 * values up to the order of the sums (~1e-16); any n_k, any path count. */
int32_t nrx_chan_setup_paths_f64(const void* gains, const double* coeff, const void* spec, int64_t spec_stride, int32_t n_items, int32_t n_t,
                                 int32_t nc, int32_t n_rx, int32_t n_tx, int32_t n_paths, int32_t cl, int32_t K, int32_t nfft, int32_t k0,
                                 int32_t n_k, int32_t* chan_offset, void* H, void* stream);
int32_t nrx_td_path_spectra_bins_f64(const double* taps, const int32_t* tap_off, int32_t n_paths, int32_t flen, int32_t K, int32_t nfft,
                                     void* spec, void* stream);
int32_t nrx_mmse_equalize_paths_f64(const void* rx, const void* gains, int32_t n_sets, const void* spec, const int32_t* chan_off,
                                    const double* noise_var, int32_t nv_stride, int32_t n_rx, int32_t n_layers, int32_t n_paths,
                                    int32_t L, int32_t K, int32_t nfft, uint32_t sym_mask, void* eq, void* scale, int32_t n_batch,
                                    void* stream);

/* nrx_chest_ls_f64 + nrx_mmse_equalize_f64 in one call without materialising the (L, K, Nr, P) estimate (at most two
 * DMRS time groups): hk_ws is caller-owned scratch of n_batch * (n_ds/l_cdm) * (K + n_k/k_cdm) * nr * P complex128 (the
 * estimates at the DMRS time groups, then the CDM-group means they are interpolated from); eq (n,P,L,K) complex128,
 * scale (n,P,L,K) float64.  Results are identical to the two separate calls. */
int32_t nrx_chest_ls_mmse_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                              const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                              int32_t K, int32_t nr, int32_t P, const double* noise_var, int32_t nv_stride, void* hk_ws,
                              void* eq, double* scale, int32_t n_batch, void* stream);
/* The same for the OFDM symbols of sym_mask only (bit l = symbol l, L <= 32; the equalised values of symbols without data REs
 * -- the DMRS symbols -- are read by nobody): eq / scale of the other symbols are left untouched. */
int32_t nrx_chest_ls_mmse_syms_f64(const void* rx, const void* pilots, const int32_t* pil_set, const int32_t* port_ks,
                                   const int32_t* dmrs_syms, int32_t n_ds, int32_t l_cdm, int32_t k_cdm, int32_t n_k, int32_t L,
                                   int32_t K, int32_t nr, int32_t P, const double* noise_var, int32_t nv_stride, void* hk_ws,
                                   void* eq, double* scale, int32_t n_batch, uint32_t sym_mask, void* stream);

/* ------------------------------------------------------------------------------------------------ Polar
 * Control-channel codec (DCI / PBCH / UCI).  The code construction (polar.py:298-408 PolarBase.initialize) is
 * integer bookkeeping done by the host binding; the kernels take its index tables.
 *
 * nrx_polar_encode -- polar.py:527-564 PolarEncoder.encode.  cbs: n_cw x K bits (CRC attached), in_il: K input
 *   interleaver indices or NULL, msg_pos: K message-bit positions, pc_pos: n_pc parity-check positions.
 *   coded: n_cw x N.
 * nrx_polar_rate_match -- polar.py:567-603 rateMatch; gather[e] = index into the N coded bits (sub-block
 *   interleave o bit selection o coded-bit interleave composed by the host).
 * nrx_polar_rate_recover_f64 -- polar.py:882-928 recoverRate; deinterleave = inverse coded-bit interleaver (E) or
 *   NULL, inv_subblock = inverse sub-block interleaver (N).  E >= N adds the repeated LLRs (TS 38.212 5.4.1.2; the
 *   reference raises there, polar.py:914-915).
 * nrx_polar_scl_decode_f64 -- polar.py:606-720 SclDecoder + :931-982 PolarDecoder.decode: clip to +-20, min-sum
 *   SCL with list_size <= 8, CRC-aided pick.  info_mask: N bytes, bit 0 = non-frozen leaf (message or parity-check
 *   bit), n_info of them; optionally, on a frozen leaf i, bits 1..4 = s0 > 0 announce that leaves [i, i + 2^s0) are
 *   all frozen and i is a multiple of 2^s0 (a rate-0 node): the kernel then expands that subtree level by level and
 *   adds its 2^s0 penalties in leaf order -- same values, same path costs, far fewer sequential steps; msg_src[m]: which non-frozen leaf (in leaf order) carries message bit m after input
 *   de-interleaving (K entries).  msg_out: n_cw x K (first CRC-passing candidate, else the cheapest), crc_ok: n_cw;
 *   optional cand_out: n_cw x list_size x K and cost_out: n_cw x list_size (all candidates, cheapest first).
 *   crc_poly_id -1 = no CRC (cheapest candidate).  crc_expect (nullable, device, n_cw entries): the value the CRC register
 *   over message + parity must end at for code word cw instead of 0 -- a mask XORed onto the parity bits (the RNTI of a
 *   DCI, the effect of the 24 ones TS 38.212 7.3.2 prepends to the CRC input) moves the end value to the register value of
 *   the mask alone, because the CRC is linear; one entry per code word lets one launch test candidates x RNTIs. */
int32_t nrx_polar_encode(const uint8_t* cbs, int32_t n_cw, int32_t K, int32_t N, const int32_t* in_il,
                         const int32_t* msg_pos, const int32_t* pc_pos, int32_t n_pc, uint8_t* coded, void* stream);
int32_t nrx_polar_rate_match(const uint8_t* coded, int32_t n_cw, int32_t N, int32_t E, const int32_t* gather,
                             uint8_t* out, void* stream);
int32_t nrx_polar_rate_recover_f64(const double* llr, int32_t n_cw, int32_t N, int32_t E, int32_t K,
                                   const int32_t* deinterleave, const int32_t* inv_subblock, double* out,
                                   void* stream);
int32_t nrx_polar_scl_decode_f64(const double* llr, int32_t n_cw, int32_t N, int32_t list_size,
                                 const uint8_t* info_mask, int32_t n_info, const int32_t* msg_src, int32_t K,
                                 int32_t crc_poly_id, const uint32_t* crc_expect, uint8_t* msg_out, uint8_t* crc_ok,
                                 uint8_t* cand_out, double* cost_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NRX_H_ */
