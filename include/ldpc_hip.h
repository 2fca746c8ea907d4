/*
 * ldpc_hip.h -- C ABI of libldpc_hip.so: MI355X (gfx950) belief-propagation LDPC decoding.
 *
 * This is the drop-in boundary for the BP hot path of thadikari/ldpc_decoders.  The upstream code is pure
 * Python on this path; its own FFI precedent is the ctypes binding of the ADMM projection
 * (src/parity_polytope/exact.py:12-21,49-53 -> extern "C" proj_vec/proj_csr, projection.cpp:252,266):
 * extern "C", caller-allocated buffers, plain pointers and sizes.  The entry points below follow that
 * pattern; each cites the upstream interface it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative LDPC_E_* code; ldpc_last_error() gives the text
 *     (thread-local).  Nothing throws across the boundary.
 *   - "dev" pointers are device (HBM) addresses on the handle's GPU; "host" pointers are ordinary memory.
 *   - frames are rows: priors / y / xhat are [B, n] row-major, exactly what numpy hands over.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Work is enqueued on it; calls may
 *     synchronise that stream internally (early-termination polling) but never the device.
 *   - handles are not thread-safe: one decoder per host thread / stream.
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ldpc_code_s* ldpc_code_t;
typedef struct ldpc_decoder_s* ldpc_decoder_t;

enum { LDPC_ALG_MSA = 0, LDPC_ALG_SPA = 1, LDPC_ALG_BEC = 2 };      /* decoder selector: src/utils.py:16, main.py:12 */
enum { LDPC_DTYPE_F32 = 0, LDPC_DTYPE_F64 = 1,                       /* message arithmetic                             */
       LDPC_DTYPE_F16 = 2 };  /* fp16 STORAGE of the check messages on the streaming kernels, fp32 arithmetic, fp32 priors / channel output:
                               * a throughput mode for codes whose state lives in HBM (the E-sized traffic halves); held to a
                               * stated tolerance, never the parity mode.  ldpc_decoder_create only. */
enum { LDPC_BACKEND_AUTO = 0, LDPC_BACKEND_STREAM = 1, LDPC_BACKEND_FUSED = 2 };
enum { LDPC_CH_BIAWGN = 0, LDPC_CH_BSC = 1, LDPC_CH_BEC = 2 };       /* channel selector: src/models.py:3               */
enum { LDPC_CH_RAW_OBSERVATION = 0x100 };  /* or-ed into LDPC_CH_BIAWGN for ldpc_channel: write y itself, not -2y/sigma^2 */
enum { LDPC_FLAG_NO_EARLY_EXIT = 1 };                                /* NOT reference behaviour: run exactly max_iter   */
/* Exact-in-fp32 min-sum (no upstream counterpart; the injection point is BPA.decode(y, priors), src/bpa.py:17): priors rounded to
 * multiples of 2^-k.  Min-sum only adds, subtracts and compares, so on such priors fp32 arithmetic reproduces the fp64 reference BIT FOR
 * BIT for as long as every prior and every check message stays below L = 2^(24-k) / (dv_max + 2), rounded down to a power of two (2^(21-k)
 * for variable degrees up to 6): then no partial sum, marginal or v2c of any sweep reaches 2^(24-k).  The LDS-resident fp32 kernels check
 * exactly that (priors once per frame, outgoing check magnitudes in every sweep) and count the frames
 * that do not (ldpc_decoder_grid_violations -- a run is exact iff the count is 0).  k = 0..23.
 *   ldpc_simulate / ldpc_decode: flags | LDPC_FLAG_PRIOR_GRID(k)   (simulate: quantises the generated priors AND arms the guard; decode: arms the guard;
 *                                both return LDPC_E_UNSUPPORTED for an fp32 / fp16 decoder that runs on the streaming kernels, which have no guard;
 *                                fp64 decoders need none: the flag is a no-op there)
 *   ldpc_channel:                channel | LDPC_CH_PRIOR_GRID(k)   (quantises the LLRs it writes; any dtype) */
#define LDPC_FLAG_PRIOR_GRID(k) ((((uint32_t)(k)) + 1u) << 8)
#define LDPC_FLAG_PRIOR_GRID_OF(flags) ((int)(((flags) >> 8) & 0x1fu) - 1) /* -1: off */
#define LDPC_CH_PRIOR_GRID(k) ((((int)(k)) + 1) << 12)
#define LDPC_CH_PRIOR_GRID_OF(channel) ((((channel) >> 12) & 0x1f) - 1)
enum { LDPC_E_ARG = -1, LDPC_E_HIP = -2, LDPC_E_GRAPH = -3, LDPC_E_UNSUPPORTED = -4, LDPC_E_NOMEM = -5 };

/* counters written by ldpc_count_errors / ldpc_simulate (int64 each) */
enum { LDPC_CNT_TOT = 0, LDPC_CNT_WEC = 1, LDPC_CNT_BEC = 2, LDPC_CNT_ITER_SUM = 3, LDPC_CNT_HIST0 = 4 };

const char* ldpc_last_error(void);
int ldpc_abi_version(void);
int ldpc_device_count(int* count);

/* Tanner graph from the row-major edge list of H: edge k = (edge_chk[k], edge_var[k]) sorted by check, then
 * variable -- the order of `xx, yy = np.where(parity_mtx)` in BPA.__init__ (src/bpa.py:9-15) and bec.SPA.__init__
 * (src/bec.py:77).  Host pointers; the graph is copied to `device` once (CSR + CSC index lists in HBM). */
int ldpc_code_create(int device, int32_t m, int32_t n, int64_t E, const int32_t* edge_chk, const int32_t* edge_var,
                     ldpc_code_t* out);
int ldpc_code_destroy(ldpc_code_t code);
int ldpc_code_info(ldpc_code_t code, int32_t* m, int32_t* n, int64_t* E, int32_t* max_dc, int32_t* max_dv);

/* Decoder = graph + algorithm + arithmetic + workspace.  Replaces the constructors bpa.SPA / bpa.MSA
 * (src/bpa.py:66-84) and bec.SPA / bec.MSA (src/bec.py:70-81,125). */
int ldpc_decoder_create(ldpc_code_t code, int alg, int dtype, int backend, ldpc_decoder_t* out);
int ldpc_decoder_destroy(ldpc_decoder_t dec);
/* backend actually used by the last decode (LDPC_BACKEND_*) and the number of sweeps the batch ran */
int ldpc_decoder_last_stats(ldpc_decoder_t dec, int* backend, int* sweeps);
/* streaming backend: how often the last decode gathered its live frames into dense tiles (per-frame early termination,
 * src/bpa.py:28-29: a frame that has left costs nothing; tiles of 64 frames are re-formed from the live ones) */
int ldpc_decoder_last_repacks(ldpc_decoder_t dec, int* repacks);
/* streaming backend: frames per pass through the kernels (sized once from the free HBM: the reference decodes one frame per call,
 * src/main.py:37-48, so any batch size is the build's own) and how often a failed workspace reservation made ldpc_decode halve it */
int ldpc_decoder_chunk_state(ldpc_decoder_t dec, int64_t* chunk_frames, int* retries);
/* exact-in-fp32 mode: frames in which a message left the range where fp32 sums of grid multiples are exact (a frame caught in a
 * trapping set: its min-sum messages grow geometrically; about 1 in 10^4 at 2 dB), since the last reset.  Such a frame is NOT counted
 * by ldpc_simulate, and ldpc_decode marks it with iters = -1 - sweeps; its global frame index (frame0 + position) is listed in
 * frames[0 .. min(count, cap, 4095)) so that the caller decodes it again in fp64 on the same priors (ldpc_decoders_amd/_device.py does).
 * Synchronises the device. */
int ldpc_decoder_grid_violations(ldpc_decoder_t dec, int64_t* count, int64_t* frames, int64_t cap, int reset);

/* The same redo list WITHOUT a host round trip (exact-in-fp32 Monte-Carlo rounds kept in flight, ldpc_decoders_amd/montecarlo.py): the
 * list as it lives in device memory -- list_dev[0] = frames set aside since the last reset, list_dev[1 .. cap] their global indices --
 * consumed on a stream by ldpc_channel_list (the priors of exactly those frames) -> ldpc_decode (fp64 decoder, `rows` frames: rows beyond
 * the list decode whatever the buffer holds and are not counted) -> ldpc_count_errors_list (counts the first min(list_dev[0], rows) rows,
 * each into the counter row of ITS round: counters_dev + ((frame - frame_base) / round_stride) * counter_stride, so that `nrounds` guarded
 * launches may share one redo pass and still keep one exact counter row per round; redone2_dev[0] += the rows counted, redone2_dev[1] += 1
 * if the list held more than `rows`: the caller must treat those rounds as failed) -> ldpc_decoder_grid_list_reset.  The guarded kernel
 * of the NEXT launch of the same decoder must be ordered behind the reset.  No upstream counterpart (see LDPC_FLAG_PRIOR_GRID). */
int ldpc_decoder_grid_list(ldpc_decoder_t dec, uint64_t** list_dev, int64_t* cap, void* stream);
int ldpc_decoder_grid_list_reset(ldpc_decoder_t dec, void* stream);
int ldpc_channel_list(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, const uint64_t* list_dev,
                      int64_t rows, int32_t n, void* priors_dev, void* stream);
int ldpc_count_errors_list(const uint8_t* xhat_dev, int codeword, const int32_t* iters_dev, const uint64_t* list_dev, int64_t rows, int32_t n,
                           int32_t hist_bins, int64_t* counters_dev, int64_t counter_stride, uint64_t frame_base, uint64_t round_stride,
                           int64_t nrounds, int64_t* redone2_dev, void* stream);

/* Fused-backend plan of this decoder: out8 = {wavefronts per frame (0 = fused backend unavailable), conflict-free LDS gather cycles per sweep, extra bank-conflict
 * cycles with the trivial placement, extra cycles with the planned placement, resident waves per CU, LDS bytes per
 * frame, check rounds, variable rounds}. */
int ldpc_decoder_fused_info(ldpc_decoder_t dec, double* out8);

/* Host-only (no GPU needed): the LDS layout plan the fused backend would use for this (graph, algorithm, arithmetic) -- chosen
 * kernel shape, bank-conflict-minimising placement of checks / variables / edge positions (csrc/ldpc_layout.hpp) -- annealed with
 * `moves` moves (0: the default short run) and stored as <key>.plan in out_dir (NULL: not stored).  moves < 0: what decoder
 * construction does -- a stored plan is used if there is one (info4[0] is then negated), otherwise ONE process per node anneals the
 * default run (lock file next to the plan; the others wait for the file) and keeps it in out_dir / the per-user cache.  Decoders find such files in
 * $LDPC_FUSED_PLAN_DIR, in <package>/plans and in the per-user cache.  No upstream counterpart (the placement is a property of this
 * implementation; the graph arguments are those of ldpc_code_create, i.e. BPA.__init__'s edge lists, src/bpa.py:9-15).
 * info4 = {wavefronts per frame (0: no fused shape for this graph), conflict-free LDS gather cycles per sweep, extra bank-conflict
 * cycles of the trivial placement, extra cycles of the plan}. */
int ldpc_plan_layout(int32_t m, int32_t n, int64_t E, const int32_t* edge_chk, const int32_t* edge_var, int alg, int dtype,
                     int64_t moves, const char* out_dir, double* info4);

/* Name of the LDS-resident kernel this decoder launches (simulate != 0: the Monte-Carlo variant behind ldpc_simulate), exactly as
 * rocprofv3 prints it, e.g. "k_fused_bp<0, 6, 3, 5, 10, 2, true, 0, 3>" -- the key under which its committed PMC counters are filed
 * (profiles/roofline_counters.json).  Empty string: the decoder runs on the streaming kernels.  No upstream counterpart. */
int ldpc_decoder_kernel_name(ldpc_decoder_t dec, int simulate, char* buf, int64_t len);

/* Per-kernel timing for roofline reports: when enabled, decode calls bracket their dominant kernels with HIP events
 * recorded ON THE DECODE STREAM and accumulate elapsed milliseconds / launch counts per kernel class:
 * [0] streaming check pass, [1] streaming variable pass, [2] fused decode kernel, [3] a whole streaming decode, first to last
 * enqueued kernel (the two passes + tile load, syndrome, repack and unpack kernels: [3] - [0] - [1] is what the side kernels cost).
 * Enabling it makes every decode call end with a stream synchronise. */
int ldpc_decoder_profile(ldpc_decoder_t dec, int enable);
int ldpc_decoder_profile_read(ldpc_decoder_t dec, double* ms4, int64_t* launches4, int reset);

/* Batched BPA.decode(y, priors) (src/bpa.py:17-63) / bec.SPA.decode(y) (src/bec.py:83-122).
 *   priors_dev  [B,n] float or double per `dtype` (ignored for LDPC_ALG_BEC)
 *   y0_dev      [B,n] uint8 or NULL: hard received word checked at iteration 0 (src/bpa.py:20,29: BSC), or
 *               the received symbols {0,1,2} for LDPC_ALG_BEC (required)
 *   max_iter    sweep cap; <= 0 means "until every frame has left" as upstream (src/bpa.py:28), bounded at 100000
 *   xhat_dev    [B,n] uint8 out: decisions in {0,1} ({0,1,2} for BEC, 2 = still erased)
 *   iters_dev   [B]  int32 out: sweeps executed by each frame (0 = left at the iteration-0 check, x_hat = y0) */
int ldpc_decode(ldpc_decoder_t dec, const void* priors_dev, const uint8_t* y0_dev, int64_t B, int32_t max_iter,
                uint32_t flags, uint8_t* xhat_dev, int32_t* iters_dev, void* stream);
/* The same decode with PACKED decisions -- the information BPA.decode returns (src/bpa.py:62: n hard decisions) in n bits instead of n
 * bytes:
 *   xhat_bits_dev    [B, W] uint32, W = ceil(n / 32): bit (v & 31) of word (v >> 5) of row f = decision of variable v of frame f
 *                    (little-endian bit order: np.unpackbits(words.view(np.uint8), bitorder="little")[:, :n] gives the bytes of ldpc_decode);
 *                    padding bits of the last word are 0
 *   erased_bits_dev  [B, W] uint32: LDPC_ALG_BEC (required there): bit set = the symbol is still erased (x_hat = 2, src/bec.py:120), its
 *                    decision bit is then 0.  LLR decoders: may be NULL; written as all-zero when given.
 * The LLR decoders on the streaming kernels write the words straight from their decision bit planes (no [B,n] byte array exists). */
int ldpc_decode_bits(ldpc_decoder_t dec, const void* priors_dev, const uint8_t* y0_dev, int64_t B, int32_t max_iter, uint32_t flags,
                     uint32_t* xhat_bits_dev, uint32_t* erased_bits_dev, int32_t* iters_dev, void* stream);
/* Same (on whichever backend the decoder uses: streaming or fused), additionally returning the soft output:
 * marginals_dev [B,n] (`dtype`) = the marginal LLRs (prior + sum of check messages, src/bpa.py:35 -- a local of the
 * reference's loop, captured upstream only through its sum_cols hook) of each frame's LAST executed sweep (0 where a
 * frame never swept).  LLR decoders only; B <= 2^17. */
int ldpc_decode_soft(ldpc_decoder_t dec, const void* priors_dev, const uint8_t* y0_dev, int64_t B, int32_t max_iter,
                     uint32_t flags, uint8_t* xhat_dev, int32_t* iters_dev, void* marginals_dev, void* stream);
/* Same with host buffers (numpy ndpointer style, as exact.proj_csr); copies in, decodes, copies out, synchronises.  The decisions cross
 * PCIe PACKED (ldpc_decode_bits: n / 8 bytes per frame) and are expanded to bytes on the host. */
int ldpc_decode_host(ldpc_decoder_t dec, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                     uint8_t* xhat, int32_t* iters);
/* Host buffers in, packed decisions out (layout of ldpc_decode_bits; erased_bits required for LDPC_ALG_BEC, optional otherwise). */
int ldpc_decode_host_bits(ldpc_decoder_t dec, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                          uint32_t* xhat_bits, uint32_t* erased_bits, int32_t* iters);

/* Channel.send + LLR for frames [frame0, frame0+B) of the all-`codeword` word, Philox4x32-10 keyed by
 * (seed, stream_id, global frame index) -- biawgn.Channel.send/LLR.decode (src/biawgn.py:13-28),
 * bsc (src/bsc.py:11-25), bec.Channel.send (src/bec.py:11-18).  priors_dev [B,n] (`dtype`; NULL for BEC),
 * y_dev [B,n] uint8 (BSC: received bits, BEC: symbols; may be NULL for BI-AWGN).  priors_dev may be NULL for the BSC.
 * LDPC_CH_BIAWGN | LDPC_CH_RAW_OBSERVATION writes the received values y into priors_dev instead of their LLRs. */
int ldpc_channel(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0,
                 int64_t B, int32_t n, void* priors_dev, uint8_t* y_dev, void* stream);

/* The same for a RANDOM codeword per frame -- `--codeword -1`, src/main.py:38: x = code.cb[np.random.choice(K)] -- frame f sends
 * word floor(w * K / 2^32) of codebook_dev ([K,n] uint8, Code.cb of src/codes.py:11-14, K <= 2^31), w = first Philox word of block
 * 0xFFFFFFFE of the frame; sent_dev [B,n] uint8 receives the word each frame sent (for ldpc_count_errors_words). */
int ldpc_channel_words(int channel, int dtype, double param, const uint8_t* codebook_dev, int64_t K, uint64_t seed, uint64_t stream_id,
                       uint64_t frame0, int64_t B, int32_t n, void* priors_dev, uint8_t* y_dev, uint8_t* sent_dev, void* stream);

/* Monte-Carlo counters of main.test (src/main.py:41-45), ACCUMULATED into counters_dev (int64[4 + hist_bins]):
 * tot += B, wec += #frames with errors, bec += bit errors, iter_sum += sum(iters), hist[min(iters, bins-1)] += 1.
 * `sent_dev` is the transmitted word [n] or NULL for the all-`codeword` word; iters_dev may be NULL. */
int ldpc_count_errors(const uint8_t* xhat_dev, const uint8_t* sent_dev, int codeword, const int32_t* iters_dev, int64_t B,
                      int32_t n, int32_t hist_bins, int64_t* counters_dev, void* stream);

/* The same counters from PACKED decisions (ldpc_decode_bits): errors of a frame = popcount((xhat_bits ^ sent) | erased) over its n bits.
 * sent_bits_dev [W] = the transmitted word, packed, or NULL for the all-`codeword` word; erased_bits_dev may be NULL (LLR decoders). */
int ldpc_count_errors_bits(const uint32_t* xhat_bits_dev, const uint32_t* erased_bits_dev, const uint32_t* sent_bits_dev, int codeword,
                           const int32_t* iters_dev, int64_t B, int32_t n, int32_t hist_bins, int64_t* counters_dev, void* stream);

/* The same against one sent word PER FRAME: sent_dev [B,n] (ldpc_channel_words). */
int ldpc_count_errors_words(const uint8_t* xhat_dev, const uint8_t* sent_dev, const int32_t* iters_dev, int64_t B, int32_t n,
                            int32_t hist_bins, int64_t* counters_dev, void* stream);

/* ---- Maximum-likelihood decoding of the short codes by codebook search -----------------------------------------
 * Replaces biawgn.ML (src/biawgn.py:66-78), bsc.ML (src/bsc.py:63-75) and bec.ML (src/bec.py:21-36).  `codebook` is
 * Code.cb (src/codes.py:11-14): [K, n] bytes in {0,1}, host pointer, K <= 2^20 words of n <= 64 bits. */
typedef struct ldpc_ml_s* ldpc_ml_t;
int ldpc_ml_create(int device, const uint8_t* codebook, int64_t K, int32_t n, ldpc_ml_t* out);
int ldpc_ml_destroy(ldpc_ml_t ml);
/* ML.decode for B frames.  y_dev: [B,n] observations -- double or float per `dtype` for LDPC_CH_BIAWGN, uint8 symbols
 * ({0,1}, 2 = erased) for LDPC_CH_BSC / LDPC_CH_BEC.  coef2 (host) holds the constants the upstream constructor
 * computes: {2*noise_var, unused} for BI-AWGN, {log p, log(1-p)} for BSC / BEC.  The log-likelihood of every
 * codeword is evaluated in fp64 in the upstream operation order (including numpy's summation order), so its maximum
 * (best_dev, [B] double) and the set of maximisers (ties_dev [B] = how many; tie_mask_dev [B, ceil(K/32)] uint32,
 * bit k of word k/32 = codeword k attains the maximum) are bit-identical to upstream's.  The pick among the
 * maximisers (math_utils.arg_max_rand, src/math_utils.py:72-74) is maximiser number floor(pick[f] * ties / 2^32) in
 * codebook order, the first one if pick_dev is NULL; index_dev [B] int32 and xhat_dev [B,n] uint8 receive it.  Every
 * output pointer may be NULL. */
int ldpc_ml_decode(ldpc_ml_t ml, int channel, int dtype, const double* coef2, const void* y_dev, int64_t B,
                   const uint32_t* pick_dev, int32_t* index_dev, int32_t* ties_dev, uint32_t* tie_mask_dev, double* best_dev,
                   uint8_t* xhat_dev, void* stream);
/* Channel.send + ML.decode + the counters of main.test for frames [frame0, frame0+B) of the all-`codeword` word; same
 * Philox keying as ldpc_channel, the tie-break word is block 0xFFFFFFFF of the frame.  Accumulates tot/wec/bec. */
int ldpc_ml_simulate(ldpc_ml_t ml, int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id,
                     uint64_t frame0, int64_t B, int64_t* counters_dev, void* stream);

/* ---- ADMM LP decoding ------------------------------------------------------------------------------------------
 * Replaces admm.ADMM (src/admm.py:9-77) together with its native projection (src/parity_polytope/projection.cpp:30-275, bound
 * upstream through ctypes in exact.py:12-53).  Check degrees up to 16. */
typedef struct ldpc_admm_s* ldpc_admm_t;
int ldpc_admm_create(ldpc_code_t code, ldpc_admm_t* out);
int ldpc_admm_destroy(ldpc_admm_t admm);
/* ADMM_Base.decode(y, gamma) for B frames (src/admm.py:42-69).  gamma_dev [B,n] double: the LLR vectors the channel wrappers
 * hand over (src/biawgn.py:28, src/bsc.py:25, src/bec.py:38-45).  mu, eps, max_iter: the constructor's kwargs (max_iter <= 0 =
 * no cap upstream; bounded at 100000 here).  x_dev [B,n] double out: x_hat as it stands at return, BEFORE
 * math_utils.pseudo_to_cw (src/math_utils.py:28-34); iters_dev [B] int32: iter_count at return (the bin upstream's histogram
 * increments, src/admm.py:49); converged_dev [B] uint8 or NULL: 1 where the stopping test fired.  fp64 throughout, in the
 * upstream operation order: results are bit-identical to upstream's for the same gamma. */
int ldpc_admm_decode(ldpc_admm_t admm, const double* gamma_dev, int64_t B, double mu, double eps, int32_t max_iter, double* x_dev,
                     int32_t* iters_dev, uint8_t* converged_dev, void* stream);

/* how often the last ldpc_admm_decode gathered its live frames into dense tiles (frames leave one by one, src/admm.py:65-66) */
int ldpc_admm_last_repacks(ldpc_admm_t admm, int* repacks);
/* which kernels the last ldpc_admm_decode ran on: 0 = the streaming kernels (state in HBM, any code), 1 = the LDS-resident kernel (one
 * workgroup per frame, z / lambda / x in the LDS of the CU: codes whose checks all have six edges and whose frame fits 160 KB).
 * Same arithmetic either way: estimates and iteration counts are bit-identical.  LDPC_ADMM_BACKEND=stream forces 0. */
int ldpc_admm_last_backend(ldpc_admm_t admm, int* backend);

/* Profiling aid: coalesced 4-byte-per-lane device copy of a known size, used to calibrate the profiler's HBM byte
 * counters for the access width of the streaming kernels. */
int ldpc_debug_copy4(const void* src_dev, void* dst_dev, int64_t nbytes, void* stream);

/* One pass of the whole hot path for frames [frame0, frame0+B): channel -> LLR -> decode -> count, everything on
 * the device; counters accumulate as in ldpc_count_errors.  This is the body of `while wec < min_wec`
 * (src/main.py:37-45) for B frames at once. */
int ldpc_simulate(ldpc_decoder_t dec, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id,
                  uint64_t frame0, int64_t B, int32_t max_iter, uint32_t flags, int32_t hist_bins, int64_t* counters_dev,
                  void* stream);

/* `rounds` passes of ldpc_simulate in ONE call: round r decodes frames frame0 + r * round_stride + [0, B) (round_stride >= B: the
 * distance between the rounds of one rank when a round of the whole job is sharded over ranks; = B on a single GPU) and accumulates into
 * its own counter row counters_dev + r * (4 + hist_bins).  Each row is exactly what ldpc_simulate would have produced for that round --
 * the caller applies the stopping rule of `while wec < min_wec` (src/main.py:37) to the rows in order, so the counters stay a function
 * of the round size, not of how many rounds travelled together.  The LDS-resident erasure decoder runs all rounds as ONE launch (its
 * frame positions are refilled across round boundaries, no drain between rounds); every other decoder is called round by round. */
int ldpc_simulate_rounds(ldpc_decoder_t dec, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0,
                         int64_t B, int32_t rounds, uint64_t round_stride, int32_t max_iter, uint32_t flags, int32_t hist_bins,
                         int64_t* counters_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
