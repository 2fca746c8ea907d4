// Shared declarations of the gfx950 LDPC belief-propagation library (libldpc_hip.so).
//
// Data model:
//   * H is flattened once into a row-major edge list (edge k = (chk[k], var[k]), the order of
//     np.where(H) used by the reference, src/bpa.py:12) + CSR row pointers + CSC lists, resident in HBM.
//   * Frames are processed in TILES of 64: one wavefront lane <-> one frame, so every H index is
//     wave-uniform (scalar loads) and every message access is one contiguous 64-element line.
//   * Streaming backend: per tile, check -> variable messages c2v[tile][edge][64] and marginals marg[tile][n][64] live in HBM;
//     the check pass forms v2c = marg - c2v_old on the fly, the variable pass rebuilds the marginals (ldpc_stream.hip).
//   * Fused backend (regular codes whose state fits the LDS): one wavefront owns one frame for all
//     iterations; messages never leave the CU.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ldpc_hip.h"

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace ldpc {

enum Alg : int { ALG_MSA = 0, ALG_SPA = 1, ALG_BEC = 2 };
enum DType : int { DT_F32 = 0, DT_F64 = 1, DT_F16 = 2 };  // DT_F16: fp16 STORAGE of the streaming messages, fp32 arithmetic and priors
enum Backend : int { BK_AUTO = 0, BK_STREAM = 1, BK_FUSED = 2 };
enum Channel : int { CH_BIAWGN = 0, CH_BSC = 1, CH_BEC = 2 };
constexpr int CH_RAW_OBSERVATION = 0x100;  // or-ed into the channel id: BI-AWGN writes y itself instead of the LLR -2y/sigma^2

constexpr int TILE = 64;  // frames per tile == wavefront width on gfx950

// error codes returned over the C ABI come from include/ldpc_hip.h (LDPC_E_*); 0 == success
constexpr int LDPC_OK = 0;

void set_error(const char* fmt, ...);
const char* last_error();

#define LDPC_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ::ldpc::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LDPC_E_HIP;                                                              \
        }                                                                                           \
    } while (0)

#define LDPC_TRY(expr)            \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// Tanner graph resident on one device.
struct Code {
    int device = 0;
    int32_t m = 0, n = 0;
    int64_t E = 0;
    int32_t max_dc = 0, min_dc = 0, max_dv = 0, min_dv = 0;
    // device arrays
    int32_t* d_row_ptr = nullptr;   // [m+1]
    int32_t* d_edge_var = nullptr;  // [E]  variable of edge k (row-major edge order)
    int32_t* d_edge_chk = nullptr;  // [E]
    int32_t* d_col_ptr = nullptr;   // [n+1]
    int32_t* d_col_edge = nullptr;  // [E]  edges of variable v in ascending edge order
    // host mirrors
    std::vector<int32_t> row_ptr, edge_var, edge_chk, col_ptr, col_edge;
};

// host part of ldpc_code_create: validates the edge list and fills the host mirrors (no device call)
int code_build_host(int32_t m, int32_t n, int64_t E, const int32_t* chk, const int32_t* var, Code* c);

// Growable device buffer owned by a decoder (workspace is kept between calls).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t need);
    void release();
};

struct FusedPlan;  // ldpc_fused.hip

struct Decoder {
    Code* code = nullptr;
    int alg = ALG_MSA, dtype = DT_F32, backend = BK_AUTO;
    // streaming workspace
    DevBuf msg, marg, prior, xbits, xera, live, flags, scratch;  // msg = check -> variable messages
    DevBuf msg2, marg2, prior2, xbits2, live2, fmap, fmap2, rbase, rmap;  // second state set + frame maps of the early-termination repack
    // fused backend
    FusedPlan* fused = nullptr;
    // staging used by the *_host entry points
    DevBuf h_in, h_y0, h_out, h_iters;
    void* pinned = nullptr;  // small page-locked host block (polling word, counters)
    // low-latency host path (ldpc_decode_host, a few frames per call -- the reference's one-frame-per-call loop, src/main.py:37-48):
    // page-locked, device-mapped staging the decode kernel reads priors from and writes decisions to DIRECTLY (no copy engine in the
    // path), a private stream and one event recorded right behind the kernel
    void* lat_pin = nullptr;
    size_t lat_bytes = 0;
    hipStream_t lat_stream = nullptr;
    hipEvent_t lat_event = nullptr;
    hipEvent_t after_kernel_event = nullptr;  // set for the duration of a low-latency call
    // optional per-kernel timing with HIP events recorded on the decode stream (bench.py roofline leg)
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[4] = {0, 0, 0, 0};     // [0] check pass, [1] variable pass, [2] fused decode kernel, [3] whole streaming decode
    int64_t prof_launches[4] = {0, 0, 0, 0};
    // statistics of the last decode call
    int last_sweeps = 0;
    int last_backend = BK_STREAM;
    int last_repacks = 0;  // frame repacks of the last streaming decode
    // exact-in-fp32 mode: device counter of guard events (LDPC_FLAG_PRIOR_GRID, ldpc_decoder_grid_violations)
    DevBuf gridviol;
    // ldpc_decode_bits: for the duration of that call the streaming kernels write their decisions here as packed words [B, ceil(n/32)]
    // instead of bytes (null otherwise)
    uint32_t* out_bits = nullptr;
    int chunk_retries = 0;  // how often decode_dev halved the streaming chunk after a failed reservation (ldpc_decoder_chunk_state)
    DevBuf h_bits, h_era;  // packed staging of ldpc_decode_host (decisions, erased mask)
    int64_t stream_chunk = 0;  // frames per pass through the streaming kernels (0: not decided yet; ldpc_api.hip stream_chunk_frames)
};

// event-pair bookkeeping used when Decoder::profile is set
struct ProfSpan {
    int kind;
    hipEvent_t a, b;
};
int prof_event(Decoder* d, size_t idx, hipEvent_t* out);
int prof_collect(Decoder* d, const std::vector<ProfSpan>& spans);

// ---- backends (each returns an LDPC_* code) -------------------------------------------------------
int stream_decode(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                  uint8_t* xhat, int32_t* iters, void* soft_out, hipStream_t st);

// bit-sliced erasure decoder on the streaming kernels (ldpc_bec_stream.hip)
int becs_stream_decode(Decoder* d, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags, uint8_t* xhat, int32_t* iters, hipStream_t st);
int becs_stream_simulate(Decoder* d, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B, int32_t max_iter,
                         uint32_t flags, int32_t hist_bins, int64_t* counters, hipStream_t st);

int stream_simulate_biawgn(Decoder* d, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B,
                           int32_t max_iter, uint32_t flags, uint8_t* xhat, int32_t* iters, hipStream_t st);

int fused_plan_create(Decoder* d);
int fused_plan_host(const Code* c, int alg, int dtype, long moves, const char* out_dir, double* info4);  // host only, no device
void fused_plan_destroy(Decoder* d);
bool fused_supported(const Decoder* d);
bool fused_simulate_supported(const Decoder* d, int channel, double param, int hist_bins);
int fused_simulate(Decoder* d, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B,
                   int32_t max_iter, uint32_t flags, int32_t hist_bins, int64_t* counters, hipStream_t st, int rounds = 1, uint64_t round_stride = 0);
bool fused_simulate_rounds_supported(const Decoder* d);
int fused_info(const Decoder* d, double* out8);
int fused_kernel_name(const Decoder* d, bool sim, char* buf, size_t len);
int fused_decode(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                 uint8_t* xhat, int32_t* iters, void* soft_out, hipStream_t st);

// ---- channel / counting kernels -------------------------------------------------------------------
int channel_generate(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id,
                     uint64_t frame0, int64_t B, int32_t n, void* priors, uint8_t* y, hipStream_t st);
int channel_generate_words(int channel, int dtype, double param, int codeword, const uint8_t* codebook, int64_t K, uint64_t seed,
                           uint64_t stream_id, uint64_t frame0, int64_t B, int32_t n, void* priors, uint8_t* y, uint8_t* sent, hipStream_t st,
                           const unsigned long long* list = nullptr);
int channel_generate_list(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, const uint64_t* list_dev, int64_t cap,
                          int32_t n, void* priors, hipStream_t st);
int count_errors_words(const uint8_t* xhat, const uint8_t* sent, int sent_per_frame, int codeword, const int32_t* iters, int64_t B, int32_t n,
                       int32_t hist_bins, int64_t* counters, hipStream_t st);
int count_errors_list(const uint8_t* xhat, int codeword, const int32_t* iters, const uint64_t* list_dev, int64_t rows, int32_t n, int32_t hist_bins,
                      int64_t* counters, int64_t counter_stride, uint64_t frame_base, uint64_t round_stride, int64_t nrounds, int64_t* redone2,
                      hipStream_t st);
int count_errors(const uint8_t* xhat, const uint8_t* sent, int codeword, const int32_t* iters, int64_t B, int32_t n,
                 int32_t max_iter_hist, int64_t* counters, hipStream_t st);

// ---- maximum-likelihood decoder of the short codes (ldpc_ml.hip) ------------------------------------
struct MlDecoder;
int ml_create(int device, const uint8_t* codebook, int64_t K, int32_t n, MlDecoder** out);
void ml_destroy(MlDecoder* d);
void ml_info(const MlDecoder* d, int64_t* K, int32_t* n, int32_t* W);
int ml_decode(MlDecoder* d, int channel, int dtype, const double* coef, const void* y, int64_t B, const uint32_t* pick,
              int32_t* index, int32_t* ties, uint32_t* tie_mask, double* best, uint8_t* xhat, hipStream_t st);
int ml_simulate(MlDecoder* d, int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id,
                uint64_t frame0, int64_t B, int64_t* counters, hipStream_t st);

// ---- ADMM LP decoder (ldpc_admm.hip) -------------------------------------------------------------
struct AdmmDecoder;
int admm_create(Code* code, AdmmDecoder** out);
void admm_destroy(AdmmDecoder* d);
int admm_last_repacks(const AdmmDecoder* d);
int admm_last_backend(const AdmmDecoder* d);
int admm_decode(AdmmDecoder* d, const double* gamma, int64_t B, double mu, double eps, int32_t max_iter, double* x_out, int32_t* iters,
                uint8_t* converged, hipStream_t st);

// packed decisions: bit (v & 31) of word (v >> 5) of frame row f = decision of variable v; W = ceil(n / 32) words per frame
int pack_bits(const uint8_t* xhat, int64_t B, int32_t n, uint32_t* bits, uint32_t* erased, hipStream_t st);
int count_errors_bits(const uint32_t* bits, const uint32_t* erased, const uint32_t* sent_bits, int codeword, const int32_t* iters, int64_t B,
                      int32_t n, int32_t hist_bins, int64_t* counters, hipStream_t st);

int debug_copy4(const void* src, void* dst, int64_t nbytes, hipStream_t st);

constexpr uint32_t FLAG_NO_EARLY_EXIT = 1u;  // run exactly max_iter sweeps (NOT reference behaviour; benchmarking aid)

}  // namespace ldpc
