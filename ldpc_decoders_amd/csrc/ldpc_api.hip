// extern "C" surface of libldpc_hip.so (declared in include/ldpc_hip.h).
#include <array>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "ldpc_common.hpp"

namespace ldpc {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
const char* last_error() { return g_err.c_str(); }

int DevBuf::reserve(size_t need) {
    if (need <= bytes) return LDPC_OK;
    if (p) {
        LDPC_HIP_TRY(hipFree(p));
        p = nullptr;
        bytes = 0;
    }
    const size_t slack = need / 8 < ((size_t)64 << 20) ? need / 8 : ((size_t)64 << 20);  // growth headroom, bounded: workspaces reach 100 GB
    size_t want = need + slack + 256;
    // fault injection for tests/test_gpu_packed_bits.py (the chunk-halving retry of decode_dev cannot be reached on a 288 GB card without
    // tens of GB of input): LDPC_TEST_FAIL_RESERVE=K makes the K-th growing reservation from then on ask for 2^50 bytes -- a genuine
    // hipMalloc failure, sticky error included
    // (read at every growth, counted from the moment the value changes, so that a test arms it in-process right before the call under test)
    static std::string armed;
    static int grown = 0;
    const char* env = getenv("LDPC_TEST_FAIL_RESERVE");
    if (armed != (env ? env : "")) {
        armed = env ? env : "";
        grown = 0;
    }
    if (env && atoi(env) > 0 && ++grown == atoi(env)) want = (size_t)1 << 50;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        // since ROCm 7 hipGetLastError() returns the last REAL error, not the last call's status: consume this one, or the
        // LDPC_HIP_TRY(hipGetLastError()) behind the next successful launch (the retry with a smaller chunk) would report it again
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return LDPC_E_NOMEM;
    }
    bytes = want;
    return LDPC_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int prof_event(Decoder* d, size_t idx, hipEvent_t* out) {
    while (d->ev_pool.size() <= idx) {
        hipEvent_t e;
        LDPC_HIP_TRY(hipEventCreate(&e));
        d->ev_pool.push_back(e);
    }
    *out = d->ev_pool[idx];
    return LDPC_OK;
}

int prof_collect(Decoder* d, const std::vector<ProfSpan>& spans) {
    for (const ProfSpan& s : spans) {
        float ms = 0.f;
        LDPC_HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        d->prof_ms[s.kind] += ms;
        d->prof_launches[s.kind] += 1;
    }
    return LDPC_OK;
}

namespace {
template <typename T>
int upload(const std::vector<T>& h, T** d) {
    LDPC_HIP_TRY(hipMalloc((void**)d, (h.size() + 1) * sizeof(T)));
    LDPC_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LDPC_OK;
}

// Frames per pass through the streaming kernels: at most 2^17, and no more than fit -- with the second state set of the frame repack
// (three quarters of the first) -- into 80 % of the memory that is free now.  n = 64 800 in fp32 is 1.3 MB of state per frame: 2^17
// frames would want 297 GB with the repack set; 98 304 take 223 GB.
int64_t stream_chunk_frames(Decoder* d) {
    // decided once per decoder (a driver query per call would make the chunking follow the allocator state of the moment); when a later
    // reservation fails all the same -- other allocations took the memory in between -- the caller halves it (decode_dev)
    if (d->stream_chunk > 0) return d->stream_chunk;
    const size_t esz = d->alg == ALG_BEC ? 1 : (d->dtype == DT_F64 ? 8 : 4);
    const double n = (double)d->code->n, E = (double)d->code->E;
    // per frame: messages + marginals + priors, the second state set of the frame repack (three quarters of the first), the decision /
    // erasure bit planes of both sets (n / 8 bytes each), the frame maps (4 B), and the staged priors / decisions / iteration counts
    const double per_frame = 1.75 * (double)esz * (E + 2.0 * n) + 1.75 * 2.0 * n / 8.0 + 8.0 + (double)esz * n + 2.0 * n + 4.0;
    size_t free_b = 0, total_b = 0;
    int64_t step = (int64_t)1 << 17;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
        // what this decoder already holds is re-used, not needed again
        const size_t held = d->msg.bytes + d->marg.bytes + d->prior.bytes + d->msg2.bytes + d->marg2.bytes + d->prior2.bytes;
        const double budget = 0.8 * ((double)free_b + (double)held);
        const int64_t fit = (int64_t)(budget / per_frame) / 64 * 64;
        if (fit < step) step = fit < 64 ? 64 : fit;
    }
    d->stream_chunk = step;
    return step;
}

// Frees the streaming state (both sets): the chunk-halving retry of decode_dev must give back what the failed, larger reservation had
// already taken -- buffers reserved before the one that failed keep their full size otherwise and the smaller chunk frees nothing.
static void release_stream_workspaces(Decoder* d) {
    for (DevBuf* b : {&d->msg, &d->marg, &d->prior, &d->xbits, &d->xera, &d->live, &d->msg2, &d->marg2, &d->prior2, &d->xbits2, &d->live2, &d->fmap,
                      &d->fmap2, &d->rmap})
        b->release();
}

// d->out_bits is per-call state of the streaming kernels (packed decisions straight from the planes): reset on EVERY exit path, an
// exception caught by guarded() included -- a stale pointer would send the next plain ldpc_decode's decisions to freed caller memory.
struct OutBitsScope {
    Decoder* d;
    OutBitsScope(Decoder* dec, uint32_t* bits) : d(dec) { d->out_bits = bits; }
    ~OutBitsScope() { d->out_bits = nullptr; }
};

// LDPC_FLAG_PRIOR_GRID arms the exactness guard of the LDS-resident fp32 min-sum kernels.  The streaming kernels have no such guard:
// a call that asks for it there is refused (as ldpc_simulate refuses it) instead of returning frames nobody vouched for.  fp64 decoders
// need no guard (their arithmetic IS the reference's), the flag is then a no-op.
int grid_guard_available(const Decoder* d, int bk, uint32_t flags, const char* who) {
    if (LDPC_FLAG_PRIOR_GRID_OF(flags) < 0 || d->dtype == DT_F64) return LDPC_OK;
    if (bk == BK_FUSED && d->alg == ALG_MSA) return LDPC_OK;
    set_error("%s: prior grid: the exactness guard lives in the LDS-resident fp32 min-sum kernels; this decoder runs on the streaming kernels", who);
    return LDPC_E_UNSUPPORTED;
}

int pick_backend(Decoder* d) {
    if (d->backend == BK_STREAM) return BK_STREAM;
    if (fused_supported(d)) return BK_FUSED;
    if (d->backend == BK_FUSED) {
        set_error("fused backend does not support this (code, algorithm, dtype); use LDPC_BACKEND_AUTO or _STREAM");
        return LDPC_E_UNSUPPORTED;
    }
    return BK_STREAM;
}
}  // namespace

int code_build_host(int32_t m, int32_t n, int64_t E, const int32_t* chk, const int32_t* var, Code* c) {
    if (!chk || !var || m <= 0 || n <= 0 || E <= 0 || E > (int64_t)1 << 30) {
        set_error("bad graph arguments (m=%d n=%d E=%lld)", m, n, (long long)E);
        return LDPC_E_ARG;
    }
    c->m = m;
    c->n = n;
    c->E = E;
    c->row_ptr.assign((size_t)m + 1, 0);
    c->col_ptr.assign((size_t)n + 1, 0);
    c->edge_var.assign(var, var + E);
    c->edge_chk.assign(chk, chk + E);
    for (int64_t k = 0; k < E; ++k) {
        if (chk[k] < 0 || chk[k] >= m || var[k] < 0 || var[k] >= n) {
            set_error("edge %lld = (%d,%d) outside a %dx%d matrix", (long long)k, chk[k], var[k], m, n);
            return LDPC_E_GRAPH;
        }
        if (k && (chk[k] < chk[k - 1] || (chk[k] == chk[k - 1] && var[k] <= var[k - 1]))) {
            set_error("edge list must be strictly row-major sorted (np.where order); violated at edge %lld", (long long)k);
            return LDPC_E_GRAPH;
        }
        c->row_ptr[chk[k] + 1]++;
        c->col_ptr[var[k] + 1]++;
    }
    c->max_dc = c->max_dv = 0;
    c->min_dc = c->min_dv = INT32_MAX;
    for (int i = 0; i < m; ++i) {
        const int d = c->row_ptr[i + 1];
        c->max_dc = d > c->max_dc ? d : c->max_dc;
        c->min_dc = d < c->min_dc ? d : c->min_dc;
        c->row_ptr[i + 1] += c->row_ptr[i];
    }
    for (int i = 0; i < n; ++i) {
        const int d = c->col_ptr[i + 1];
        c->max_dv = d > c->max_dv ? d : c->max_dv;
        c->min_dv = d < c->min_dv ? d : c->min_dv;
        c->col_ptr[i + 1] += c->col_ptr[i];
    }
    c->col_edge.assign((size_t)E, 0);
    std::vector<int32_t> fill(c->col_ptr.begin(), c->col_ptr.end() - 1);
    for (int64_t k = 0; k < E; ++k) c->col_edge[fill[var[k]]++] = (int32_t)k;  // ascending edge order per variable
    return LDPC_OK;
}
}  // namespace ldpc

using namespace ldpc;

// Nothing throws across the C boundary (include/ldpc_hip.h): every entry point runs inside this guard.  The host side of the library
// works with std::vector / std::string (edge lists, layout plans, plan files), so std::bad_alloc / std::length_error are possible.
template <class F>
static int guarded(const char* who, F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", who);
        return LDPC_E_NOMEM;
    } catch (const std::length_error& e) {
        set_error("%s: size beyond what the host containers hold (%s)", who, e.what());
        return LDPC_E_NOMEM;
    } catch (const std::exception& e) {
        set_error("%s: %s", who, e.what());
        return LDPC_E_ARG;
    } catch (...) {
        set_error("%s: unknown C++ exception", who);
        return LDPC_E_ARG;
    }
}


extern "C" {

const char* ldpc_last_error(void) { return last_error(); }
int ldpc_abi_version(void) { return 4; }

int ldpc_device_count(int* count) {
    return guarded("ldpc_device_count", [&]() -> int {
        if (!count) return LDPC_E_ARG;
        LDPC_HIP_TRY(hipGetDeviceCount(count));
        return LDPC_OK;
    });
}

int ldpc_code_create(int device, int32_t m, int32_t n, int64_t E, const int32_t* chk, const int32_t* var, ldpc_code_t* out) {
    return guarded("ldpc_code_create", [&]() -> int {
        if (!out) {
            set_error("ldpc_code_create: out is null");
            return LDPC_E_ARG;
        }
        std::unique_ptr<Code> holder(new Code());  // (an exception from the containers below unwinds through the guard: nothing leaks)
        Code* c = holder.get();
        c->device = device;
        LDPC_TRY(code_build_host(m, n, E, chk, var, c));
        int rc = LDPC_OK;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            set_error("hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
            return LDPC_E_HIP;
        }
        if ((rc = upload(c->row_ptr, &c->d_row_ptr)) || (rc = upload(c->edge_var, &c->d_edge_var)) ||
            (rc = upload(c->edge_chk, &c->d_edge_chk)) || (rc = upload(c->col_ptr, &c->d_col_ptr)) ||
            (rc = upload(c->col_edge, &c->d_col_edge))) {
            ldpc_code_destroy((ldpc_code_t)holder.release());
            return rc;
        }
        *out = (ldpc_code_t)holder.release();
        return LDPC_OK;
    });
}

int ldpc_code_destroy(ldpc_code_t h) {
    return guarded("ldpc_code_destroy", [&]() -> int {
        Code* c = (Code*)h;
        if (!c) return LDPC_OK;
        (void)hipSetDevice(c->device);
        for (int32_t* p : {c->d_row_ptr, c->d_edge_var, c->d_edge_chk, c->d_col_ptr, c->d_col_edge})
            if (p) (void)hipFree(p);
        delete c;
        return LDPC_OK;
    });
}

int ldpc_code_info(ldpc_code_t h, int32_t* m, int32_t* n, int64_t* E, int32_t* max_dc, int32_t* max_dv) {
    return guarded("ldpc_code_info", [&]() -> int {
        Code* c = (Code*)h;
        if (!c) return LDPC_E_ARG;
        if (m) *m = c->m;
        if (n) *n = c->n;
        if (E) *E = c->E;
        if (max_dc) *max_dc = c->max_dc;
        if (max_dv) *max_dv = c->max_dv;
        return LDPC_OK;
    });
}

int ldpc_plan_layout(int32_t m, int32_t n, int64_t E, const int32_t* chk, const int32_t* var, int alg, int dtype, int64_t moves,
                     const char* out_dir, double* info4) {
    return guarded("ldpc_plan_layout", [&]() -> int {
        if (!info4 || alg < 0 || alg > 2 || dtype < 0 || dtype > 1) {
            set_error("ldpc_plan_layout: bad arguments (alg=%d dtype=%d)", alg, dtype);
            return LDPC_E_ARG;
        }
        Code c;
        LDPC_TRY(code_build_host(m, n, E, chk, var, &c));
        return fused_plan_host(&c, alg, dtype, (long)moves, out_dir, info4);
    });
}

int ldpc_decoder_create(ldpc_code_t code, int alg, int dtype, int backend, ldpc_decoder_t* out) {
    return guarded("ldpc_decoder_create", [&]() -> int {
        if (!code || !out || alg < 0 || alg > 2 || dtype < 0 || dtype > 2 || (dtype == DT_F16 && (alg == ALG_BEC || backend == BK_FUSED)) ||
            backend < 0 || backend > 2) {
            set_error("ldpc_decoder_create: bad arguments (alg=%d dtype=%d backend=%d)", alg, dtype, backend);
            return LDPC_E_ARG;
        }
        Decoder* d = new (std::nothrow) Decoder();
        if (!d) return LDPC_E_NOMEM;
        d->code = (Code*)code;
        d->alg = alg;
        d->dtype = dtype;
        d->backend = backend;
        (void)hipSetDevice(d->code->device);
        hipError_t e = hipHostMalloc(&d->pinned, 4096, hipHostMallocDefault);
        if (e != hipSuccess) {
            set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
            delete d;
            return LDPC_E_HIP;
        }
        int rc = LDPC_OK;
        try {
            rc = fused_plan_create(d);  // host containers (layout planner, table builders): may throw
        } catch (...) {
            ldpc_decoder_destroy((ldpc_decoder_t)d);
            throw;  // translated by the guard
        }
        if (rc) {
            ldpc_decoder_destroy((ldpc_decoder_t)d);
            return rc;
        }
        if (backend == BK_FUSED && !fused_supported(d)) {
            set_error("fused backend does not support this (code, algorithm, dtype)");
            ldpc_decoder_destroy((ldpc_decoder_t)d);
            return LDPC_E_UNSUPPORTED;
        }
        *out = (ldpc_decoder_t)d;
        return LDPC_OK;
    });
}

int ldpc_decoder_destroy(ldpc_decoder_t h) {
    return guarded("ldpc_decoder_destroy", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d) return LDPC_OK;
        (void)hipSetDevice(d->code->device);
        fused_plan_destroy(d);
        for (DevBuf* b : {&d->msg, &d->marg, &d->marg2, &d->prior, &d->xbits, &d->xera, &d->live, &d->flags, &d->scratch, &d->gridviol, &d->msg2, &d->prior2, &d->xbits2, &d->live2, &d->fmap, &d->fmap2, &d->rbase, &d->rmap, &d->h_in, &d->h_y0, &d->h_out,
                          &d->h_iters, &d->h_bits, &d->h_era})
            b->release();
        if (d->pinned) (void)hipHostFree(d->pinned);
        if (d->lat_pin) (void)hipHostFree(d->lat_pin);
        if (d->lat_event) (void)hipEventDestroy(d->lat_event);
        if (d->lat_stream) (void)hipStreamDestroy(d->lat_stream);
        for (hipEvent_t e : d->ev_pool) (void)hipEventDestroy(e);
        delete d;
        return LDPC_OK;
    });
}

int ldpc_decoder_chunk_state(ldpc_decoder_t h, int64_t* chunk_frames, int* retries) {
    return guarded("ldpc_decoder_chunk_state", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !chunk_frames || !retries) return LDPC_E_ARG;
        *chunk_frames = d->stream_chunk;
        *retries = d->chunk_retries;
        return LDPC_OK;
    });
}

int ldpc_decoder_last_repacks(ldpc_decoder_t h, int* repacks) {
    return guarded("ldpc_decoder_last_repacks", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !repacks) return LDPC_E_ARG;
        *repacks = d->last_backend == BK_STREAM ? d->last_repacks : 0;
        return LDPC_OK;
    });
}

int ldpc_decoder_grid_violations(ldpc_decoder_t h, int64_t* count, int64_t* frames, int64_t cap, int reset) {
    return guarded("ldpc_decoder_grid_violations", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !count || cap < 0 || (cap > 0 && !frames)) return LDPC_E_ARG;
        *count = 0;
        if (!d->gridviol.p) return LDPC_OK;  // no call with LDPC_FLAG_PRIOR_GRID yet
        LDPC_HIP_TRY(hipSetDevice(d->code->device));
        LDPC_HIP_TRY(hipDeviceSynchronize());
        LDPC_HIP_TRY(hipMemcpy(count, d->gridviol.p, 8, hipMemcpyDeviceToHost));
        const int64_t listed = *count < 4095 ? *count : 4095;  // GRID_REDO_CAP of the kernels
        const int64_t take = listed < cap ? listed : cap;
        if (take > 0) LDPC_HIP_TRY(hipMemcpy(frames, (const char*)d->gridviol.p + 8, (size_t)take * 8, hipMemcpyDeviceToHost));
        if (reset) LDPC_HIP_TRY(hipMemset(d->gridviol.p, 0, 8));
        return LDPC_OK;
    });
}

int ldpc_decoder_grid_list(ldpc_decoder_t h, uint64_t** list_dev, int64_t* cap, void* stream) {
    return guarded("ldpc_decoder_grid_list", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !list_dev) return LDPC_E_ARG;
        LDPC_HIP_TRY(hipSetDevice(d->code->device));
        if (!d->gridviol.p) {  // as the first guarded launch would: count + list, zeroed
            LDPC_TRY(d->gridviol.reserve((size_t)(1 + 4095) * 8));
            LDPC_HIP_TRY(hipMemsetAsync(d->gridviol.p, 0, (size_t)(1 + 4095) * 8, (hipStream_t)stream));
        }
        *list_dev = (uint64_t*)d->gridviol.p;
        if (cap) *cap = 4095;
        return LDPC_OK;
    });
}

int ldpc_decoder_grid_list_reset(ldpc_decoder_t h, void* stream) {
    return guarded("ldpc_decoder_grid_list_reset", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d) return LDPC_E_ARG;
        if (!d->gridviol.p) return LDPC_OK;
        LDPC_HIP_TRY(hipSetDevice(d->code->device));
        LDPC_HIP_TRY(hipMemsetAsync(d->gridviol.p, 0, 8, (hipStream_t)stream));
        return LDPC_OK;
    });
}

int ldpc_channel_list(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, const uint64_t* list_dev, int64_t cap,
                      int32_t n, void* priors, void* stream) {
    return guarded("ldpc_channel_list", [&]() -> int {
        if (cap < 0 || n <= 0 || dtype < 0 || dtype > 1 || !priors || !list_dev) {
            set_error("ldpc_channel_list: bad arguments");
            return LDPC_E_ARG;
        }
        return channel_generate_list(channel, dtype, param, codeword, seed, stream_id, list_dev, cap, n, priors, (hipStream_t)stream);
    });
}

int ldpc_count_errors_list(const uint8_t* xhat, int codeword, const int32_t* iters, const uint64_t* list_dev, int64_t rows, int32_t n,
                           int32_t hist_bins, int64_t* counters, int64_t counter_stride, uint64_t frame_base, uint64_t round_stride, int64_t nrounds,
                           int64_t* redone2, void* stream) {
    return guarded("ldpc_count_errors_list", [&]() -> int {
        if (!xhat || !iters || !counters || !list_dev || !redone2 || rows < 0 || n <= 0 || hist_bins < 0 || nrounds < 1 || counter_stride < 4 + hist_bins) {
            set_error("ldpc_count_errors_list: bad arguments");
            return LDPC_E_ARG;
        }
        return count_errors_list(xhat, codeword, iters, list_dev, rows, n, hist_bins, counters, counter_stride, frame_base, round_stride, nrounds, redone2,
                                 (hipStream_t)stream);
    });
}

int ldpc_decoder_last_stats(ldpc_decoder_t h, int* backend, int* sweeps) {
    return guarded("ldpc_decoder_last_stats", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d) return LDPC_E_ARG;
        if (backend) *backend = d->last_backend;
        if (sweeps) *sweeps = d->last_sweeps;
        return LDPC_OK;
    });
}

int ldpc_decoder_fused_info(ldpc_decoder_t h, double* out8) {
    return guarded("ldpc_decoder_fused_info", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !out8) return LDPC_E_ARG;
        return fused_info(d, out8);
    });
}

int ldpc_decoder_kernel_name(ldpc_decoder_t h, int simulate, char* buf, int64_t len) {
    return guarded("ldpc_decoder_kernel_name", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !buf || len <= 0) return LDPC_E_ARG;
        buf[0] = 0;
        if (d->backend == BK_STREAM) return LDPC_OK;  // a decoder pinned to the streaming kernels never launches its LDS-resident shape
        return fused_kernel_name(d, simulate != 0, buf, (size_t)len);
    });
}

int ldpc_decoder_profile(ldpc_decoder_t h, int enable) {
    return guarded("ldpc_decoder_profile", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d) return LDPC_E_ARG;
        d->profile = enable != 0;
        return LDPC_OK;
    });
}

int ldpc_decoder_profile_read(ldpc_decoder_t h, double* ms, int64_t* launches, int reset) {
    return guarded("ldpc_decoder_profile_read", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d) return LDPC_E_ARG;
        for (int i = 0; i < 4; ++i) {
            if (ms) ms[i] = d->prof_ms[i];
            if (launches) launches[i] = d->prof_launches[i];
            if (reset) {
                d->prof_ms[i] = 0;
                d->prof_launches[i] = 0;
            }
        }
        return LDPC_OK;
    });
}

// Body of ldpc_decode / ldpc_decode_bits: decisions as bytes [B,n] (xhat) or as packed words [B,W] (bits; erased: the erasure
// decoder's "still erased" mask).  The LLR decoders on the streaming kernels write the words straight from their decision planes; the
// LDS-resident kernels and the erasure decoder produce bytes in the decoder's staging buffer, packed by one more kernel.
static int decode_dev(Decoder* d, const char* who, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags, uint8_t* xhat,
                      uint32_t* bits, uint32_t* erased, int32_t* iters, hipStream_t st) {
    LDPC_HIP_TRY(hipSetDevice(d->code->device));
    const int bk = pick_backend(d);
    if (bk < 0) return bk;
    LDPC_TRY(grid_guard_available(d, bk, flags, who));
    // bound the workspace: at most 2^17 frames per pass through a backend, fewer where the streaming state would not fit the HBM
    int64_t step = bk == BK_STREAM ? stream_chunk_frames(d) : (int64_t)1 << 17;
    const size_t esz = d->dtype == DT_F64 ? 8 : 4, n = (size_t)d->code->n, W = (n + 31) / 32;
    const bool planes_direct = bits && bk == BK_STREAM && d->alg != ALG_BEC;
    if (bits && !planes_direct) LDPC_TRY(d->h_out.reserve((size_t)(B < step ? B : step) * n));
    int sweeps = 0;
    for (int64_t b0 = 0; b0 < B;) {
        const int64_t nb = (B - b0) < step ? (B - b0) : step;
        const void* p = priors ? (const char*)priors + (size_t)b0 * n * esz : nullptr;
        const uint8_t* y = y0 ? y0 + (size_t)b0 * n : nullptr;
        uint8_t* xh = bits ? (planes_direct ? nullptr : (uint8_t*)d->h_out.p) : xhat + (size_t)b0 * n;
        int rc;
        {
            OutBitsScope scope(d, planes_direct ? bits + (size_t)b0 * W : nullptr);
            rc = bk == BK_FUSED ? fused_decode(d, p, y, nb, max_iter, flags, xh, iters + b0, nullptr, st)
                                : stream_decode(d, p, y, nb, max_iter, flags, xh, iters + b0, nullptr, st);
        }
        if (rc == LDPC_E_NOMEM && bk == BK_STREAM && nb > 64) {
            // the chunk was sized from the memory that was free when this decoder first asked (stream_chunk_frames); other allocations
            // since (a twin decoder, torch tensors) may have taken it: halve the chunk -- for this decoder's lifetime -- give back what
            // the failed attempt had reserved, and try again
            step = ((nb / 2 + 63) / 64) * 64;
            d->stream_chunk = step;
            LDPC_HIP_TRY(hipStreamSynchronize(st));
            release_stream_workspaces(d);
            ++d->chunk_retries;
            continue;
        }
        if (rc) return rc;
        if (bits && !planes_direct) LDPC_TRY(pack_bits(xh, nb, (int32_t)n, bits + (size_t)b0 * W, erased ? erased + (size_t)b0 * W : nullptr, st));
        sweeps = d->last_sweeps > sweeps ? d->last_sweeps : sweeps;
        b0 += nb;
    }
    if (planes_direct && erased) LDPC_HIP_TRY(hipMemsetAsync(erased, 0, (size_t)B * W * 4, st));  // an LLR decoder never leaves a bit erased
    d->last_sweeps = sweeps;
    return LDPC_OK;
}

int ldpc_decode(ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                uint8_t* xhat, int32_t* iters, void* stream) {
    return guarded("ldpc_decode", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !xhat || !iters || B < 0) {
            set_error("ldpc_decode: bad arguments");
            return LDPC_E_ARG;
        }
        return decode_dev(d, "ldpc_decode", priors, y0, B, max_iter, flags, xhat, nullptr, nullptr, iters, (hipStream_t)stream);
    });
}

int ldpc_decode_bits(ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                     uint32_t* xhat_bits, uint32_t* erased_bits, int32_t* iters, void* stream) {
    return guarded("ldpc_decode_bits", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !xhat_bits || !iters || B < 0) {
            set_error("ldpc_decode_bits: bad arguments");
            return LDPC_E_ARG;
        }
        if (d->alg == ALG_BEC && !erased_bits) {
            set_error("ldpc_decode_bits: the erasure decoder needs erased_bits (a bit it could not resolve is neither 0 nor 1)");
            return LDPC_E_ARG;
        }
        return decode_dev(d, "ldpc_decode_bits", priors, y0, B, max_iter, flags, nullptr, xhat_bits, erased_bits, iters, (hipStream_t)stream);
    });
}

int ldpc_decode_soft(ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                     uint8_t* xhat, int32_t* iters, void* marginals, void* stream) {
    return guarded("ldpc_decode_soft", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !xhat || !iters || !marginals || B < 0 || B > ((int64_t)1 << 17) || d->alg == ALG_BEC) {
            set_error("ldpc_decode_soft: bad arguments (LLR decoders only, at most 2^17 frames per call)");
            return LDPC_E_ARG;
        }
        LDPC_HIP_TRY(hipSetDevice(d->code->device));
        const int bk = pick_backend(d);
        if (bk < 0) return bk;
        LDPC_TRY(grid_guard_available(d, bk, flags, "ldpc_decode_soft"));
        if (bk == BK_FUSED) return fused_decode(d, priors, y0, B, max_iter, flags, xhat, iters, marginals, (hipStream_t)stream);
        return stream_decode(d, priors, y0, B, max_iter, flags, xhat, iters, marginals, (hipStream_t)stream);
    });
}

// packed words -> bytes on the host: 8 decisions per table look-up, rows split over a few threads for large batches
static void unpack_bits_host(const uint32_t* bits, const uint32_t* era, int64_t B, size_t n, uint8_t* xhat) {
    static const std::array<uint64_t, 256> lut = [] {
        std::array<uint64_t, 256> t{};
        for (int b = 0; b < 256; ++b)
            for (int k = 0; k < 8; ++k) t[b] |= (uint64_t)((b >> k) & 1) << (8 * k);
        return t;
    }();
    const size_t W = (n + 31) / 32;
    auto rows = [&](int64_t f0, int64_t f1) {
        for (int64_t f = f0; f < f1; ++f) {
            const uint8_t* src = (const uint8_t*)(bits + (size_t)f * W);
            const uint8_t* se = era ? (const uint8_t*)(era + (size_t)f * W) : nullptr;
            uint8_t* dst = xhat + (size_t)f * n;
            size_t v = 0;
            for (; v + 8 <= n; v += 8) {
                uint64_t w = lut[src[v >> 3]];
                if (se) {
                    const uint64_t e = lut[se[v >> 3]];
                    w = (w & ~e) | (e << 1);  // erased -> symbol 2
                }
                memcpy(dst + v, &w, 8);
            }
            for (; v < n; ++v) dst[v] = se && ((se[v >> 3] >> (v & 7)) & 1) ? 2 : (uint8_t)((src[v >> 3] >> (v & 7)) & 1);
        }
    };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (size_t)B * n < ((size_t)1 << 22) ? 1 : (int)(hw < 2 ? 1 : (hw > 8 ? 8 : hw));
    if (nt == 1) {
        rows(0, B);
        return;
    }
    std::vector<std::thread> pool;
    int started = 0;
    try {
        pool.reserve((size_t)nt);
        for (; started < nt - 1; ++started) pool.emplace_back(rows, B * started / nt, B * (started + 1) / nt);
    } catch (...) {  // no more threads to be had: the rows nobody took are expanded on this one (a joinable thread must never be destroyed)
    }
    rows(B * started / nt, B);
    for (std::thread& t : pool) t.join();
}

// host buffers in, decode on the device, PACKED decisions back over PCIe (n / 8 bytes per frame instead of n); xhat (bytes) and / or
// xhat_bits + erased_bits (packed) are filled on the host side
static int decode_host_impl(Decoder* d, ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                            uint8_t* xhat, uint32_t* out_bits, uint32_t* out_era, int32_t* iters) {
    if (B == 0) return LDPC_OK;
    LDPC_HIP_TRY(hipSetDevice(d->code->device));
    const size_t n = (size_t)d->code->n, esz = d->dtype == DT_F64 ? 8 : 4, W = (n + 31) / 32;
    void* dp = nullptr;
    uint8_t* dy = nullptr;
    if (d->alg != ALG_BEC && !priors) {
        set_error("ldpc_decode_host: priors is null");
        return LDPC_E_ARG;
    }
    if (d->alg == ALG_BEC && !y0) {
        set_error("erasure decoder needs the received symbols (y0)");
        return LDPC_E_ARG;
    }
    // A few frames on the LDS-resident kernels: ONE kernel launch and one event wait.  The kernel reads the priors from, and writes
    // the decisions to, page-locked host memory mapped into the device (a frame is 5-10 KB: a few microseconds over PCIe, less than
    // starting a copy engine twice); no allocation, no memset in front of the launch, no device-wide synchronisation.
    const size_t in_bytes = (size_t)B * n * esz, y_bytes = (size_t)B * n;
    if (xhat && B <= 64 && in_bytes <= ((size_t)256 << 10) && pick_backend(d) == BK_FUSED && !d->profile) {
        const size_t off_y = (in_bytes + 255) & ~(size_t)255, off_out = off_y + ((y_bytes + 255) & ~(size_t)255);
        const size_t off_it = off_out + ((y_bytes + 255) & ~(size_t)255), need = off_it + 64 * sizeof(int32_t);
        if (d->lat_bytes < need) {
            if (d->lat_pin) (void)hipHostFree(d->lat_pin);
            d->lat_pin = nullptr;
            d->lat_bytes = 0;
            LDPC_HIP_TRY(hipHostMalloc(&d->lat_pin, need, hipHostMallocMapped));
            d->lat_bytes = need;
        }
        if (!d->lat_stream) LDPC_HIP_TRY(hipStreamCreateWithFlags(&d->lat_stream, hipStreamNonBlocking));
        if (!d->lat_event) LDPC_HIP_TRY(hipEventCreateWithFlags(&d->lat_event, hipEventDisableTiming));
        char* hp = (char*)d->lat_pin;
        void* devp = nullptr;
        LDPC_HIP_TRY(hipHostGetDevicePointer(&devp, d->lat_pin, 0));
        char* gp = (char*)devp;
        if (d->alg != ALG_BEC) memcpy(hp, priors, in_bytes);
        if (y0) memcpy(hp + off_y, y0, y_bytes);
        d->after_kernel_event = d->lat_event;
        const int rc = fused_decode(d, d->alg == ALG_BEC ? nullptr : gp, y0 ? (const uint8_t*)(gp + off_y) : nullptr, B, max_iter, flags,
                                    (uint8_t*)(gp + off_out), (int32_t*)(gp + off_it), nullptr, d->lat_stream);
        d->after_kernel_event = nullptr;
        if (rc) return rc;
        LDPC_HIP_TRY(hipEventSynchronize(d->lat_event));
        memcpy(xhat, hp + off_out, y_bytes);
        memcpy(iters, hp + off_it, (size_t)B * sizeof(int32_t));
        return LDPC_OK;
    }
    if (d->alg != ALG_BEC) {
        LDPC_TRY(d->h_in.reserve((size_t)B * n * esz));
        dp = d->h_in.p;
        LDPC_HIP_TRY(hipMemcpyAsync(dp, priors, (size_t)B * n * esz, hipMemcpyHostToDevice, nullptr));
    }
    if (y0) {
        LDPC_TRY(d->h_y0.reserve((size_t)B * n));
        dy = (uint8_t*)d->h_y0.p;
        LDPC_HIP_TRY(hipMemcpyAsync(dy, y0, (size_t)B * n, hipMemcpyHostToDevice, nullptr));
    }
    const bool era = d->alg == ALG_BEC;
    LDPC_TRY(d->h_bits.reserve((size_t)B * W * 4));
    if (era) LDPC_TRY(d->h_era.reserve((size_t)B * W * 4));
    LDPC_TRY(d->h_iters.reserve((size_t)B * sizeof(int32_t)));
    LDPC_TRY(ldpc_decode_bits(h, dp, dy, B, max_iter, flags, (uint32_t*)d->h_bits.p, era ? (uint32_t*)d->h_era.p : nullptr, (int32_t*)d->h_iters.p, nullptr));
    // packed words land in the caller's buffer if there is one, otherwise in a scratch vector that is unpacked into xhat
    std::vector<uint32_t> tmp_bits, tmp_era;
    uint32_t* hb = out_bits;
    uint32_t* he = out_era;
    if (!hb) {
        tmp_bits.resize((size_t)B * W);
        hb = tmp_bits.data();
    }
    if (era && !he) {
        tmp_era.resize((size_t)B * W);
        he = tmp_era.data();
    }
    LDPC_HIP_TRY(hipMemcpyAsync(hb, d->h_bits.p, (size_t)B * W * 4, hipMemcpyDeviceToHost, nullptr));
    if (era) LDPC_HIP_TRY(hipMemcpyAsync(he, d->h_era.p, (size_t)B * W * 4, hipMemcpyDeviceToHost, nullptr));
    else if (he) memset(he, 0, (size_t)B * W * 4);
    LDPC_HIP_TRY(hipMemcpyAsync(iters, d->h_iters.p, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, nullptr));
    LDPC_HIP_TRY(hipStreamSynchronize(nullptr));
    if (xhat) unpack_bits_host(hb, era ? he : nullptr, B, n, xhat);
    return LDPC_OK;
}

int ldpc_decode_host(ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                     uint8_t* xhat, int32_t* iters) {
    return guarded("ldpc_decode_host", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !xhat || !iters || B < 0) {
            set_error("ldpc_decode_host: bad arguments");
            return LDPC_E_ARG;
        }
        return decode_host_impl(d, h, priors, y0, B, max_iter, flags, xhat, nullptr, nullptr, iters);
    });
}

int ldpc_decode_host_bits(ldpc_decoder_t h, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                          uint32_t* xhat_bits, uint32_t* erased_bits, int32_t* iters) {
    return guarded("ldpc_decode_host_bits", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !xhat_bits || !iters || B < 0 || (d->alg == ALG_BEC && !erased_bits)) {
            set_error("ldpc_decode_host_bits: bad arguments (the erasure decoder needs erased_bits)");
            return LDPC_E_ARG;
        }
        return decode_host_impl(d, h, priors, y0, B, max_iter, flags, nullptr, xhat_bits, erased_bits, iters);
    });
}

int ldpc_channel(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0,
                 int64_t B, int32_t n, void* priors, uint8_t* y, void* stream) {
    return guarded("ldpc_channel", [&]() -> int {
        if (B < 0 || n <= 0 || dtype < 0 || dtype > 1) {
            set_error("ldpc_channel: bad arguments");
            return LDPC_E_ARG;
        }
        return channel_generate(channel, dtype, param, codeword, seed, stream_id, frame0, B, n, priors, y, (hipStream_t)stream);
    });
}

int ldpc_channel_words(int channel, int dtype, double param, const uint8_t* codebook, int64_t K, uint64_t seed, uint64_t stream_id,
                       uint64_t frame0, int64_t B, int32_t n, void* priors, uint8_t* y, uint8_t* sent, void* stream) {
    return guarded("ldpc_channel_words", [&]() -> int {
        if (B < 0 || n <= 0 || dtype < 0 || dtype > 1 || !codebook || !sent) {
            set_error("ldpc_channel_words: bad arguments");
            return LDPC_E_ARG;
        }
        return channel_generate_words(channel, dtype, param, 0, codebook, K, seed, stream_id, frame0, B, n, priors, y, sent, (hipStream_t)stream);
    });
}

int ldpc_count_errors_words(const uint8_t* xhat, const uint8_t* sent, const int32_t* iters, int64_t B, int32_t n, int32_t hist_bins,
                            int64_t* counters, void* stream) {
    return guarded("ldpc_count_errors_words", [&]() -> int {
        if (!xhat || !sent || !counters || B < 0 || n <= 0 || hist_bins < 0) {
            set_error("ldpc_count_errors_words: bad arguments");
            return LDPC_E_ARG;
        }
        return count_errors_words(xhat, sent, 1, 0, iters, B, n, hist_bins, counters, (hipStream_t)stream);
    });
}

int ldpc_debug_copy4(const void* src_dev, void* dst_dev, int64_t nbytes, void* stream) {
    return guarded("ldpc_debug_copy4", [&]() -> int {
        if (!src_dev || !dst_dev || nbytes < 0) return LDPC_E_ARG;
        return debug_copy4(src_dev, dst_dev, nbytes, (hipStream_t)stream);
    });
}

int ldpc_count_errors(const uint8_t* xhat, const uint8_t* sent, int codeword, const int32_t* iters, int64_t B, int32_t n,
                      int32_t hist_bins, int64_t* counters, void* stream) {
    return guarded("ldpc_count_errors", [&]() -> int {
        if (!xhat || !counters || B < 0 || n <= 0 || hist_bins < 0) {
            set_error("ldpc_count_errors: bad arguments");
            return LDPC_E_ARG;
        }
        return count_errors(xhat, sent, codeword, iters, B, n, hist_bins, counters, (hipStream_t)stream);
    });
}

int ldpc_count_errors_bits(const uint32_t* xhat_bits, const uint32_t* erased_bits, const uint32_t* sent_bits, int codeword, const int32_t* iters,
                           int64_t B, int32_t n, int32_t hist_bins, int64_t* counters, void* stream) {
    return guarded("ldpc_count_errors_bits", [&]() -> int {
        if (!xhat_bits || !counters || B < 0 || n <= 0 || hist_bins < 0) {
            set_error("ldpc_count_errors_bits: bad arguments");
            return LDPC_E_ARG;
        }
        return count_errors_bits(xhat_bits, erased_bits, sent_bits, codeword, iters, B, n, hist_bins, counters, (hipStream_t)stream);
    });
}

static int simulate_impl(ldpc_decoder_t h, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id,
                         uint64_t frame0, int64_t B, int32_t max_iter, uint32_t flags, int32_t hist_bins, int64_t* counters,
                         void* stream) {
    {
        Decoder* d = (Decoder*)h;
        if (!d || !counters || B < 0) {
            set_error("ldpc_simulate: bad arguments");
            return LDPC_E_ARG;
        }
        if ((channel == CH_BEC) != (d->alg == ALG_BEC)) {
            set_error("ldpc_simulate: the erasure channel pairs with LDPC_ALG_BEC decoders (and only with them)");
            return LDPC_E_ARG;
        }
        if (B == 0) return LDPC_OK;
        LDPC_HIP_TRY(hipSetDevice(d->code->device));
        const size_t n = (size_t)d->code->n, esz = d->dtype == DT_F64 ? 8 : 4;
        hipStream_t st = (hipStream_t)stream;
        if (codeword != 0 && codeword != 1) {
            set_error("device channel kernels send the all-zero (0) or all-one (1) word; got codeword=%d", codeword);
            return LDPC_E_ARG;
        }
        const int grid_k = LDPC_FLAG_PRIOR_GRID_OF(flags);
        // (fp64 decoders with a prior grid take the composed path below: quantised priors from the channel kernel, no guard needed)
        if (d->backend != BK_STREAM && fused_simulate_supported(d, channel, param, hist_bins) && !(grid_k >= 0 && d->dtype == DT_F64 && d->alg != ALG_BEC))
            return fused_simulate(d, channel, param, codeword, seed, stream_id, frame0, B, max_iter, flags, hist_bins, counters, st);
        if (grid_k >= 0 && (d->dtype != DT_F64 || channel == CH_BEC)) {
            set_error("prior grid: the exactness guard lives in the LDS-resident fp32 min-sum kernels; this decoder runs on the streaming kernels");
            return LDPC_E_UNSUPPORTED;
        }
        const int ch_grid = grid_k >= 0 ? LDPC_CH_PRIOR_GRID(grid_k) : 0;  // fp64 decoders: quantised priors, no guard needed
        // bounded staging: priors for at most 2^17 frames at a time (fewer where the streaming state would not fit the HBM)
        const int64_t step = pick_backend(d) == BK_STREAM ? stream_chunk_frames(d) : (int64_t)1 << 17;
        const int64_t cap = B < step ? B : step;
        // BI-AWGN on the streaming kernels: the noise is generated straight into the tile layout (no [B,n] prior array, no transposing load)
        const bool tiled_noise = channel == CH_BIAWGN && d->alg != ALG_BEC && pick_backend(d) == BK_STREAM && grid_k < 0;
        // erasure decoder on the streaming kernels: received word drawn into the bit planes, decisions counted from them (no [B,n] bytes at all)
        if (d->alg == ALG_BEC && pick_backend(d) == BK_STREAM) {
            for (int64_t b0 = 0; b0 < B; b0 += step) {
                const int64_t nb = (B - b0) < step ? (B - b0) : step;
                LDPC_TRY(becs_stream_simulate(d, param, codeword, seed, stream_id, frame0 + (uint64_t)b0, nb, max_iter, flags, hist_bins, counters, st));
            }
            return LDPC_OK;
        }
        if (channel != CH_BEC && !tiled_noise) LDPC_TRY(d->h_in.reserve((size_t)cap * n * esz));
        if (channel != CH_BIAWGN) LDPC_TRY(d->h_y0.reserve((size_t)cap * n));
        if (tiled_noise) LDPC_TRY(d->h_bits.reserve((size_t)cap * ((n + 31) / 32) * 4));  // decisions leave the planes as packed words: no [B,n] bytes
        else LDPC_TRY(d->h_out.reserve((size_t)cap * n));
        LDPC_TRY(d->h_iters.reserve((size_t)cap * sizeof(int32_t)));
        for (int64_t b0 = 0; b0 < B; b0 += step) {
            const int64_t nb = (B - b0) < step ? (B - b0) : step;
            if (tiled_noise) {
                int rc;
                {
                    OutBitsScope scope(d, (uint32_t*)d->h_bits.p);
                    rc = stream_simulate_biawgn(d, param, codeword, seed, stream_id, frame0 + (uint64_t)b0, nb, max_iter, flags, nullptr,
                                                (int32_t*)d->h_iters.p, st);
                }
                if (rc) return rc;
                LDPC_TRY(count_errors_bits((const uint32_t*)d->h_bits.p, nullptr, nullptr, codeword, (int32_t*)d->h_iters.p, nb, (int32_t)n, hist_bins, counters, st));
                continue;
            }
            void* pri = channel == CH_BEC ? nullptr : d->h_in.p;
            uint8_t* y = channel == CH_BIAWGN ? nullptr : (uint8_t*)d->h_y0.p;
            LDPC_TRY(channel_generate(channel | ch_grid, d->dtype == DT_F16 ? DT_F32 : d->dtype, param, codeword, seed, stream_id, frame0 + (uint64_t)b0, nb, (int32_t)n, pri,
                                      y, st));
            LDPC_TRY(ldpc_decode(h, pri, y, nb, max_iter, flags, (uint8_t*)d->h_out.p, (int32_t*)d->h_iters.p, stream));
            LDPC_TRY(count_errors((uint8_t*)d->h_out.p, nullptr, codeword, (int32_t*)d->h_iters.p, nb, (int32_t)n, hist_bins, counters,
                                  st));
        }
        return LDPC_OK;
    }
}

int ldpc_simulate(ldpc_decoder_t h, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id,
                  uint64_t frame0, int64_t B, int32_t max_iter, uint32_t flags, int32_t hist_bins, int64_t* counters,
                  void* stream) {
    return guarded("ldpc_simulate", [&]() -> int {
        return simulate_impl(h, channel, param, codeword, seed, stream_id, frame0, B, max_iter, flags, hist_bins, counters, stream);
    });
}

int ldpc_simulate_rounds(ldpc_decoder_t h, int channel, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0,
                         int64_t B, int32_t rounds, uint64_t round_stride, int32_t max_iter, uint32_t flags, int32_t hist_bins,
                         int64_t* counters, void* stream) {
    return guarded("ldpc_simulate_rounds", [&]() -> int {
        Decoder* d = (Decoder*)h;
        if (!d || !counters || B < 0 || rounds < 0 || hist_bins < 0 || (rounds > 1 && round_stride < (uint64_t)B)) {
            set_error("ldpc_simulate_rounds: bad arguments (rounds must not overlap: round_stride >= B)");
            return LDPC_E_ARG;
        }
        if (B == 0 || rounds == 0) return LDPC_OK;
        // one launch for all rounds where the kernel refills its frame positions across round boundaries (the erasure Monte-Carlo kernel)
        if (rounds > 1 && channel == CH_BEC && d->alg == ALG_BEC && d->backend != BK_STREAM && (codeword == 0 || codeword == 1) &&
            fused_simulate_rounds_supported(d) && fused_simulate_supported(d, channel, param, hist_bins) && LDPC_FLAG_PRIOR_GRID_OF(flags) < 0 &&
            (B + 31) / 32 * (int64_t)rounds < ((int64_t)1 << 31)) {
            LDPC_HIP_TRY(hipSetDevice(d->code->device));
            return fused_simulate(d, channel, param, codeword, seed, stream_id, frame0, B, max_iter, flags, hist_bins, counters, (hipStream_t)stream,
                                  rounds, round_stride);
        }
        for (int32_t r = 0; r < rounds; ++r)  // every other decoder: round by round
            LDPC_TRY(simulate_impl(h, channel, param, codeword, seed, stream_id, frame0 + (uint64_t)r * round_stride, B, max_iter, flags, hist_bins,
                                   counters + (size_t)r * (size_t)(4 + hist_bins), stream));
        return LDPC_OK;
    });
}

// ---- maximum-likelihood decoder (ldpc_ml.hip) ----
int ldpc_ml_create(int device, const uint8_t* codebook, int64_t K, int32_t n, ldpc_ml_t* out) {
    return guarded("ldpc_ml_create", [&]() -> int {
        MlDecoder* d = nullptr;
        LDPC_TRY(ml_create(device, codebook, K, n, &d));
        *out = (ldpc_ml_t)d;
        return LDPC_OK;
    });
}

int ldpc_ml_destroy(ldpc_ml_t h) {
    return guarded("ldpc_ml_destroy", [&]() -> int {
        ml_destroy((MlDecoder*)h);
        return LDPC_OK;
    });
}

int ldpc_ml_decode(ldpc_ml_t h, int channel, int dtype, const double* coef2, const void* y_dev, int64_t B,
                   const uint32_t* pick_dev, int32_t* index_dev, int32_t* ties_dev, uint32_t* tie_mask_dev, double* best_dev,
                   uint8_t* xhat_dev, void* stream) {
    return guarded("ldpc_ml_decode", [&]() -> int {
        if (!h || !coef2 || !y_dev || B < 0 || dtype < 0 || dtype > 1) {
            set_error("ldpc_ml_decode: bad arguments");
            return LDPC_E_ARG;
        }
        return ml_decode((MlDecoder*)h, channel, dtype, coef2, y_dev, B, pick_dev, index_dev, ties_dev, tie_mask_dev, best_dev, xhat_dev,
                         (hipStream_t)stream);
    });
}

int ldpc_ml_simulate(ldpc_ml_t h, int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id,
                     uint64_t frame0, int64_t B, int64_t* counters_dev, void* stream) {
    return guarded("ldpc_ml_simulate", [&]() -> int {
        if (!h || !counters_dev || B < 0 || dtype < 0 || dtype > 1 || (codeword != 0 && codeword != 1)) {
            set_error("ldpc_ml_simulate: bad arguments");
            return LDPC_E_ARG;
        }
        return ml_simulate((MlDecoder*)h, channel, dtype, param, codeword, seed, stream_id, frame0, B, counters_dev, (hipStream_t)stream);
    });
}

// ---- ADMM LP decoder (ldpc_admm.hip) ----
int ldpc_admm_create(ldpc_code_t code, ldpc_admm_t* out) {
    return guarded("ldpc_admm_create", [&]() -> int {
        if (!code || !out) {
            set_error("ldpc_admm_create: bad arguments");
            return LDPC_E_ARG;
        }
        AdmmDecoder* d = nullptr;
        LDPC_TRY(admm_create((Code*)code, &d));
        *out = (ldpc_admm_t)d;
        return LDPC_OK;
    });
}

int ldpc_admm_destroy(ldpc_admm_t h) {
    return guarded("ldpc_admm_destroy", [&]() -> int {
        admm_destroy((AdmmDecoder*)h);
        return LDPC_OK;
    });
}

int ldpc_admm_last_backend(ldpc_admm_t h, int* backend) {
    return guarded("ldpc_admm_last_backend", [&]() -> int {
        if (!h || !backend) return LDPC_E_ARG;
        *backend = admm_last_backend((AdmmDecoder*)h);
        return LDPC_OK;
    });
}

int ldpc_admm_last_repacks(ldpc_admm_t h, int* repacks) {
    return guarded("ldpc_admm_last_repacks", [&]() -> int {
        if (!h || !repacks) return LDPC_E_ARG;
        *repacks = admm_last_repacks((AdmmDecoder*)h);
        return LDPC_OK;
    });
}

int ldpc_admm_decode(ldpc_admm_t h, const double* gamma_dev, int64_t B, double mu, double eps, int32_t max_iter, double* x_dev,
                     int32_t* iters_dev, uint8_t* converged_dev, void* stream) {
    return guarded("ldpc_admm_decode", [&]() -> int {
        if (!h || !gamma_dev || !x_dev || !iters_dev || B < 0) {
            set_error("ldpc_admm_decode: bad arguments");
            return LDPC_E_ARG;
        }
        return admm_decode((AdmmDecoder*)h, gamma_dev, B, mu, eps, max_iter, x_dev, iters_dev, converged_dev, (hipStream_t)stream);
    });
}

}  // extern "C"
