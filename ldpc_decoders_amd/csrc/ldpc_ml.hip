// Maximum-likelihood decoding of the short built-in codes by exhaustive codebook search.
//
//   BI-AWGN  reference src/biawgn.py:66-78   log_prob_k = sum_j -((2 c_kj - 1 - y_j)^2) / (2 noise_var)
//   BSC      reference src/bsc.py:63-75      log_prob_k = num_diffs_k * log p + num_agrees_k * log(1-p)
//   BEC      reference src/bec.py:21-36      log_prob_k = num_erasures * log p + num_agrees_k * log(1-p), -inf if the word disagrees
//   pick     reference src/math_utils.py:72-74  uniformly among the maximisers (arg_max_rand)
//
// One lane owns one frame (n <= 64, so a codeword is one 64-bit mask and every codebook access is wave-uniform).  The
// metric is computed in fp64 with the reference's operation order, including numpy's summation order for
// np.sum(exponent, axis=1) (0.0 + pairwise sum: straight loop from -0.0 below 8 terms, otherwise 8 strided accumulators
// combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the remainder added one by one), so the metric maximum and the
// SET of maximisers are bit-identical to the reference's.  Three passes over the codebook (maximum; tie set; pick):
// recomputing a 12-term sum is cheaper than keeping 2^k metrics per lane.
#include "ldpc_common.hpp"
#include "ldpc_rng.hpp"

#include <cmath>
#include <new>

namespace ldpc {

struct MlDecoder {
    int device = 0;
    int64_t K = 0;
    int32_t n = 0;
    int32_t W = 0;  // 32-bit words of a tie mask
    uint64_t* d_cb = nullptr;  // [K] bit j = codeword[j]
    DevBuf obs, sym;           // simulate scratch: observations / symbols of one chunk
};

namespace {

constexpr uint32_t ML_PICK_BLOCK = 0xFFFFFFFFu;  // Philox block index of a frame's tie-break word (never a noise block)

struct MlArgs {
    const uint64_t* cb;
    int64_t K;
    int n, W;
    int64_t B;
    const void* y;          // [B,n] double / float (BI-AWGN) or uint8 symbols (BSC / BEC)
    double c0, c1;          // BI-AWGN: 2*noise_var, unused ; BSC/BEC: log p, log(1-p)
    const uint32_t* pick;   // [B] tie-break draws or null
    int use_philox;         // draw the tie-break word from Philox(seed, stream, frame0 + f) instead
    uint64_t seed, frame0;
    uint32_t stream;
    int32_t* index;         // [B] out
    int32_t* ties;          // [B] out
    uint32_t* tie_mask;     // [B,W] out or null
    double* best;           // [B] out or null
    uint8_t* xhat;          // [B,n] out or null
    int64_t* counters;      // simulate: tot, wec, bec accumulated, or null
    int codeword;
};

// 0.0 + numpy's pairwise sum of the n selected terms; e is this lane's column of the term table in the LDS:
// term(c, j) at e[(c * n + j) * 64]
__device__ __forceinline__ double ml_term(const double* e, int n, uint64_t bits, int j) {
    return e[((((bits >> j) & 1ull) ? n : 0) + j) * 64];
}
__device__ double np_row_sum(const double* e, int n, uint64_t bits) {
    double res;
    if (n < 8) {
        res = -0.0;
        for (int j = 0; j < n; ++j) res += ml_term(e, n, bits, j);
    } else {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = ml_term(e, n, bits, k);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += ml_term(e, n, bits, i + k);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += ml_term(e, n, bits, i);
    }
    return 0.0 + res;
}

// CH: 0 BI-AWGN (T = observation type), 1 BSC, 2 BEC
template <typename T, int CH>
__global__ __launch_bounds__(64) void k_ml(const MlArgs A) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int n = A.n;
    const uint64_t maskn = n == 64 ? ~0ull : ((1ull << n) - 1);
    const double ninf = -__builtin_huge_val();
    double* e = lds + lane;
    int acc_tot = 0, acc_wec = 0, acc_bec = 0;  // simulate counters of this lane (flushed once per workgroup)

    // persistent workgroups: tiles of 64 frames, grid-stride (one set of counter atomics per workgroup, not per tile)
    for (int64_t tile = blockIdx.x; tile * 64 < A.B; tile += gridDim.x) {
        const int64_t f = tile * 64 + lane;
        const bool act = f < A.B;
        uint64_t is0 = 0, is1 = 0, era = 0;
        if constexpr (CH == 0) {
            const T* y = (const T*)A.y + (act ? f : 0) * n;
            for (int j = 0; j < n; ++j) {
                const double yj = (double)y[j];
                const double t0 = -1.0 - yj, t1 = 1.0 - yj;  // (cb*2 - 1) - y
                e[j * 64] = -(t0 * t0) / A.c0;
                e[(n + j) * 64] = -(t1 * t1) / A.c0;
            }
        } else {
            const uint8_t* y = (const uint8_t*)A.y + (act ? f : 0) * n;
            for (int j = 0; j < n; ++j) {
                const uint8_t s = y[j];
                is0 |= (uint64_t)(s == 0) << j;
                is1 |= (uint64_t)(s == 1) << j;
                era |= (uint64_t)(s > 1) << j;
            }
        }
        const double n_era = (double)__popcll(era);

        auto metric = [&](uint64_t c) -> double {
            if constexpr (CH == 0) {
                return np_row_sum(e, n, c);
            } else {
                const int agrees = __popcll(((~c & is0) | (c & is1)) & maskn);
                if constexpr (CH == 1) {
                    const int diffs = n - agrees;
                    return (double)diffs * A.c0 + (double)agrees * A.c1;
                } else {
                    const int diffs = n - agrees - __popcll(era);
                    const double lp = n_era * A.c0 + (double)agrees * A.c1;
                    return diffs > 0 ? ninf : lp;
                }
            }
        };

        // pass 1: the maximum (np.max: comparisons only)
        double best = metric(A.cb[0]);
        for (int64_t k = 1; k < A.K; ++k) {
            const double m = metric(A.cb[k]);
            best = m > best ? m : best;
        }
        // pass 2: the set of maximisers (values == np.max(values))
        int ties = 0;
        int64_t first = 0;
        uint32_t word = 0;
        for (int64_t k = 0; k < A.K; ++k) {
            const bool hit = metric(A.cb[k]) == best;
            if (hit && ties == 0) first = k;
            ties += hit;
            word |= (uint32_t)hit << (k & 31);
            if ((k & 31) == 31 || k == A.K - 1) {
                if (A.tie_mask && act) A.tie_mask[f * A.W + (k >> 5)] = word;
                word = 0;
            }
        }
        // pass 3: the pick
        int64_t chosen = first;
        uint32_t draw = 0;
        bool have_draw = false;
        if (A.use_philox) {
            draw = philox_word_block(A.seed, A.stream, A.frame0 + (uint64_t)f, ML_PICK_BLOCK).w[0];
            have_draw = true;
        } else if (A.pick) {
            draw = A.pick[act ? f : 0];
            have_draw = true;
        }
        if (have_draw && ties > 1) {
            // floor(draw * ties / 2^32): uniform over the tie set up to 2^-32
            int target = (int)(((uint64_t)draw * (uint64_t)ties) >> 32);
            for (int64_t k = 0; k < A.K; ++k) {
                if (metric(A.cb[k]) == best) {
                    if (target == 0) {
                        chosen = k;
                        break;
                    }
                    --target;
                }
            }
        }
        const uint64_t cw = A.cb[chosen];
        if (act) {
            if (A.index) A.index[f] = (int32_t)chosen;
            if (A.ties) A.ties[f] = ties;
            if (A.best) A.best[f] = best;
            if (A.xhat) {
                uint8_t* o = A.xhat + f * n;
                for (int j = 0; j < n; ++j) o[j] = (uint8_t)((cw >> j) & 1ull);
            }
            const uint64_t sent = A.codeword ? maskn : 0ull;
            const int err = __popcll((cw ^ sent) & maskn);
            acc_tot += 1;
            acc_wec += err > 0;
            acc_bec += err;
        }
    }
    if (A.counters) {
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            acc_bec += __shfl_xor(acc_bec, o);
            acc_wec += __shfl_xor(acc_wec, o);
            acc_tot += __shfl_xor(acc_tot, o);
        }
        if (lane == 0) {
            atomicAdd((unsigned long long*)&A.counters[LDPC_CNT_TOT], (unsigned long long)acc_tot);
            atomicAdd((unsigned long long*)&A.counters[LDPC_CNT_WEC], (unsigned long long)acc_wec);
            atomicAdd((unsigned long long*)&A.counters[LDPC_CNT_BEC], (unsigned long long)acc_bec);
        }
    }
}

int ml_launch(MlDecoder* d, int channel, int dtype, MlArgs& A, hipStream_t st) {
    if (A.B <= 0) return LDPC_OK;
    A.cb = d->d_cb;
    A.K = d->K;
    A.n = d->n;
    A.W = d->W;
    const int64_t tiles = (A.B + 63) / 64;
    const dim3 grid((unsigned)(tiles < 8192 ? tiles : 8192)), block(64);  // 256 CUs x 32 resident single-wave workgroups
    const size_t lds = channel == CH_BIAWGN ? (size_t)2 * d->n * 64 * sizeof(double) : 0;
    if (channel == CH_BIAWGN) {
        if (dtype == DT_F64)
            hipLaunchKernelGGL((k_ml<double, 0>), grid, block, lds, st, A);
        else
            hipLaunchKernelGGL((k_ml<float, 0>), grid, block, lds, st, A);
    } else if (channel == CH_BSC) {
        hipLaunchKernelGGL((k_ml<uint8_t, 1>), grid, block, lds, st, A);
    } else if (channel == CH_BEC) {
        hipLaunchKernelGGL((k_ml<uint8_t, 2>), grid, block, lds, st, A);
    } else {
        set_error("unknown channel id %d", channel);
        return LDPC_E_ARG;
    }
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

}  // namespace

int ml_create(int device, const uint8_t* codebook, int64_t K, int32_t n, MlDecoder** out) {
    if (!codebook || !out || K <= 0 || K > ((int64_t)1 << 20) || n <= 0 || n > 64) {
        set_error("ldpc_ml_create: need a codebook of 1..2^20 words of 1..64 bits (K=%lld n=%d)", (long long)K, n);
        return LDPC_E_ARG;
    }
    MlDecoder* d = new (std::nothrow) MlDecoder();
    if (!d) return LDPC_E_NOMEM;
    d->device = device;
    d->K = K;
    d->n = n;
    d->W = (int32_t)((K + 31) / 32);
    std::vector<uint64_t> bits((size_t)K, 0);
    for (int64_t k = 0; k < K; ++k)
        for (int j = 0; j < n; ++j) {
            const uint8_t b = codebook[k * n + j];
            if (b > 1) {
                set_error("codebook entry (%lld,%d) = %d is not a bit", (long long)k, j, (int)b);
                delete d;
                return LDPC_E_ARG;
            }
            bits[(size_t)k] |= (uint64_t)b << j;
        }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_cb, (size_t)K * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(d->d_cb, bits.data(), (size_t)K * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("ldpc_ml_create: %s", hipGetErrorString(e));
        if (d->d_cb) (void)hipFree(d->d_cb);
        delete d;
        return LDPC_E_HIP;
    }
    *out = d;
    return LDPC_OK;
}

void ml_destroy(MlDecoder* d) {
    if (!d) return;
    if (d->d_cb) (void)hipFree(d->d_cb);
    d->obs.release();
    d->sym.release();
    delete d;
}

void ml_info(const MlDecoder* d, int64_t* K, int32_t* n, int32_t* W) {
    if (K) *K = d->K;
    if (n) *n = d->n;
    if (W) *W = d->W;
}

int ml_decode(MlDecoder* d, int channel, int dtype, const double* coef, const void* y, int64_t B, const uint32_t* pick,
              int32_t* index, int32_t* ties, uint32_t* tie_mask, double* best, uint8_t* xhat, hipStream_t st) {
    MlArgs A{};
    A.B = B;
    A.y = y;
    A.c0 = coef[0];
    A.c1 = coef[1];
    A.pick = pick;
    A.index = index;
    A.ties = ties;
    A.tie_mask = tie_mask;
    A.best = best;
    A.xhat = xhat;
    return ml_launch(d, channel, dtype, A, st);
}

int ml_simulate(MlDecoder* d, int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id,
                uint64_t frame0, int64_t B, int64_t* counters, hipStream_t st) {
    // observations of one chunk are staged in HBM (<= 256 MB): written by the channel kernel, read once by the search
    const size_t obs_elem = channel == CH_BIAWGN ? (dtype == DT_F64 ? 8 : 4) : 1;
    int64_t chunk = (int64_t)(((size_t)256 << 20) / ((size_t)d->n * obs_elem));
    chunk = chunk < 64 ? 64 : chunk & ~(int64_t)63;
    for (int64_t done = 0; done < B; done += chunk) {
        const int64_t cnt = B - done < chunk ? B - done : chunk;
        MlArgs A{};
        A.B = cnt;
        A.use_philox = 1;
        A.seed = seed;
        A.stream = (uint32_t)stream_id;
        A.frame0 = frame0 + (uint64_t)done;
        A.counters = counters;
        A.codeword = codeword;
        if (channel == CH_BIAWGN) {
            const size_t es = dtype == DT_F64 ? 8 : 4;
            LDPC_TRY(d->obs.reserve((size_t)cnt * d->n * es));
            LDPC_TRY(channel_generate(CH_BIAWGN | CH_RAW_OBSERVATION, dtype, param, codeword, seed, stream_id, A.frame0, cnt, d->n,
                                      d->obs.p, nullptr, st));
            A.y = d->obs.p;
            A.c0 = 2.0 * pow(10.0, -param / 10.0);  // 2 * noise_var (src/biawgn.py:10,75)
        } else {
            LDPC_TRY(d->sym.reserve((size_t)cnt * d->n));
            LDPC_TRY(channel_generate(channel, DT_F32, param, codeword, seed, stream_id, A.frame0, cnt, d->n, nullptr,
                                      (uint8_t*)d->sym.p, st));
            A.y = d->sym.p;
            A.c0 = log(param);  // src/bsc.py:67, src/bec.py:25
            A.c1 = log(1.0 - param);
        }
        LDPC_TRY(ml_launch(d, channel, dtype, A, st));
    }
    return LDPC_OK;
}

}  // namespace ldpc
