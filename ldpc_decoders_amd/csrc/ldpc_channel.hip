// Channel + LLR kernels and the error counters of the Monte-Carlo loop.
//
// Reference statements (thadikari/ldpc_decoders):
//   BI-AWGN  sigma^2 = 10^(-snr/10); y = (2x-1) + N(0, sigma); prior = -2y/sigma^2 ...... src/biawgn.py:10-28
//   BSC      y = (x + [u < p]) mod 2; prior = (log(1-p) - log(p)) * (1 - 2y) ............ src/bsc.py:11-25
//   BEC      y = 2 where u < p else x .................................................. src/bec.py:11-18
//   counters errors = #(x != x_hat); wec += errors > 0; bec += errors; tot += 1 ........ src/main.py:41-45
// The reference draws from numpy's global MT19937; the device draws from Philox keyed by the global frame
// index (ldpc_rng.hpp) -- host-generated noise is used wherever bit parity with the reference is claimed.
#include "ldpc_common.hpp"
#include "ldpc_rng.hpp"

namespace ldpc {
namespace {

// Random codeword of a frame (--codeword -1, src/main.py:38: x = code.cb[np.random.choice(K)]): word floor(w * K / 2^32) of the
// codebook, w = first Philox word of block 0xFFFFFFFE of the frame (the noise uses blocks 0 .. n/4, the ML tie-break 0xFFFFFFFF).
struct WordBook {
    const uint8_t* cb;  // [K, n] bytes in {0,1} (Code.cb, src/codes.py:11-14); null: the all-`codeword` word
    int64_t K;
    uint8_t* sent;      // [B, n] out: the word each frame sent
};
__device__ __forceinline__ int64_t word_of_frame(const WordBook& wb, uint64_t seed, uint32_t stream, uint64_t frame) {
    const Philox4 p = philox_word_block(seed, stream, frame, 0xFFFFFFFEu);
    return (int64_t)(((unsigned long long)p.w[0] * (unsigned long long)wb.K) >> 32);
}

// one thread = 4 consecutive variables of one frame (one Philox block)
template <typename T, bool WORDS>
__global__ __launch_bounds__(256) void k_biawgn(double sigma, double inv_var2, int codeword, uint64_t seed, uint32_t stream,
                                                uint64_t frame0, int64_t B, int n, int blocks_per_frame, T* __restrict__ priors, WordBook wb,
                                                double grid_scale, const unsigned long long* __restrict__ list) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t f = gid / blocks_per_frame;
    const int j = (int)(gid - f * blocks_per_frame);
    if (f >= B) return;
    // `list` (ldpc_channel_list): row f is global frame list[1 + f], for the first min(list[0], B) rows -- a frame list that lives in
    // device memory (the redo list of the exactness guard): no host round trip between the kernel that wrote it and this one
    if (list && (unsigned long long)f >= list[0]) return;
    if (list) frame0 = list[1 + f] - (uint64_t)f;
    const Philox4 p = philox_word_block(seed, stream, frame0 + (uint64_t)f, (uint32_t)j);
    T z[4];
    box_muller<T>(p.w[0], p.w[1], z[0], z[1]);
    box_muller<T>(p.w[2], p.w[3], z[2], z[3]);
    const T sg = (T)sigma, k = (T)inv_var2;
    T mean4[4];
    if constexpr (WORDS) {
        const uint8_t* word = wb.cb + word_of_frame(wb, seed, stream, frame0 + (uint64_t)f) * n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint8_t bit = 4 * j + q < n ? word[4 * j + q] : (uint8_t)0;
            mean4[q] = (T)(2 * (int)bit - 1);
            if (4 * j + q < n) wb.sent[f * n + 4 * j + q] = bit;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) mean4[q] = (T)(2 * codeword - 1);
    }
    T out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const T y = mean4[q] + sg * z[q];
        out[q] = -(k * y);  // -2y/sigma^2 with k = 2/sigma^2
        // exact-in-fp32 mode (LDPC_CH_PRIOR_GRID): the LLR rounded to a multiple of 2^-k; both scalings are exact (powers of two)
        if (grid_scale != 0.0) out[q] = __builtin_rint(out[q] * (T)grid_scale) * (T)(1.0 / grid_scale);
    }
    T* dst = priors + f * n + 4 * j;
    if ((n & 3) == 0) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            *reinterpret_cast<double2*>(dst) = make_double2(out[0], out[1]);
            *reinterpret_cast<double2*>(dst + 2) = make_double2(out[2], out[3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * j + q < n) dst[q] = out[q];
    }
}

// BSC / BEC: one word per variable; event <=> (w + 0.5) * 2^-32 < p <=> w < thr
template <typename T, int CH, bool WORDS>
__global__ __launch_bounds__(256) void k_discrete(uint64_t thr, double llr, int codeword, uint64_t seed, uint32_t stream,
                                                  uint64_t frame0, int64_t B, int n, int blocks_per_frame,
                                                  T* __restrict__ priors, uint8_t* __restrict__ y, WordBook wb, double grid_scale) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t f = gid / blocks_per_frame;
    const int j = (int)(gid - f * blocks_per_frame);
    if (f >= B) return;
    const Philox4 p = philox_word_block(seed, stream, frame0 + (uint64_t)f, (uint32_t)j);
    const uint8_t* word = nullptr;
    if constexpr (WORDS) word = wb.cb + word_of_frame(wb, seed, stream, frame0 + (uint64_t)f) * n;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int v = 4 * j + q;
        if (v < n) {
            const bool hit = (uint64_t)p.w[q] < thr;
            int bit = codeword;
            if constexpr (WORDS) {
                bit = word[v];
                wb.sent[f * n + v] = (uint8_t)bit;
            }
            uint8_t s;
            if constexpr (CH == CH_BSC) {
                s = (uint8_t)(bit ^ (hit ? 1 : 0));
                if (priors) {
                    const T l = grid_scale != 0.0 ? (T)(__builtin_rint((T)llr * (T)grid_scale) * (T)(1.0 / grid_scale)) : (T)llr;
                    priors[f * n + v] = l * (T)(1 - 2 * (int)s);
                }
            } else {
                s = hit ? (uint8_t)2 : (uint8_t)bit;
            }
            y[f * n + v] = s;
        }
    }
}

// one wavefront per frame, grid-stride; counters are combined per block (LDS) before touching global atomics
__global__ __launch_bounds__(256) void k_count(const uint8_t* __restrict__ xhat, const uint8_t* __restrict__ sent, int codeword,
                                               const int32_t* __restrict__ iters, int64_t B, int n, int hist_bins,
                                               unsigned long long* __restrict__ counters, int sent_per_frame) {
    extern __shared__ unsigned int s_hist[];  // [hist_bins]
    __shared__ unsigned long long s_cnt[4];
    for (int i = threadIdx.x; i < hist_bins; i += 256) s_hist[i] = 0;
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long tot = 0, wec = 0, bec = 0, itsum = 0;  // meaningful on lane 0 of each wave
    for (int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); f < B; f += (int64_t)gridDim.x * 4) {
        int err = 0;
        const uint8_t* row = xhat + f * n;
        for (int v = lane; v < n; v += 64) {
            const uint8_t want = sent ? sent[sent_per_frame ? f * n + v : v] : (uint8_t)codeword;
            err += row[v] != want;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) err += __shfl_xor(err, off, 64);
        if (lane == 0) {
            tot += 1;
            wec += err > 0;
            bec += (unsigned long long)err;
            if (iters) {
                const int it = iters[f];
                itsum += (unsigned long long)it;
                if (hist_bins > 0) atomicAdd(&s_hist[it < hist_bins ? it : hist_bins - 1], 1u);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&s_cnt[0], tot);
        atomicAdd(&s_cnt[1], wec);
        atomicAdd(&s_cnt[2], bec);
        atomicAdd(&s_cnt[3], itsum);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
    for (int i = threadIdx.x; i < hist_bins; i += 256)
        if (s_hist[i]) atomicAdd(&counters[4 + i], (unsigned long long)s_hist[i]);
}

// decisions [B,n] bytes -> packed words [B,W] (bit v & 31 of word v >> 5); `erased` (erasure decoder: symbol 2) optional.  One wave
// per 64 variables of a frame: two ballots.
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ xhat, int64_t B, int n, int W, uint32_t* __restrict__ bits,
                                                   uint32_t* __restrict__ erased) {
    const int lane = threadIdx.x & 63;
    const int nvb = (n + 63) / 64;
    const int64_t tasks = B * nvb;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tasks; t += (int64_t)gridDim.x * 4) {
        const int64_t f = t / nvb;
        const int vb = (int)(t - f * nvb), v = vb * 64 + lane;
        const uint8_t x = v < n ? xhat[f * n + v] : (uint8_t)0;
        const unsigned long long one = __ballot(x == 1), era = __ballot(x == 2);
        if (lane < 2 && 2 * vb + lane < W) {
            bits[f * W + 2 * vb + lane] = (uint32_t)(lane ? one >> 32 : one);
            if (erased) erased[f * W + 2 * vb + lane] = (uint32_t)(lane ? era >> 32 : era);
        }
    }
}

// the counters of k_count from packed decisions: errors of a frame = popcount((bits ^ sent) | erased) over its W words
__global__ __launch_bounds__(256) void k_count_bits(const uint32_t* __restrict__ bits, const uint32_t* __restrict__ erased,
                                                    const uint32_t* __restrict__ sent, int codeword, const int32_t* __restrict__ iters, int64_t B,
                                                    int n, int W, int hist_bins, unsigned long long* __restrict__ counters) {
    extern __shared__ unsigned int s_hist[];  // [hist_bins]
    __shared__ unsigned long long s_cnt[4];
    for (int i = threadIdx.x; i < hist_bins; i += 256) s_hist[i] = 0;
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long tot = 0, wec = 0, bec = 0, itsum = 0;
    for (int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); f < B; f += (int64_t)gridDim.x * 4) {
        int err = 0;
        for (int w = lane; w < W; w += 64) {
            const int left = n - 32 * w;
            const uint32_t mask = left >= 32 ? 0xffffffffu : ((1u << left) - 1u);
            const uint32_t want = sent ? sent[w] : (codeword ? 0xffffffffu : 0u);
            uint32_t diff = bits[f * W + w] ^ want;
            if (erased) diff |= erased[f * W + w];
            err += __popc(diff & mask);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) err += __shfl_xor(err, off, 64);
        if (lane == 0) {
            tot += 1;
            wec += err > 0;
            bec += (unsigned long long)err;
            if (iters) {
                const int it = iters[f];
                itsum += (unsigned long long)it;
                if (hist_bins > 0) atomicAdd(&s_hist[it < hist_bins ? it : hist_bins - 1], 1u);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&s_cnt[0], tot);
        atomicAdd(&s_cnt[1], wec);
        atomicAdd(&s_cnt[2], bec);
        atomicAdd(&s_cnt[3], itsum);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_cnt[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_cnt[threadIdx.x]);
    for (int i = threadIdx.x; i < hist_bins; i += 256)
        if (s_hist[i]) atomicAdd(&counters[4 + i], (unsigned long long)s_hist[i]);
}

// The counters of k_count for the rows of a device-resident frame list (the fp64 re-decodes of the exactness guard): row i < min(list[0], rows)
// is global frame list[1 + i] and is counted into the counter row of ITS round -- (frame - frame_base) / round_stride -- so that a block of
// Monte-Carlo rounds that shared one redo pass keeps one exact counter row per round.  One wave per frame, direct atomics (a handful of frames).
__global__ __launch_bounds__(256) void k_count_list(const uint8_t* __restrict__ xhat, int codeword, const int32_t* __restrict__ iters,
                                                    const unsigned long long* __restrict__ list, int64_t rows, int n, int hist_bins,
                                                    unsigned long long* __restrict__ counters, int64_t counter_stride, unsigned long long frame_base,
                                                    unsigned long long round_stride, int64_t nrounds, unsigned long long* __restrict__ redone) {
    const unsigned long long have = list[0];
    const int64_t B = have < (unsigned long long)rows ? (int64_t)have : rows;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&redone[0], (unsigned long long)B);
        if (have > (unsigned long long)rows) atomicAdd(&redone[1], 1ull);
    }
    const int lane = threadIdx.x & 63;
    for (int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); f < B; f += (int64_t)gridDim.x * 4) {
        int err = 0;
        const uint8_t* row = xhat + f * n;
        for (int v = lane; v < n; v += 64) err += row[v] != (uint8_t)codeword;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) err += __shfl_xor(err, off, 64);
        if (lane == 0) {
            int64_t r = round_stride ? (int64_t)((list[1 + f] - frame_base) / round_stride) : 0;
            r = r < 0 ? 0 : (r >= nrounds ? nrounds - 1 : r);
            unsigned long long* c = counters + r * counter_stride;
            const int it = iters[f];
            atomicAdd(&c[0], 1ull);
            if (err > 0) atomicAdd(&c[1], 1ull);
            if (err > 0) atomicAdd(&c[2], (unsigned long long)err);
            atomicAdd(&c[3], (unsigned long long)it);
            if (hist_bins > 0) atomicAdd(&c[4 + (it < hist_bins ? it : hist_bins - 1)], 1ull);
        }
    }
}

// 4-byte-per-lane coalesced copy with a known byte count: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for the access
// width the streaming kernels use (MI355X_MICROARCH.md: the counters are only calibrated for 16-byte-per-lane streams)
__global__ __launch_bounds__(256) void k_copy4(const float* __restrict__ src, float* __restrict__ dst, int64_t nwords) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

int debug_copy4(const void* src, void* dst, int64_t nbytes, hipStream_t st) {
    hipLaunchKernelGGL(k_copy4, dim3(8192), dim3(256), 0, st, (const float*)src, (float*)dst, nbytes / 4);
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int channel_generate(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0,
                     int64_t B, int32_t n, void* priors, uint8_t* y, hipStream_t st) {
    return channel_generate_words(channel, dtype, param, codeword, nullptr, 0, seed, stream_id, frame0, B, n, priors, y, nullptr, st);
}

// rows = the frames of a device-resident list ([0] = how many, [1..] = global frame indices), at most `cap` of them (BI-AWGN)
int channel_generate_list(int channel, int dtype, double param, int codeword, uint64_t seed, uint64_t stream_id, const uint64_t* list_dev, int64_t cap,
                          int32_t n, void* priors, hipStream_t st) {
    if ((channel & 0xff) != CH_BIAWGN || !list_dev) {
        set_error("ldpc_channel_list: BI-AWGN with a device frame list");
        return LDPC_E_ARG;
    }
    return channel_generate_words(channel, dtype, param, codeword, nullptr, 0, seed, stream_id, 0, cap, n, priors, nullptr, nullptr, st,
                                  (const unsigned long long*)list_dev);
}

int channel_generate_words(int channel, int dtype, double param, int codeword, const uint8_t* codebook, int64_t K, uint64_t seed,
                           uint64_t stream_id, uint64_t frame0, int64_t B, int32_t n, void* priors, uint8_t* y, uint8_t* sent,
                           hipStream_t st, const unsigned long long* list) {
    if (B <= 0) return LDPC_OK;
    const bool words = codebook != nullptr;
    if (words && (K <= 0 || K > ((int64_t)1 << 31) || !sent)) {
        set_error("random-codeword channel: needs 1 <= K <= 2^31 codebook words and the sent-word output buffer");
        return LDPC_E_ARG;
    }
    if (!words && codeword != 0 && codeword != 1) {
        set_error("device channel kernels send the all-zero (0) or all-one (1) word; got codeword=%d", codeword);
        return LDPC_E_ARG;
    }
    const WordBook wb{codebook, K, sent};
    const bool raw = (channel & CH_RAW_OBSERVATION) != 0;
    const int grid_k = LDPC_CH_PRIOR_GRID_OF(channel);  // -1: off
    const double gs = grid_k >= 0 ? ldexp(1.0, grid_k) : 0.0;
    channel &= ~(CH_RAW_OBSERVATION | (0x1f << 12));
    if (grid_k > 23 || (grid_k >= 0 && raw)) {
        set_error("prior grid: k = 0..23, LLR output only");
        return LDPC_E_ARG;
    }
    if (raw && channel != CH_BIAWGN) {
        set_error("the raw-observation flag applies to the BI-AWGN channel only");
        return LDPC_E_ARG;
    }
    const int bpf = (n + 3) / 4;
    const int64_t threads = B * bpf;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (channel == CH_BIAWGN) {
        if (!priors) {
            set_error("biawgn channel needs a priors output buffer");
            return LDPC_E_ARG;
        }
        const double var = pow(10.0, -param / 10.0);  // src/biawgn.py:10
        const double sigma = sqrt(var), k = raw ? -1.0 : 2.0 / var;  // the kernel writes -(k*y): k = -1 hands over y itself
        if (dtype == DT_F64) {
            if (words) hipLaunchKernelGGL((k_biawgn<double, true>), grid, block, 0, st, sigma, k, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (double*)priors, wb, gs, list);
            else hipLaunchKernelGGL((k_biawgn<double, false>), grid, block, 0, st, sigma, k, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (double*)priors, wb, gs, list);
        } else {
            if (words) hipLaunchKernelGGL((k_biawgn<float, true>), grid, block, 0, st, sigma, k, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)priors, wb, gs, list);
            else hipLaunchKernelGGL((k_biawgn<float, false>), grid, block, 0, st, sigma, k, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)priors, wb, gs, list);
        }
    } else if (channel == CH_BSC || channel == CH_BEC) {
        if (!y) {
            set_error("discrete channels need the y output buffer");
            return LDPC_E_ARG;
        }
        if (!(param >= 0.0 && param <= 1.0)) {
            set_error("channel probability %g outside [0,1]", param);
            return LDPC_E_ARG;
        }
        // (w + 0.5) * 2^-32 < p  <=>  w < ceil(p * 2^32 - 0.5)
        double t = ceil(param * 4294967296.0 - 0.5);
        if (t < 0) t = 0;
        const uint64_t thr = (uint64_t)t;
        const double llr = log(1.0 - param) - log(param);  // src/bsc.py:21
        if (channel == CH_BSC) {
            if (dtype == DT_F64) {
                if (words) hipLaunchKernelGGL((k_discrete<double, CH_BSC, true>), grid, block, 0, st, thr, llr, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (double*)priors, y, wb, gs);
                else hipLaunchKernelGGL((k_discrete<double, CH_BSC, false>), grid, block, 0, st, thr, llr, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (double*)priors, y, wb, gs);
            } else {
                if (words) hipLaunchKernelGGL((k_discrete<float, CH_BSC, true>), grid, block, 0, st, thr, llr, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)priors, y, wb, gs);
                else hipLaunchKernelGGL((k_discrete<float, CH_BSC, false>), grid, block, 0, st, thr, llr, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)priors, y, wb, gs);
            }
        } else {
            if (words) hipLaunchKernelGGL((k_discrete<float, CH_BEC, true>), grid, block, 0, st, thr, 0.0, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)nullptr, y, wb, 0.0);
            else hipLaunchKernelGGL((k_discrete<float, CH_BEC, false>), grid, block, 0, st, thr, 0.0, codeword, seed, (uint32_t)stream_id, frame0, B, n, bpf, (float*)nullptr, y, wb, 0.0);
        }
    } else {
        set_error("unknown channel id %d", channel);
        return LDPC_E_ARG;
    }
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int pack_bits(const uint8_t* xhat, int64_t B, int32_t n, uint32_t* bits, uint32_t* erased, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    const int64_t want = (B * ((n + 63) / 64) + 3) / 4;
    const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(k_pack_bits, dim3(grid), dim3(256), 0, st, xhat, B, n, (n + 31) / 32, bits, erased);
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int count_errors_bits(const uint32_t* bits, const uint32_t* erased, const uint32_t* sent_bits, int codeword, const int32_t* iters, int64_t B,
                      int32_t n, int32_t hist_bins, int64_t* counters, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    if (hist_bins > 8192) {
        set_error("at most 8192 histogram bins");
        return LDPC_E_ARG;
    }
    const int64_t want = (B + 3) / 4;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_count_bits, dim3(grid), dim3(256), (size_t)hist_bins * sizeof(unsigned int), st, bits, erased, sent_bits, codeword, iters, B,
                       n, (n + 31) / 32, hist_bins, (unsigned long long*)counters);
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int count_errors_list(const uint8_t* xhat, int codeword, const int32_t* iters, const uint64_t* list_dev, int64_t rows, int32_t n, int32_t hist_bins,
                      int64_t* counters, int64_t counter_stride, uint64_t frame_base, uint64_t round_stride, int64_t nrounds, int64_t* redone2,
                      hipStream_t st) {
    if (rows <= 0) return LDPC_OK;
    hipLaunchKernelGGL(k_count_list, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, xhat, codeword, iters, (const unsigned long long*)list_dev, rows, n,
                       hist_bins, (unsigned long long*)counters, counter_stride, (unsigned long long)frame_base, (unsigned long long)round_stride, nrounds,
                       (unsigned long long*)redone2);
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int count_errors(const uint8_t* xhat, const uint8_t* sent, int codeword, const int32_t* iters, int64_t B, int32_t n,
                 int32_t hist_bins, int64_t* counters, hipStream_t st) {
    return count_errors_words(xhat, sent, 0, codeword, iters, B, n, hist_bins, counters, st);
}

int count_errors_words(const uint8_t* xhat, const uint8_t* sent, int sent_per_frame, int codeword, const int32_t* iters, int64_t B, int32_t n,
                       int32_t hist_bins, int64_t* counters, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    if (hist_bins > 8192) {
        set_error("at most 8192 histogram bins");
        return LDPC_E_ARG;
    }
    const int64_t want = (B + 3) / 4;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_count, dim3(grid), dim3(256), (size_t)hist_bins * sizeof(unsigned int), st, xhat, sent, codeword, iters, B, n,
                       hist_bins, (unsigned long long*)counters, sent_per_frame);
    LDPC_HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

}  // namespace ldpc
