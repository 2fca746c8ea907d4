// Bit-sliced erasure decoder, streaming backend: the message state of codes that do not fit the LDS lives in HBM.
//
// Same decoder as ldpc_bec_kernels.hpp (src/bec.py:83-122; the rules and their bit-plane form are described there): a message is two
// bits, one lane word holds them for 32 frames.  A SUPERTILE is 64 lanes x 32 frames = 2048 frames; frame f of a batch sits in
// supertile f / 2048, lane (f % 2048) / 32, bit f % 32.  Per supertile, lines of 64 eight-byte elements {k plane, v plane}:
//     v2c  [E]  variable -> check messages in VARIABLE-major order (line p = position p of the CSC edge list): written as a stream by
//               the variable pass, each line gathered exactly once by the check that owns the edge;
//     sum  [m]  one summary element per check: A = exactly one incoming message erased, B = none erased | (A & parity of the +1s) --
//               the check's whole answer; the variable rebuilds its incoming message from (A, B) and its own last outgoing message;
//     prior[n], xhat[n]  received word and current decisions {erased plane, value plane}.
// HBM bytes per frame-sweep: check pass reads E lines, writes m; variable pass reads E (summaries, each re-used dc times on chip) + E
// (its own messages) + 2n, writes E + n: (4E + m + 3n) / 4 bytes -- 4 650 B at n = 1200 against the 14 400 B of the int8 lines this
// replaces and the 62 400 B of SURVEY 8(d)'s fp32 model.
//
// Exits are per frame (src/bec.py:96-97,120): `live` holds one word per lane; a frame that has left keeps its decisions (the update
// of xhat is gated by the live word), a supertile without a live frame is skipped by both passes.
#include <cstdlib>

#include "ldpc_bec_planes.hpp"
#include "ldpc_common.hpp"
#include "ldpc_rng.hpp"

namespace ldpc {

namespace {

using u64 = unsigned long long;
constexpr int SUPER = 64 * BEC_SLAB;  // frames per supertile

__device__ __forceinline__ P2 ld2(const uint2* p) {
    const uint2 t = *p;
    return P2{t.x, t.y};
}
__device__ __forceinline__ P2 ld2_nt(const uint2* p) {
    const u64 t = __builtin_nontemporal_load(reinterpret_cast<const u64*>(p));
    return P2{(uint32_t)t, (uint32_t)(t >> 32)};
}
__device__ __forceinline__ void st2_nt(uint2* p, uint32_t k, uint32_t v) {
    __builtin_nontemporal_store(((u64)v << 32) | k, reinterpret_cast<u64*>(p));
}

// received symbols [B, n] bytes {0, 1, 2 = erased} -> prior / xhat planes.  One workgroup = (supertile, 64 variables): lane l of every wave
// follows variable v0 + l through the frames (each load instruction reads the 64 consecutive bytes of one frame), builds the plane word
// of 32 frames in registers and parks it in an LDS tile [variable][lane word]; the tile then leaves as 512-byte lines.  (The first
// version had one thread per variable write its 8-byte elements 512 B apart: 4.2 GB of write traffic for 1.1 GB of planes at n = 64 800.)
// Frames beyond the batch are a known 0 (never erased: they leave before the first sweep).
__global__ __launch_bounds__(256) void k_becs_load(const uint8_t* __restrict__ y, int64_t B, int n, uint2* __restrict__ prior,
                                                   uint2* __restrict__ xhat, uint32_t* __restrict__ live, uint32_t* __restrict__ flags) {
    __shared__ uint2 tile[64][65];
    const int T = blockIdx.y, v0 = blockIdx.x * 64;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t f0 = (int64_t)T * SUPER;
    const bool has_v = v0 + l < n;
    for (int L = w; L < 64; L += 4) {  // lane word L of the supertile: frames f0 + 32 L ... + 31
        uint32_t kk = ~0u, vv = 0u;
        const int64_t fl = f0 + (int64_t)L * 32;
        if (fl < B) {
            kk = 0u;
            const int nb = (int)((B - fl) < 32 ? (B - fl) : 32);
            if (has_v) {
                const uint8_t* yp = y + fl * n + v0 + l;
                if (nb == 32) {
#pragma unroll 8
                    for (int b = 0; b < 32; ++b) {
                        const uint32_t sy = yp[(int64_t)b * n];
                        kk |= (sy != 2u ? 1u : 0u) << b;
                        vv |= (sy == 1u ? 1u : 0u) << b;
                    }
                } else {
                    for (int b = 0; b < nb; ++b) {
                        const uint32_t sy = yp[(int64_t)b * n];
                        kk |= (sy != 2u ? 1u : 0u) << b;
                        vv |= (sy == 1u ? 1u : 0u) << b;
                    }
                }
            } else {
                kk = ~0u;
            }
            if (nb < 32) kk |= ~0u << nb;
        }
        tile[l][L] = make_uint2(kk, vv);
        uint32_t era = has_v ? ~kk : 0u;  // frames of this lane word that hold an erasure among these 64 variables
#pragma unroll
        for (int o = 32; o; o >>= 1) era |= __shfl_xor(era, o);
        if (l == 0 && era) atomicOr(&flags[((size_t)T * 2 + 1) * 64 + L], era);
    }
    __syncthreads();
    for (int i = w; i < 64; i += 4) {
        if (v0 + i >= n) break;
        const uint2 e = tile[i][l];
        const size_t at = ((size_t)T * n + v0 + i) * 64 + l;
        prior[at] = e;
        xhat[at] = make_uint2(~e.x, e.y);  // x_hat starts as the received word (src/bec.py:89): {erased, value}
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int64_t fl = f0 + (int64_t)threadIdx.x * 32;
        const int64_t rem = B - fl;
        live[(size_t)T * 64 + threadIdx.x] = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
    }
}

// Frames leave (src/bec.py:96-97,120): before the first sweep those without an erasure, afterwards those whose last sweep changed
// nothing or left nothing erased.  flags[T][0] = changed, flags[T][1] = erased, per lane; both are cleared for the next sweep.
__global__ __launch_bounds__(64) void k_becs_check(uint32_t* __restrict__ flags, uint32_t* __restrict__ live, int32_t* __restrict__ iters,
                                                   int* __restrict__ live_tiles, int64_t B, int sweeps) {
    const int T = blockIdx.x, lane = threadIdx.x;
    const uint32_t lv = live[(size_t)T * 64 + lane];
    const uint32_t chg = sweeps == 0 ? ~0u : flags[((size_t)T * 2) * 64 + lane];
    const uint32_t era = flags[((size_t)T * 2 + 1) * 64 + lane];
    const uint32_t stay = lv & chg & era;
    uint32_t leave = lv & ~stay;
    live[(size_t)T * 64 + lane] = stay;
    flags[((size_t)T * 2) * 64 + lane] = 0u;
    flags[((size_t)T * 2 + 1) * 64 + lane] = 0u;
    const int64_t fl = (int64_t)T * SUPER + (int64_t)lane * 32;
    while (leave) {
        const int b = __builtin_ctz(leave);
        leave &= leave - 1u;
        if (fl + b < B) iters[fl + b] = sweeps;
    }
    if (__ballot(stay != 0u) != 0ull && lane == 0) atomicAdd(live_tiles, 1);
}

__global__ __launch_bounds__(64) void k_becs_finish(const uint32_t* __restrict__ live, int32_t* __restrict__ iters, int64_t B, int sweeps) {
    const int T = blockIdx.x, lane = threadIdx.x;
    uint32_t lv = live[(size_t)T * 64 + lane];
    const int64_t fl = (int64_t)T * SUPER + (int64_t)lane * 32;
    while (lv) {
        const int b = __builtin_ctz(lv);
        lv &= lv - 1u;
        if (fl + b < B) iters[fl + b] = sweeps;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Check pass: one wave = (supertile, run of checks).  A check reads the v2c line of each of its edges (each line exactly once in a
// sweep: non-temporal) and writes ONE summary line.
// The first sweep reads v2c = prior (src/bec.py:86) straight from the prior lines of the edges' variables (the caller passes the prior
// planes, n lines per supertile, and the edge -> variable table): no initialisation pass over the E message lines.
__global__ __launch_bounds__(256) void k_becs_cn(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_vpos,
                                                 const uint2* __restrict__ v2c, uint2* __restrict__ sum, const uint32_t* __restrict__ live,
                                                 int m, int64_t E, int tiles, int chunks, int cpw) {  // E: lines per supertile of `v2c`
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int T = __builtin_amdgcn_readfirstlane(task / chunks), chunk = __builtin_amdgcn_readfirstlane(task - (task / chunks) * chunks);
    if (T >= tiles) return;
    if (__ballot(live[(size_t)T * 64 + lane] != 0u) == 0ull) return;  // nobody left in this supertile
    const uint2* vt = v2c + (size_t)T * E * 64 + lane;
    uint2* st = sum + (size_t)T * m * 64 + lane;
    const int c_end = min(m, (chunk + 1) * cpw);
    for (int c = chunk * cpw; c < c_end; ++c) {
        const int k0 = row_ptr[c], k1 = row_ptr[c + 1];
        uint32_t all = ~0u, two = 0u, par = 0u;  // every incoming message known so far; at least two erased; parity of the +1s
        for (int k = k0; k < k1; k += 4) {
            P2 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = (k + j < k1) ? ld2_nt(vt + (size_t)edge_vpos[k + j] * 64) : P2{~0u, 0u};  // a known 0 is neutral
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                two = B3(two, all, e[j].k, X0 | (~X1 & ~X2));
                all &= e[j].k;
                par ^= e[j].v;
            }
        }
        const uint32_t sa = B3(all, two, two, ~X0 & ~X1);    // exactly one erased
        const uint32_t sb = B3(all, sa, par, X0 | (X1 & X2));  // none erased, or the parity the erased edge learns
        st[(size_t)c * 64] = make_uint2(sa, sb);
    }
}

// Variable pass: one wave = (supertile, run of variables).  Per variable: gather the summaries of its checks, stream its own last
// messages in and the new ones out (variable-major lines: contiguous), update the decisions of the live frames.
// DVMAX <= 16: the messages of a variable stay in registers between the counting pass and the output pass.
template <int DVMAX, bool FIRST>
__global__ __launch_bounds__(256) void k_becs_vn(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ chk_of_pos,
                                                 const uint2* __restrict__ sum, uint2* __restrict__ v2c, const uint2* __restrict__ prior,
                                                 uint2* __restrict__ xhat, const uint32_t* __restrict__ live, uint32_t* __restrict__ flags,
                                                 int n, int m, int64_t E, int tiles, int chunks, int vpw) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int T = __builtin_amdgcn_readfirstlane(task / chunks), chunk = __builtin_amdgcn_readfirstlane(task - (task / chunks) * chunks);
    if (T >= tiles) return;
    const uint32_t L = live[(size_t)T * 64 + lane];
    if (__ballot(L != 0u) == 0ull) return;
    const uint2* st = sum + (size_t)T * m * 64 + lane;
    uint2* vt = v2c + (size_t)T * E * 64 + lane;
    const uint2* pt = prior + (size_t)T * n * 64 + lane;
    uint2* xt = xhat + (size_t)T * n * 64 + lane;
    uint32_t chg = 0u, era = 0u;
    const int v_end = min(n, (chunk + 1) * vpw);
    for (int v = chunk * vpw; v < v_end; ++v) {
        const int p0 = col_ptr[v], deg = col_ptr[v + 1] - p0;
        const P2 pr = ld2(pt + (size_t)v * 64);
        const P2 xo = ld2(xt + (size_t)v * 64);
        uint32_t ck[DVMAX], cv[DVMAX], in[2 * (DVMAX + 1)];
        in[0] = pr.v;
        in[1] = B3(pr.k, pr.v, pr.v, ~X0 | X1);  // [prior >= 0]
        P2 s[DVMAX], o[DVMAX];
#pragma unroll
        for (int j = 0; j < DVMAX; ++j) {
            if (j < deg) {
                s[j] = ld2(st + (size_t)chk_of_pos[p0 + j] * 64);
                if constexpr (FIRST) o[j] = pr;  // the message of sweep 0 is the prior (src/bec.py:86)
                else o[j] = ld2_nt(vt + (size_t)(p0 + j) * 64);
            } else {
                s[j] = P2{0u, 0u};  // a missing edge: "no message"
                o[j] = P2{0u, 0u};
            }
        }
#pragma unroll
        for (int j = 0; j < DVMAX; ++j) {
            ck[j] = B3(s[j].k, s[j].v, o[j].k, (X1 & ~X0) | (X0 & ~X2));       // echo of a known message, or the one erased edge of its check
            cv[j] = s[j].v & B3(s[j].k, o[j].k, o[j].v, (X0 & ~X1) | (~X0 & X2));
            in[2 + 2 * j] = cv[j];
            in[3 + 2 * j] = B3(ck[j], cv[j], cv[j], ~X0 | X1);
        }
        uint32_t S[BitsFor<2 * (DVMAX + 1)>::value];
        plane_count<2 * (DVMAX + 1)>(in, S);  // S = marginal + DVMAX + 1
        const uint32_t ge0 = plane_ge(S, DVMAX + 1), ge1 = plane_ge(S, DVMAX + 2), ge2 = plane_ge(S, DVMAX + 3), gem1 = plane_ge(S, DVMAX);
        const uint32_t ne = B3(ge0, ge1, ge1, X0 & ~X1), nv = ge1;  // sign(marginal) -> erased / 1 / 0 (src/bec.py:119)
        const uint32_t xe = mux(L, ne, xo.k), xv = mux(L, nv, xo.v);
        chg |= (xe ^ xo.k) | (xv ^ xo.v);
        era |= xe & L;
        xt[(size_t)v * 64] = make_uint2(xe, xv);
#pragma unroll
        for (int j = 0; j < DVMAX; ++j) {
            if (j < deg) {
                const uint32_t pos = mux(ck[j], mux(cv[j], ge2, ge0), ge1);  // v2c_j = sign(marginal - c_j) (src/bec.py:116)
                const uint32_t neg = B3(ck[j], B3(cv[j], ge1, gem1, (X0 & ~X1) | (~X0 & ~X2)), ge0, (X0 & X1) | (~X0 & ~X2));
                st2_nt(vt + (size_t)(p0 + j) * 64, pos | neg, pos);
            }
        }
    }
    if (chg) atomicOr(&flags[((size_t)T * 2) * 64 + lane], chg);
    if (era) atomicOr(&flags[((size_t)T * 2 + 1) * 64 + lane], era);
}

// decisions -> [B, n] bytes {0, 1, 2 = still erased}; one thread per variable (coalesced along v)
__global__ __launch_bounds__(256) void k_becs_unpack(const uint2* __restrict__ xhat, uint8_t* __restrict__ out, int64_t B, int n) {
    // the reverse of k_becs_load: 512-byte lines in, an LDS tile [variable][lane word], then lane l writes variable v0 + l frame after
    // frame (64 consecutive bytes per store instruction)
    __shared__ uint2 tile[64][65];
    const int T = blockIdx.y, v0 = blockIdx.x * 64;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t f0 = (int64_t)T * SUPER;
    for (int i = w; i < 64; i += 4)
        if (v0 + i < n) tile[i][l] = xhat[((size_t)T * n + v0 + i) * 64 + l];
    __syncthreads();
    if (v0 + l >= n) return;
    for (int L = w; L < 64; L += 4) {
        const int64_t fl = f0 + (int64_t)L * 32;
        if (fl >= B) break;
        const uint2 e = tile[l][L];
        const int nb = (int)((B - fl) < 32 ? (B - fl) : 32);
        uint8_t* op = out + fl * n + v0 + l;
        for (int b = 0; b < nb; ++b) op[(int64_t)b * n] = ((e.x >> b) & 1u) ? (uint8_t)2 : (uint8_t)((e.y >> b) & 1u);
    }
}

// ---- Monte-Carlo path (ldpc_simulate): the received word is drawn straight into the planes and the decisions are counted straight from them --
// no [B, n] byte array on either side of the decoder.

// Erasure channel: the draws of k_discrete<float, CH_BEC> (ldpc_channel.hip) -- Philox block j of a frame holds the words of variables
// 4j .. 4j+3, erased <=> word < thr -- so that counters equal the composed channel -> decode -> count path frame for frame.  One wave =
// (supertile, run of blocks); lane l draws the 32 frames of its lane word.
__global__ __launch_bounds__(256) void k_becs_channel(uint64_t thr, int codeword, uint64_t seed, uint32_t stream, uint64_t frame0, int64_t B, int n,
                                                      int bpf, int qpw, uint2* __restrict__ prior, uint2* __restrict__ xhat,
                                                      uint32_t* __restrict__ live, uint32_t* __restrict__ flags) {
    const int T = blockIdx.y, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t fl = (int64_t)T * SUPER + (int64_t)l * 32;
    const int64_t rem = B - fl;
    const int nb = rem <= 0 ? 0 : (rem < 32 ? (int)rem : 32);
    const uint32_t real = nb == 32 ? ~0u : ((1u << nb) - 1u);  // frames of this lane word that exist
    const int j0 = (blockIdx.x * 4 + w) * qpw, j1 = j0 + qpw < bpf ? j0 + qpw : bpf;
    uint32_t era = 0u;
    for (int j = j0; j < j1; ++j) {
        uint32_t hit[4] = {0u, 0u, 0u, 0u};
        for (int b = 0; b < nb; ++b) {
            const Philox4 ph = philox_word_block(seed, stream, frame0 + (uint64_t)(fl + b), (uint32_t)j);
#pragma unroll
            for (int q = 0; q < 4; ++q) hit[q] |= ((uint64_t)ph.w[q] < thr ? 1u : 0u) << b;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = 4 * j + q;
            if (v < n) {
                const uint32_t kk = ~hit[q], vv = codeword ? (kk & real) : 0u;  // frames beyond the batch: a known 0
                const size_t at = ((size_t)T * n + v) * 64 + l;
                prior[at] = make_uint2(kk, vv);
                xhat[at] = make_uint2(hit[q], vv);  // x_hat starts as the received word (src/bec.py:89)
                era |= hit[q];
            }
        }
    }
    if (era) atomicOr(&flags[((size_t)T * 2 + 1) * 64 + l], era);
    if (blockIdx.x == 0 && w == 0) live[(size_t)T * 64 + l] = real;
}

// Bit errors per frame from the decision planes: wrong = erased | (value ^ codeword bit).  A workgroup = (supertile, run of `vpb` variables);
// every lane counts its 32 frames in bit-sliced vertical counters (15 variables ripple into a 4-plane counter, which is added into a
// 12-plane one: vpb <= 4095), the four waves meet in LDS, one atomic per frame and workgroup.
__global__ __launch_bounds__(256) void k_becs_errs(const uint2* __restrict__ xhat, int codeword, int n, int64_t B, int vpb, int32_t* __restrict__ errs) {
    __shared__ uint32_t s_cnt[64][33];
    const int T = blockIdx.y, l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 33; i += 256) (&s_cnt[0][0])[i] = 0u;
    __syncthreads();
    const int vb0 = blockIdx.x * vpb, vb1 = vb0 + vpb < n ? vb0 + vpb : n;
    const uint2* xt = xhat + (size_t)T * n * 64 + l;
    const uint32_t cw = codeword ? ~0u : 0u;
    uint32_t big[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) big[k] = 0u;
    for (int v0 = vb0 + w * 15; v0 < vb1; v0 += 60) {
        uint2 e[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) e[i] = xt[(size_t)(v0 + i < vb1 ? v0 + i : v0) * 64];  // past the end: the first line again, not counted
        uint32_t sm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            uint32_t c = v0 + i < vb1 ? (e[i].x | (e[i].y ^ cw)) : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t t = sm[k] & c;
                sm[k] ^= c;
                c = t;
            }
        }
        uint32_t carry = 0u;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t a = big[k], b = k < 4 ? sm[k] : 0u;
            big[k] = a ^ b ^ carry;
            carry = (a & b) | (carry & (a ^ b));
        }
    }
    for (int b = 0; b < 32; ++b) {
        uint32_t cnt = 0u;
#pragma unroll
        for (int k = 0; k < 12; ++k) cnt |= ((big[k] >> b) & 1u) << k;
        if (cnt) atomicAdd(&s_cnt[l][b], cnt);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) {
        const uint32_t c = s_cnt[i >> 5][i & 31];
        const int64_t f = (int64_t)T * SUPER + i;
        if (c && f < B) atomicAdd(&errs[f], (int32_t)c);
    }
}

// the counters of main.test (src/main.py:41-45) from per-frame bit errors and iteration counts: tot, wec, bec, sweeps, histogram
__global__ __launch_bounds__(256) void k_becs_tally(const int32_t* __restrict__ errs, const int32_t* __restrict__ iters, int64_t B, int hist_bins,
                                                    unsigned long long* __restrict__ counters) {
    extern __shared__ unsigned int s_hist[];  // [hist_bins]
    __shared__ unsigned long long s_sum[4];
    for (int i = threadIdx.x; i < hist_bins; i += 256) s_hist[i] = 0;
    if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned tot = 0, wec = 0, bec = 0, its = 0;
    if (f < B) {
        const int e = errs[f], it = iters[f];
        tot = 1;
        wec = e > 0;
        bec = (unsigned)e;
        its = (unsigned)it;
        if (hist_bins > 0) atomicAdd(&s_hist[it < hist_bins ? it : hist_bins - 1], 1u);
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        tot += __shfl_xor(tot, o);
        wec += __shfl_xor(wec, o);
        bec += __shfl_xor(bec, o);
        its += __shfl_xor(its, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_sum[0], (unsigned long long)tot);
        atomicAdd(&s_sum[1], (unsigned long long)wec);
        atomicAdd(&s_sum[2], (unsigned long long)bec);
        atomicAdd(&s_sum[3], (unsigned long long)its);
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_sum[threadIdx.x]) atomicAdd(&counters[threadIdx.x], s_sum[threadIdx.x]);
    for (int i = threadIdx.x; i < hist_bins; i += 256)
        if (s_hist[i]) atomicAdd(&counters[4 + i], (unsigned long long)s_hist[i]);
}

struct BecsSim {  // ldpc_simulate: where the received word comes from and where the counters go
    uint64_t thr, seed, frame0;
    uint32_t stream;
    int codeword, hist_bins;
    int64_t* counters;
};

template <typename T>
int upload_i32(const std::vector<T>& h, DevBuf* buf) {
    LDPC_TRY(buf->reserve(h.size() * sizeof(T) + 16));
    LDPC_HIP_TRY(hipMemcpy(buf->p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LDPC_OK;
}

}  // namespace

// Batched bec.SPA.decode (src/bec.py:83-122) on the streaming kernels.  y0 [B, n] symbols, xhat [B, n], iters [B] -- device buffers; or, with
// `sim`, channel -> decode -> count on the planes (iters [B] is still needed: scratch of the caller).
static int becs_run(Decoder* d, const uint8_t* y0, const BecsSim* sim, int64_t B, int32_t max_iter, uint32_t flags_in, uint8_t* xhat, int32_t* iters,
                    hipStream_t st) {
    const Code* c = d->code;
    const int n = c->n, m = c->m;
    const int64_t E = c->E;
    if (c->max_dv > 64) {
        set_error("streaming erasure decoder supports variable degrees up to 64 (max_dv=%d)", c->max_dv);
        return LDPC_E_UNSUPPORTED;
    }
    const bool early = !(flags_in & FLAG_NO_EARLY_EXIT);
    const int tiles = (int)((B + SUPER - 1) / SUPER);
    // graph in variable-major order, once per decoder: edge_vpos[k] = position of row-major edge k in the CSC list, chk_of_pos
    if (!d->scratch.p) {
        std::vector<int32_t> idx((size_t)2 * E);
        for (int v = 0; v < n; ++v)
            for (int p = c->col_ptr[v]; p < c->col_ptr[v + 1]; ++p) {
                const int k = c->col_edge[p];
                idx[(size_t)k] = p;                        // edge_vpos
                idx[(size_t)E + p] = c->edge_chk[k];       // chk_of_pos
            }
        LDPC_TRY(upload_i32(idx, &d->scratch));
    }
    const int32_t* edge_vpos = (const int32_t*)d->scratch.p;
    const int32_t* chk_of_pos = edge_vpos + E;
    LDPC_TRY(d->msg.reserve((size_t)tiles * E * 64 * 8));
    LDPC_TRY(d->marg.reserve((size_t)tiles * m * 64 * 8));
    LDPC_TRY(d->prior.reserve((size_t)tiles * n * 64 * 8));
    LDPC_TRY(d->xbits.reserve((size_t)tiles * n * 64 * 8));
    LDPC_TRY(d->live.reserve((size_t)tiles * 64 * 4));
    LDPC_TRY(d->flags.reserve((size_t)tiles * 2 * 64 * 4 + 64));
    uint2* v2c = (uint2*)d->msg.p;
    uint2* sum = (uint2*)d->marg.p;
    uint2* prior = (uint2*)d->prior.p;
    uint2* xh = (uint2*)d->xbits.p;
    uint32_t* live = (uint32_t*)d->live.p;
    uint32_t* tflags = (uint32_t*)d->flags.p;
    int* live_tiles = (int*)((char*)d->flags.p + (size_t)tiles * 2 * 64 * 4);
    volatile int* poll_host = (volatile int*)d->pinned;

    LDPC_HIP_TRY(hipMemsetAsync(tflags, 0, (size_t)tiles * 2 * 64 * 4 + 64, st));
    LDPC_HIP_TRY(hipMemsetAsync(iters, 0, (size_t)B * sizeof(int32_t), st));
    if (sim) {
        const int bpf = (n + 3) / 4, qpw = 8;
        hipLaunchKernelGGL(k_becs_channel, dim3((unsigned)((bpf + 4 * qpw - 1) / (4 * qpw)), tiles), dim3(256), 0, st, sim->thr, sim->codeword, sim->seed, sim->stream,
                           sim->frame0, B, n, bpf, qpw, prior, xh, live, tflags);
    } else {
        hipLaunchKernelGGL(k_becs_load, dim3((n + 63) / 64, tiles), dim3(256), 0, st, y0, B, n, prior, xh, live, tflags);
    }

    const int cpw = 4, vpw = 8;  // nodes per wave task: short runs keep a supertile's summary lines on chip between the two passes
    const int cn_chunks = (m + cpw - 1) / cpw, vn_chunks = (n + vpw - 1) / vpw;
    const int cap = max_iter > 0 ? max_iter : 100000;  // max_iter <= 0 == unlimited upstream (src/bec.py:96); bounded here
    const int poll_every = 4;
    int sweeps = 0;
    bool all_left = false;
    for (int it = 0; it < cap && !all_left; ++it) {
        if (early) {
            const bool poll = (it % poll_every) == 0;
            if (poll) LDPC_HIP_TRY(hipMemsetAsync(live_tiles, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_becs_check, dim3(tiles), dim3(64), 0, st, tflags, live, iters, live_tiles, B, sweeps);
            if (poll) {
                LDPC_HIP_TRY(hipMemcpyAsync((void*)poll_host, live_tiles, sizeof(int), hipMemcpyDeviceToHost, st));
                LDPC_HIP_TRY(hipStreamSynchronize(st));
                if (poll_host[0] == 0) {
                    all_left = true;
                    break;
                }
            }
        }
        const bool first = it == 0;
        hipLaunchKernelGGL(k_becs_cn, dim3((unsigned)(((int64_t)tiles * cn_chunks + 3) / 4)), dim3(256), 0, st, c->d_row_ptr,
                           first ? c->d_edge_var : edge_vpos, first ? prior : v2c, sum, live, m, first ? (int64_t)n : E, tiles, cn_chunks, cpw);
        const dim3 vgrid((unsigned)(((int64_t)tiles * vn_chunks + 3) / 4));
#define LDPC_BECS_VN1(DVM, FIRST) \
    hipLaunchKernelGGL((k_becs_vn<DVM, FIRST>), vgrid, dim3(256), 0, st, c->d_col_ptr, chk_of_pos, sum, v2c, prior, xh, live, tflags, n, m, E, tiles, vn_chunks, vpw)
#define LDPC_BECS_VN(DVM)                   \
    do {                                    \
        if (first) LDPC_BECS_VN1(DVM, true); \
        else LDPC_BECS_VN1(DVM, false);      \
    } while (0)
        if (c->max_dv <= 3) LDPC_BECS_VN(3);
        else if (c->max_dv <= 4) LDPC_BECS_VN(4);
        else if (c->max_dv <= 8) LDPC_BECS_VN(8);
        else if (c->max_dv <= 16) LDPC_BECS_VN(16);
        else if (c->max_dv <= 32) LDPC_BECS_VN(32);
        else LDPC_BECS_VN(64);
#undef LDPC_BECS_VN
#undef LDPC_BECS_VN1
        ++sweeps;
    }
    hipLaunchKernelGGL(k_becs_finish, dim3(tiles), dim3(64), 0, st, live, iters, B, sweeps);
    if (sim) {
        LDPC_TRY(d->h_out.reserve((size_t)B * sizeof(int32_t)));  // bit errors per frame
        int32_t* errs = (int32_t*)d->h_out.p;
        LDPC_HIP_TRY(hipMemsetAsync(errs, 0, (size_t)B * sizeof(int32_t), st));
        const int vpb = 1020;
        hipLaunchKernelGGL(k_becs_errs, dim3((n + vpb - 1) / vpb, tiles), dim3(256), 0, st, xh, sim->codeword, n, B, vpb, errs);
        hipLaunchKernelGGL(k_becs_tally, dim3((unsigned)((B + 255) / 256)), dim3(256), (size_t)sim->hist_bins * sizeof(unsigned int), st, errs, iters, B,
                           sim->hist_bins, (unsigned long long*)sim->counters);
    } else {
        hipLaunchKernelGGL(k_becs_unpack, dim3((n + 63) / 64, tiles), dim3(256), 0, st, xh, xhat, B, n);
    }
    LDPC_HIP_TRY(hipGetLastError());
    d->last_repacks = 0;
    d->last_sweeps = sweeps;
    d->last_backend = BK_STREAM;
    return LDPC_OK;
}

int becs_stream_decode(Decoder* d, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags, uint8_t* xhat, int32_t* iters, hipStream_t st) {
    return becs_run(d, y0, nullptr, B, max_iter, flags, xhat, iters, st);
}

// ldpc_simulate on the streaming erasure decoder: frames [frame0, frame0 + B) of the Philox stream, counters accumulated
int becs_stream_simulate(Decoder* d, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B, int32_t max_iter,
                         uint32_t flags, int32_t hist_bins, int64_t* counters, hipStream_t st) {
    if (!(param >= 0.0 && param <= 1.0)) {
        set_error("channel probability %g outside [0,1]", param);
        return LDPC_E_ARG;
    }
    if (hist_bins > 8192) {
        set_error("at most 8192 histogram bins");
        return LDPC_E_ARG;
    }
    double t = ceil(param * 4294967296.0 - 0.5);  // (w + 0.5) * 2^-32 < p  <=>  w < ceil(p * 2^32 - 0.5), as channel_generate
    if (t < 0) t = 0;
    BecsSim sim{(uint64_t)t, seed, frame0, (uint32_t)stream_id, codeword, hist_bins, counters};
    LDPC_TRY(d->h_iters.reserve((size_t)B * sizeof(int32_t)));
    return becs_run(d, nullptr, &sim, B, max_iter, flags, nullptr, (int32_t*)d->h_iters.p, st);
}

}  // namespace ldpc
