// Streaming backend: flooding BP with the message state resident in HBM.
//
// Layout per tile of 64 frames (lane == frame):
//     c2v  [tile][E][64]   T     check -> variable messages, row-major edge order (the order of np.where(H), src/bpa.py:12)
//     marg [tile][n][64]   T     marginals  prior + ordered sum of c2v  (src/bpa.py:35); also the soft output
//     prior[tile][n][64]   T
//     xbits[tile/8][n][8]  u64   hard decision of variable v for the 64 frames of a tile (bit f == frame f); the words of EIGHT
//                                consecutive tiles sit in one 64-byte sector, which is what the syndrome kernel fetches per variable
//     live [tile]          u64   frames that are still iterating
// Every H index is wave-uniform (scalar loads); every message access is one contiguous 64-element line.
//
// One sweep = two passes (round 3: the variable -> check messages are no longer stored):
//   check pass     per check: stream its old c2v lines in, gather the marginal lines of its variables, v2c = marg - c2v_old
//                  (src/bpa.py:37 -- the same subtraction on the same operands, so the same bits), apply the rule, stream c2v out.
//                  Every store of the sweep's E-sized traffic is a contiguous stream; the only random accesses are READS of
//                  marginal lines, each used by dv checks (re-reads are served by the L2 / Infinity Cache).
//   variable pass  per variable: gather its c2v lines (read only), marg = prior + (((0 + c_a) + c_b) + ...), stream marg out,
//                  decision bit-plane.
// HBM bytes per frame-sweep: check pass reads E + n..E, writes E; variable pass reads E + n, writes n  --  3E + 3n when the marginal
// re-reads hit on chip, against the s(4E + n) of SURVEY.md section 8(d) (v2c written and read back); bench.py prices the kernels
// with the section-8(d) ALGORITHMIC bytes and reports the PMC traffic beside it.
//
// Reference semantics reproduced (file:line relative to thadikari/ldpc_decoders):
//   flooding loop, max_iter and syndrome exits, x_hat = (marginal < 0) ...... src/bpa.py:17-63
//   iteration-0 check of the received word (BSC) ............................ src/bpa.py:20,29
//   variable update: prior + (((0 + c_a) + c_b) + ...) in ascending edge order  src/bpa.py:35, src/math_utils.py:7
// (The erasure decoder of src/bec.py:83-122 has its own bit-sliced streaming kernels: ldpc_bec_stream.hip.)
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <string>

#include "ldpc_cn.hpp"
#include "ldpc_common.hpp"
#include "ldpc_repack.hpp"
#include "ldpc_rng.hpp"

namespace ldpc {

namespace {

using u64 = unsigned long long;

// streaming (non-temporal) access, selectable per pass and per direction (measured): the check pass streams
// its c2v lines through with non-temporal loads AND stores so that they do not push the marginal lines (re-used dv times) out of
// the caches
constexpr bool CN_NTL = true, CN_NTS = true, VN_NTL = false, VN_NTS = false;
template <bool NT, typename T> __device__ __forceinline__ T msg_ld(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT, typename T> __device__ __forceinline__ void msg_st(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

__device__ __forceinline__ u64 wave_or(u64 x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(x & 0xffffffffull), off, 64);
        const unsigned hi = __shfl_xor((unsigned)(x >> 32), off, 64);
        x |= ((u64)hi << 32) | lo;
    }
    return x;
}

// Wave task -> (tile, chunk of nodes).  XCD-aware (MI355X_MICROARCH.md: block b runs on XCD b % 8, each XCD has its own L2): the
// blocks of ONE tile all carry the same b % 8, so the marginal lines a check pass re-reads (each is used by dv checks of the tile)
// are re-read through ONE L2 instead of up to dv different ones.  Grid = 8 x ceil(tiles / 8) x blocks per tile (4 waves per block).
__device__ __forceinline__ bool task_of(int tiles, int chunks, int xcd_aware, int* tile, int* chunk) {
    const int q = blockIdx.x, w = threadIdx.y;
    if (xcd_aware) {
        const int bpt = (chunks + 3) >> 2;
        const int x = q & 7, local = q >> 3;
        const int tl = local / bpt, cb = local - tl * bpt;
        *tile = __builtin_amdgcn_readfirstlane(tl * 8 + x);
        *chunk = __builtin_amdgcn_readfirstlane(cb * 4 + w);
    } else {
        const int task = q * 4 + w;
        *tile = __builtin_amdgcn_readfirstlane(task / chunks);
        *chunk = __builtin_amdgcn_readfirstlane(task - (task / chunks) * chunks);
    }
    return *tile < tiles && *chunk < chunks;
}
__host__ inline int task_blocks(int tiles, int chunks, int xcd_aware) {
    return xcd_aware ? 8 * ((tiles + 7) / 8) * ((chunks + 3) / 4) : (int)(((long)tiles * chunks + 3) / 4);
}

// index of the bit-plane word of (tile, variable): eight tiles interleaved per variable
// Wave-uniform 8-byte read through the SCALAR cache (constant address space: the compiler emits s_load_dwordx2, counted by lgkmcnt and
// therefore independent of the vector loads and stores in flight).  Only for words no wave writes before this wave has read them in the
// same launch: the decision word of (tile, variable) is read and then written by exactly one wave per sweep.
__device__ __forceinline__ u64 uniform_ld64(const u64* p) {
    typedef const __attribute__((address_space(4))) u64* cptr;
    return *(cptr)(uintptr_t)p;
}
__host__ __device__ __forceinline__ int64_t plane_at(int tile, int64_t v, int n) { return ((int64_t)(tile >> 3) * n + v) * 8 + (tile & 7); }
__host__ inline size_t plane_words(int tiles, int n) { return (size_t)((tiles + 7) / 8) * n * 8; }

// ---------------------------------------------------------------------------------------------------
// priors [B,n] (frame-major, as numpy hands them over) -> prior tile [n][64]; optional hard word y0 -> planes.
template <typename T>
__global__ __launch_bounds__(256) void k_load_tile(const T* __restrict__ priors, const uint8_t* __restrict__ y0, int64_t B,
                                                   int n, T* __restrict__ prior_t, u64* __restrict__ xbits) {
    __shared__ T sp[64][65];
    __shared__ uint8_t sy[64][68];
    const int tile = blockIdx.y, v0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int f = ty; f < 64; f += 4) {
        const int64_t fr = (int64_t)tile * 64 + f;
        const int v = v0 + tx;
        T val = T(0);
        uint8_t yy = 0;
        if (fr < B && v < n) {
            if (y0) yy = y0[fr * n + v];
            val = priors[fr * n + v];
        }
        sp[f][tx] = val;
        sy[f][tx] = yy;
    }
    __syncthreads();
    for (int vv = ty; vv < 64; vv += 4) {
        const int v = v0 + vv;
        if (v < n) {
            prior_t[((int64_t)tile * n + v) * 64 + tx] = sp[tx][vv];
            if (y0) {
                const u64 one = __ballot(sy[tx][vv] != 0);
                if (tx == 0) xbits[plane_at(tile, v, n)] = one;
            }
        }
    }
}

// BI-AWGN channel + LLR written STRAIGHT into the prior tiles (ldpc_simulate on the streaming kernels): the Philox block, Box-Muller and
// LLR expressions of k_biawgn (ldpc_channel.hip; src/biawgn.py:10-28) -- bit-identical priors -- without the [B,n] staging array and the
// transposing tile load.  Lane == frame: one Philox block (4 consecutive variables) per lane and step, four coalesced lines per wave.
struct SimSource {
    double sigma, inv_var2;
    int codeword;
    uint64_t seed, frame0;
    uint32_t stream;
};
template <typename T>
__global__ __launch_bounds__(256) void k_biawgn_tile(SimSource s, int64_t B, int n, int bpf, int blocks_per_wave, T* __restrict__ prior_t) {
    const int lane = threadIdx.x, tile = blockIdx.y;
    const int64_t fr = (int64_t)tile * 64 + lane;
    if (fr >= B) return;
    const int j0 = (blockIdx.x * 4 + threadIdx.y) * blocks_per_wave;
    const T sg = (T)s.sigma, k = (T)s.inv_var2, mean = (T)(2 * s.codeword - 1);
    T* pt = prior_t + (int64_t)tile * n * 64 + lane;
    for (int j = j0; j < min(bpf, j0 + blocks_per_wave); ++j) {
        const Philox4 p = philox_word_block(s.seed, s.stream, s.frame0 + (uint64_t)fr, (uint32_t)j);
        T z[4];
        box_muller<T>(p.w[0], p.w[1], z[0], z[1]);
        box_muller<T>(p.w[2], p.w[3], z[2], z[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const T y = mean + sg * z[q];
            if (4 * j + q < n) pt[(int64_t)(4 * j + q) * 64] = -(k * y);  // -2y/sigma^2 with k = 2/sigma^2
        }
    }
}

__global__ void k_init_live(u64* __restrict__ live, int64_t B, int tiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const int64_t rem = B - (int64_t)t * 64;
    live[t] = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
}

// ---------------------------------------------------------------------------------------------------
// Check pass.  One wavefront = (tile, contiguous run of checks); UNR checks are in flight together: UNR*dc old-message lines (a
// stream) and UNR*dc marginal lines (gathers) outstanding per wave.  `src` = the marginals, or the priors in the first sweep, when
// the reference's v2c is the prior itself (src/bpa.py:19) and there is no old message to subtract.
template <typename T, int ALG>
__device__ __forceinline__ T v2c_of(T marg, T c_old) {
    return marg - c_old;  // src/bpa.py:37
}

// GATHER (the frame repack folded into the sweep that follows it, see run()): `tile` is a tile of the DENSE destination tiling; lane j
// reads its old messages and marginals from the (source tile, source lane) of the j-th live frame (srcmap) -- `c2v_in` / `src` are the
// SOURCE set -- and the new messages are stored as whole lines of the destination set `c2v`.  The lanes of a destination tile cover a
// contiguous run of source positions, so every source sector is fetched by one destination tile (two at a run's ends).
template <typename T, int ALG, int DCMAX, int FIXED_DC, int UNR, bool GATHER = false>
__global__ __launch_bounds__(256) void k_cn(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_var,
                                            T* __restrict__ c2v, const T* __restrict__ src,
                                            const u64* __restrict__ live, int m, int n, int64_t E, int tiles, int chunks,
                                            int cpw, int first, int xcd_aware, int freeze, const T* __restrict__ c2v_in = nullptr,
                                            const int32_t* __restrict__ srcmap = nullptr) {
    const int lane = threadIdx.x;
    int tile, chunk;
    if (!task_of(tiles, chunks, xcd_aware, &tile, &chunk)) return;
    const u64 lv = live[tile];
    if (lv == 0) return;
    // A tile with a live frame is processed by ALL its lanes: the messages of a frame that has left are never looked at again, and a
    // store that skips some lanes is a partial line -- sub-sector writes, measured 47 % slower passes at 90 % live lanes (n = 64 800:
    // 16.6 against 11.3 ms per check pass, profiles/r05_config5_timeline.txt).  Only a decode that returns the soft output freezes the
    // lanes of departed frames (`freeze`): their marginals must stay those of their own last sweep.
    const bool on = freeze ? (bool)((lv >> lane) & 1ull) : true;
    const bool dense = E * 4 >= (int64_t)m * DCMAX * 3;  // average row length at least three quarters of DCMAX (wave-uniform)
    T* ct = c2v + (int64_t)tile * E * 64 + lane;
    const T* ci = ct;                                    // old messages in
    const T* st = src + (int64_t)tile * n * 64 + lane;
    if constexpr (GATHER) {
        int sm = srcmap[(int64_t)tile * 64 + lane];
        if (sm < 0) sm = srcmap[(int64_t)tile * 64];      // a lane beyond the live frames (tail of the last tile) shadows lane 0: defined values, never looked at
        ci = c2v_in + (int64_t)(sm >> 6) * E * 64 + (sm & 63);
        st = src + (int64_t)(sm >> 6) * n * 64 + (sm & 63);
    }
    const int c_end = min(m, (chunk + 1) * cpw);
    for (int c = chunk * cpw; c < c_end; c += UNR) {
        T v[UNR][DCMAX], o[UNR][DCMAX];
        int k0[UNR], deg[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int cc = c + u;
            if (cc < c_end) {
                if constexpr (FIXED_DC > 0) {
                    k0[u] = cc * FIXED_DC;
                    deg[u] = FIXED_DC;
                } else {
                    k0[u] = row_ptr[cc];
                    deg[u] = row_ptr[cc + 1] - k0[u];
                }
            } else {
                k0[u] = 0;
                deg[u] = 0;
            }
        }
        if (on) {
            // irregular rows, `dense` (most rows nearly DCMAX long): every line is fetched unconditionally (a short row re-reads its
            // last edge, an empty row edge 0) -- a branch per line keeps the loads of a group of checks from being issued together;
            // with widely spread row lengths the branch stays
            if (FIXED_DC > 0 || dense) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
#pragma unroll
                    for (int j = 0; j < DCMAX; ++j) {
                        const int kk = FIXED_DC > 0 ? k0[u] + j : (deg[u] > 0 ? k0[u] + (j < deg[u] ? j : deg[u] - 1) : 0);
                        v[u][j] = st[(int64_t)edge_var[kk] * 64];
                        if (!first) o[u][j] = msg_ld<CN_NTL>(ci + (int64_t)kk * 64);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
#pragma unroll
                    for (int j = 0; j < DCMAX; ++j) {
                        if (j < deg[u]) {
                            v[u][j] = st[(int64_t)edge_var[k0[u] + j] * 64];
                            if (!first) o[u][j] = msg_ld<CN_NTL>(ci + (int64_t)(k0[u] + j) * 64);
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!first) {
#pragma unroll
                    for (int j = 0; j < DCMAX; ++j) v[u][j] = v2c_of<T, ALG>(v[u][j], o[u][j]);
                }
                cn_rule<T, ALG, DCMAX>(v[u], deg[u]);
#pragma unroll
                for (int j = 0; j < DCMAX; ++j) {
                    if (j < deg[u]) msg_st<CN_NTS>(ct + (int64_t)(k0[u] + j) * 64, v[u][j]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Variable pass: marginal = prior + ordered sum of c2v (c2v is only read) ; decision bit.
// FIXED_DV > 0: every variable has exactly that many edges (edge list of variable v at v * FIXED_DV): no col_ptr loads and no branch
// per line, so the lines of a group of variables are fetched together
// GATHER (second half of the folded repack): messages, marginals and decision words are those of the destination tiling already; the
// priors are still read from the source position of each lane's frame (srcmap) and copied into the destination set on the way.
template <typename T, int ALG, int DVMAX, int UNR, int FIXED_DV, bool GATHER = false>
__global__ __launch_bounds__(256) void k_vn(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ col_edge,
                                            const T* __restrict__ c2v, const T* __restrict__ prior_t, T* __restrict__ marg_t,
                                            const u64* __restrict__ live, u64* __restrict__ xbits,
                                            int n, int64_t E, int tiles, int chunks, int vpw, int xcd_aware, int freeze,
                                            T* __restrict__ prior_out = nullptr, const int32_t* __restrict__ srcmap = nullptr) {
    const int lane = threadIdx.x;
    int tile, chunk;
    if (!task_of(tiles, chunks, xcd_aware, &tile, &chunk)) return;
    const u64 lv = live[tile];
    if (lv == 0) return;
    const bool on = freeze ? (bool)((lv >> lane) & 1ull) : true;  // (see k_cn; the decision words below keep the bits of departed frames either way)
    const T* ct = c2v + (int64_t)tile * E * 64 + lane;
    const T* pt = prior_t + (int64_t)tile * n * 64 + lane;
    T* po = nullptr;
    if constexpr (GATHER) {
        int sm = srcmap[(int64_t)tile * 64 + lane];
        if (sm < 0) sm = srcmap[(int64_t)tile * 64];
        pt = prior_t + (int64_t)(sm >> 6) * n * 64 + (sm & 63);
        po = prior_out + (int64_t)tile * n * 64 + lane;
    }
    T* mt = marg_t + (int64_t)tile * n * 64 + lane;
    u64* xb = xbits + plane_at(tile, 0, n);  // word of variable v at xb[8 * v]
    const int v_end = min(n, (chunk + 1) * vpw);
    for (int vbase = chunk * vpw; vbase < v_end; vbase += UNR) {
        T c[UNR][DVMAX];
        T pr[UNR];
        u64 old[UNR];
        int p0[UNR], deg[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int vv = vbase + u;
            // decisions of frames that have left, read ahead of the lines (a read behind this variable's marginal store would wait for it)
            if (lv != ~0ull) old[u] = GATHER ? 0ull : uniform_ld64(xb + 8 * (vv < v_end ? vv : v_end - 1));  // (GATHER: the destination words hold nothing yet)
            if constexpr (FIXED_DV > 0) {
                p0[u] = (vv < v_end ? vv : v_end - 1) * FIXED_DV;  // past the end: the last variable's lines once more, result unused
                deg[u] = vv < v_end ? FIXED_DV : -1;
            } else if (vv < v_end) {
                p0[u] = col_ptr[vv];
                deg[u] = col_ptr[vv + 1] - p0[u];
            } else {
                p0[u] = 0;
                deg[u] = -1;
            }
        }
        if (on) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if constexpr (FIXED_DV > 0) {
                    pr[u] = pt[(int64_t)(vbase + u < v_end ? vbase + u : v_end - 1) * 64];
#pragma unroll
                    for (int j = 0; j < FIXED_DV; ++j) c[u][j] = msg_ld<VN_NTL>(ct + (int64_t)col_edge[p0[u] + j] * 64);
                } else {
                    if (deg[u] >= 0) pr[u] = pt[(int64_t)(vbase + u) * 64];
#pragma unroll
                    for (int j = 0; j < DVMAX; ++j) {
                        if (j < deg[u]) c[u][j] = msg_ld<VN_NTL>(ct + (int64_t)col_edge[p0[u] + j] * 64);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (deg[u] < 0) continue;  // wave-uniform
            const int du = FIXED_DV > 0 ? FIXED_DV : deg[u];
            bool b_one = false;
            if (on) {
                T s = T(0);
#pragma unroll
                for (int j = 0; j < DVMAX; ++j)
                    if (j < du) s += c[u][j];
                const T marg = pr[u] + s;
                msg_st<VN_NTS>(mt + (int64_t)(vbase + u) * 64, marg);
                if constexpr (GATHER) po[(int64_t)(vbase + u) * 64] = pr[u];
                b_one = marg < T(0);  // NaN marginal -> 0 (src/bpa.py:38,62)
            }
            const u64 one = __ballot(b_one);
            const int vv = vbase + u;
            u64 merged = one;
            if (lv != ~0ull) merged = (old[u] & ~lv) | (one & lv);
            if (lane == 0) xb[8 * vv] = merged;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Syndrome of the current hard decisions; frames whose syndrome is zero leave (iters = sweeps run so far).
// Two kernels: k_syndrome_part -- a block takes a GROUP of eight tiles and a chunk of the checks; a thread fetches, per edge of its
// check, the one 64-byte sector that holds the decision words of the variable for all eight tiles (per-tile 8-byte gathers made this
// kernel L2-bound: 0.92 ms per sweep at n = 64 800, 6 % of the sweep), XORs them into eight parities, and the block merges the OR of
// its checks into the tiles' words -- and k_syndrome_fin, one wave per tile, which retires the frames.
__global__ __launch_bounds__(256) void k_syndrome_part(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_var,
                                                       const u64* __restrict__ xbits, const u64* __restrict__ live,
                                                       u64* __restrict__ unsat_acc, int m, int n, int tiles, int checks_per_block) {
    const int grp = blockIdx.y, t = threadIdx.x;
    const int t0 = grp * 8, nt = min(8, tiles - t0);
    u64 any_live = 0;
    for (int i = 0; i < nt; ++i) any_live |= live[t0 + i];
    if (any_live == 0) return;
    const ulonglong2* xb = reinterpret_cast<const ulonglong2*>(xbits + (int64_t)grp * n * 8);  // variable v: xb[4 v .. 4 v + 3]
    const int c0 = blockIdx.x * checks_per_block, c1 = min(m, c0 + checks_per_block);
    u64 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = c0 + t; c < c1; c += 256) {
        u64 par[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = row_ptr[c]; k < row_ptr[c + 1]; ++k) {
            const ulonglong2* w = xb + (int64_t)edge_var[k] * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const ulonglong2 x = w[q];
                par[2 * q] ^= x.x;
                par[2 * q + 1] ^= x.y;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] |= par[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u64 a = wave_or(acc[i]);
        if ((t & 63) == 0 && a && i < nt) atomicOr(&unsat_acc[2 * (t0 + i)], a);
    }
}

__global__ __launch_bounds__(64) void k_syndrome_fin(u64* __restrict__ unsat_acc, u64* __restrict__ live, int32_t* __restrict__ iters,
                                                     int* __restrict__ live_tiles, int64_t B, int sweeps,
                                                     const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.x, t = threadIdx.x;
    const u64 lv = live[tile];
    if (lv == 0) return;
    const u64 unsat = unsat_acc[2 * tile];
    const u64 stay = lv & unsat, leave = lv & ~unsat;
    __syncthreads();  // every lane has read the word before lane 0 clears it for the next sweep
    if (t == 0) {
        unsat_acc[2 * tile] = 0;
        live[tile] = stay;
        if (stay && live_tiles) {
            atomicAdd(live_tiles, 1);                    // tiles that still hold a live frame
            atomicAdd(live_tiles + 1, __popcll(stay));   // live frames
        }
    }
    if ((leave >> t) & 1ull) {
        const int64_t fr = frame_of ? (int64_t)frame_of[(int64_t)tile * 64 + t] : (int64_t)tile * 64 + t;
        if (fr >= 0 && fr < B) iters[fr] = sweeps;
    }
}

// ---------------------------------------------------------------------------------------------------
// Frame repack (early termination, ldpc_repack.hpp): when many tiles hold only a few live frames, the live frames are gathered
// into dense tiles.  k_repack moves
// message lines, marginals, priors and decision bit-planes: destination lane j reads (source tile, source lane) of the j-th live
// frame; lanes that share a source tile share the 256-byte line, so a line of a source tile is fetched once per destination tile
// that draws from it.  The decisions of every frame of the old tiles are written out before (k_unpack), the moved frames
// overwrite theirs at the end.
template <typename T>
__global__ __launch_bounds__(256) void k_repack(const T* __restrict__ msg_src, T* __restrict__ msg_dst, const T* __restrict__ marg_src,
                                                T* __restrict__ marg_dst, const T* __restrict__ prior_src, T* __restrict__ prior_dst,
                                                const u64* __restrict__ xb_src, u64* __restrict__ xb_dst,
                                                const u64* __restrict__ live_src, u64* __restrict__ live_dst,
                                                const int32_t* __restrict__ base, const int32_t* __restrict__ frame_src,
                                                int32_t* __restrict__ frame_dst, int tiles_src, int n, int64_t E, int rows_per_wave) {
    const int lane = threadIdx.x;
    const int dt = blockIdx.y;  // destination tile
    int st, sl;
    const bool has = repack_source(base, live_src, tiles_src, dt * 64 + lane, &st, &sl);
    const int chunk = blockIdx.x * 4 + threadIdx.y;
    const int64_t rows = E + n;  // message lines, then per variable: marginal + prior line + one bit-plane word
    const int64_t r0 = (int64_t)chunk * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    const T* ms = msg_src + (int64_t)st * E * 64 + sl;
    T* md = msg_dst + (int64_t)dt * E * 64 + lane;
    const int64_t so = (int64_t)st * n * 64 + sl, dof = (int64_t)dt * n * 64 + lane;
    for (int64_t r = r0; r < r1; ++r) {
        // lanes beyond the live frames (the tail of the last destination tile) get zeros, not whatever the buffer held: every lane of a
        // live tile computes and stores in the passes that follow, and whole-line stores are the cheaper ones anyway
        if (r < E) {
            md[r * 64] = has ? ms[r * 64] : T(0);
        } else {
            const int64_t v = r - E;
            prior_dst[dof + v * 64] = has ? prior_src[so + v * 64] : T(0);
            marg_dst[dof + v * 64] = has ? marg_src[so + v * 64] : T(0);
            const u64 w = has ? xb_src[plane_at(st, v, n)] : 0ull;
            const u64 plane = __ballot(has && ((w >> sl) & 1ull));
            if (lane == 0) xb_dst[plane_at(dt, v, n)] = plane;
        }
    }
    if (chunk == 0) {
        frame_dst[(int64_t)dt * 64 + lane] = has ? (frame_src ? frame_src[(int64_t)st * 64 + sl] : st * 64 + sl) : -1;
        const u64 lv = __ballot(has);
        if (lane == 0) live_dst[dt] = lv;
    }
}

// The repack FOLDED into the next sweep (LLR decoders, node degrees up to 8): instead of copying E + 2n lines per tile, only the map is
// built -- srcmap[dt][lane] = source tile * 64 + source lane of the frame that destination lane will hold (-1 beyond the live frames),
// the frame indices and the live words of the destination tiling -- and the sweep that follows runs the GATHER variants of the two passes:
// they read the old state through the map and write the new state densely into the other buffer set.  A separate repack read all of the
// source and wrote the live part (13.7 ms at n = 64 800 / 59 % live) before a sweep that read and wrote the live part again; folded, the
// sweep reads the source once and writes the live part once.
__global__ __launch_bounds__(64) void k_repack_map(const u64* __restrict__ live_src, u64* __restrict__ live_dst, const int32_t* __restrict__ base,
                                                   const int32_t* __restrict__ frame_src, int32_t* __restrict__ frame_dst,
                                                   int32_t* __restrict__ srcmap, int tiles_src) {
    const int lane = threadIdx.x, dt = blockIdx.x;
    int st, sl;
    const bool has = repack_source(base, live_src, tiles_src, dt * 64 + lane, &st, &sl);
    srcmap[(int64_t)dt * 64 + lane] = has ? st * 64 + sl : -1;
    frame_dst[(int64_t)dt * 64 + lane] = has ? (frame_src ? frame_src[(int64_t)st * 64 + sl] : st * 64 + sl) : -1;
    const u64 lv = __ballot(has);
    if (lane == 0) live_dst[dt] = lv;
}

// Frames that hit max_iter: iters = sweeps ; then planes -> x_hat bytes [B,n] in {0,1}.
__global__ void k_finish_iters(const u64* __restrict__ live, int32_t* __restrict__ iters, int64_t B, int sweeps,
                               const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.x, t = threadIdx.x;
    const u64 lv = live[tile];
    const int64_t fr = frame_of ? (int64_t)frame_of[(int64_t)tile * 64 + t] : (int64_t)tile * 64 + t;
    if (((lv >> t) & 1ull) && fr >= 0 && fr < B) iters[fr] = sweeps;
}

// marginal tile [n][64] -> [B,n] (diagnostic / soft-output path; not on the throughput path)
template <typename T>
__global__ void k_soft_out(const T* __restrict__ soft_t, T* __restrict__ out, int64_t B, int n) {
    const int tile = blockIdx.y, lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t fr = (int64_t)tile * 64 + lane;
    if (v < n && fr < B) out[fr * n + v] = soft_t[((int64_t)tile * n + v) * 64 + lane];
}

__global__ __launch_bounds__(256) void k_unpack(const u64* __restrict__ xbits, uint8_t* __restrict__ xhat, int64_t B, int n,
                                                const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const u64 one = xbits[plane_at(tile, v, n)];
    const int64_t f0 = (int64_t)tile * 64;
    if (frame_of) {  // repacked tiles: lane f holds frame frame_of[tile][f] (-1: none)
        for (int f = 0; f < 64; ++f) {
            const int64_t fr = frame_of[f0 + f];
            if (fr >= 0 && fr < B) xhat[fr * n + v] = (uint8_t)((one >> f) & 1ull);
        }
        return;
    }
    const int fmax = (int)min((int64_t)64, B - f0);
    for (int f = 0; f < fmax; ++f) xhat[(f0 + f) * n + v] = (uint8_t)((one >> f) & 1ull);
}

// planes -> PACKED decisions [B, W] u32 (bit v & 31 of word v >> 5 = decision of variable v; W = ceil(n / 32)): the form
// ldpc_decode_bits returns (SURVEY 8(a2) "packed x_hat bits").  One wave per (tile, 64 variables): the 64 x 64 bit block is
// transposed with 64 ballots (lane = variable on the way in, lane = frame on the way out).
__global__ __launch_bounds__(256) void k_unpack_bits(const u64* __restrict__ xbits, uint32_t* __restrict__ bits, int64_t B, int n, int W,
                                                     const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.y, lane = threadIdx.x;
    const int vb = blockIdx.x * 4 + threadIdx.y, v0 = vb * 64;
    if (v0 >= n) return;
    const u64 col = v0 + lane < n ? xbits[plane_at(tile, v0 + lane, n)] : 0ull;
    u64 row = 0;
    for (int f = 0; f < 64; ++f) {
        const u64 b = __ballot((col >> f) & 1ull);
        if (lane == f) row = b;
    }
    const int64_t fr = frame_of ? (int64_t)frame_of[(int64_t)tile * 64 + lane] : (int64_t)tile * 64 + lane;
    if (fr >= 0 && fr < B) {
        uint32_t* out = bits + fr * W + 2 * vb;
        out[0] = (uint32_t)row;
        if (2 * vb + 1 < W) out[1] = (uint32_t)(row >> 32);
    }
}

// decisions of the frames of `tiles` tiles in the form the caller asked for: bytes [B,n] (ldpc_decode) or packed words (ldpc_decode_bits)
void launch_unpack(const Decoder* d, const u64* xbits, uint8_t* xhat, int64_t B, int n, int tiles, const int32_t* fmap, hipStream_t st) {
    if (d->out_bits)
        hipLaunchKernelGGL(k_unpack_bits, dim3(((n + 63) / 64 + 3) / 4, tiles), dim3(64, 4), 0, st, xbits, d->out_bits, B, n, (n + 31) / 32, fmap);
    else
        hipLaunchKernelGGL(k_unpack, dim3((n + 255) / 256, tiles), dim3(256), 0, st, xbits, xhat, B, n, fmap);
}

int pick_pow2_ge(int x, int lo, int hi) {
    int p = lo;
    while (p < x && p < hi) p <<= 1;
    return p;
}

// nodes kept in flight per wave: about 32 VGPRs (128 bytes per lane) of messages, at most 4 nodes
constexpr int unroll_for(int row_bytes) {
    int u = 128 / row_bytes;
    u = u > 4 ? 4 : u;
    return u >= 8 ? 8 : (u >= 4 ? 4 : (u >= 2 ? 2 : 1));
}

struct Geometry {
    int tiles, cn_chunks, cpw, vn_chunks, vpw, xcd_aware;
    int freeze = 0;  // lanes of departed frames do not compute or store (soft-output decodes)
};

template <typename T, int ALG, int DCMAX, int FIXED_DC>
void launch_cn(const Code* c, T* c2v, const T* src, const u64* live, const Geometry& g, int first, hipStream_t st) {
    constexpr int UNR = unroll_for(2 * DCMAX * (int)sizeof(T));  // old message + marginal line per edge
    hipLaunchKernelGGL((k_cn<T, ALG, DCMAX, FIXED_DC, UNR>), dim3(task_blocks(g.tiles, g.cn_chunks, g.xcd_aware)), dim3(64, 4), 0, st, c->d_row_ptr,
                       c->d_edge_var, c2v, src, live, c->m, c->n, c->E, g.tiles, g.cn_chunks, g.cpw, first, g.xcd_aware, g.freeze);
}

template <typename T, int ALG, int DVMAX, int FIXED_DV = 0>
void launch_vn(const Code* c, const T* c2v, const T* prior, T* marg, const u64* live, u64* xbits, const Geometry& g, hipStream_t st) {
    constexpr int UNR = unroll_for((DVMAX + 1) * (int)sizeof(T));
    hipLaunchKernelGGL((k_vn<T, ALG, DVMAX, UNR, FIXED_DV>), dim3(task_blocks(g.tiles, g.vn_chunks, g.xcd_aware)), dim3(64, 4), 0, st, c->d_col_ptr, c->d_col_edge,
                       c2v, prior, marg, live, xbits, c->n, c->E, g.tiles, g.vn_chunks, g.vpw, g.xcd_aware, g.freeze);
}

// GATHER variants of the two passes (folded repack): built for the node degrees the LDPC ensembles of the reference have (dc <= 8, dv <= 8)
template <typename T, int ALG, int DCMAX, int FIXED_DC>
void launch_cn_gather(const Code* c, T* c2v, const T* c2v_in, const T* src, const u64* live, const int32_t* srcmap, const Geometry& g, hipStream_t st) {
    constexpr int UNR = unroll_for(2 * DCMAX * (int)sizeof(T));
    hipLaunchKernelGGL((k_cn<T, ALG, DCMAX, FIXED_DC, UNR, true>), dim3(task_blocks(g.tiles, g.cn_chunks, g.xcd_aware)), dim3(64, 4), 0, st, c->d_row_ptr,
                       c->d_edge_var, c2v, src, live, c->m, c->n, c->E, g.tiles, g.cn_chunks, g.cpw, 0, g.xcd_aware, 0, c2v_in, srcmap);
}
template <typename T, int ALG, int DVMAX, int FIXED_DV = 0>
void launch_vn_gather(const Code* c, const T* c2v, const T* prior_in, T* prior_out, T* marg, const u64* live, u64* xbits, const int32_t* srcmap,
                      const Geometry& g, hipStream_t st) {
    constexpr int UNR = unroll_for((DVMAX + 1) * (int)sizeof(T));
    hipLaunchKernelGGL((k_vn<T, ALG, DVMAX, UNR, FIXED_DV, true>), dim3(task_blocks(g.tiles, g.vn_chunks, g.xcd_aware)), dim3(64, 4), 0, st, c->d_col_ptr,
                       c->d_col_edge, c2v, prior_in, marg, live, xbits, c->n, c->E, g.tiles, g.vn_chunks, g.vpw, g.xcd_aware, 0, prior_out, srcmap);
}
inline bool gather_passes_built(const Code* c) { return c->max_dc <= 8 && c->max_dv <= 8; }
template <typename T, int ALG>
void dispatch_cn_gather(const Code* c, T* c2v, const T* c2v_in, const T* src, const u64* live, const int32_t* srcmap, const Geometry& g, hipStream_t st) {
    if (c->min_dc == c->max_dc && c->max_dc == 6) launch_cn_gather<T, ALG, 6, 6>(c, c2v, c2v_in, src, live, srcmap, g, st);
    else if (c->max_dc > 4 && c->max_dc <= 6) launch_cn_gather<T, ALG, 6, 0>(c, c2v, c2v_in, src, live, srcmap, g, st);
    else if (c->max_dc <= 4) launch_cn_gather<T, ALG, 4, 0>(c, c2v, c2v_in, src, live, srcmap, g, st);
    else launch_cn_gather<T, ALG, 8, 0>(c, c2v, c2v_in, src, live, srcmap, g, st);
}
template <typename T, int ALG>
void dispatch_vn_gather(const Code* c, const T* c2v, const T* prior_in, T* prior_out, T* marg, const u64* live, u64* xbits, const int32_t* srcmap,
                        const Geometry& g, hipStream_t st) {
    if (c->min_dv == c->max_dv && c->max_dv == 3) launch_vn_gather<T, ALG, 3, 3>(c, c2v, prior_in, prior_out, marg, live, xbits, srcmap, g, st);
    else if (c->min_dv == c->max_dv && c->max_dv == 4) launch_vn_gather<T, ALG, 4, 4>(c, c2v, prior_in, prior_out, marg, live, xbits, srcmap, g, st);
    else if (c->max_dv <= 4) launch_vn_gather<T, ALG, 4>(c, c2v, prior_in, prior_out, marg, live, xbits, srcmap, g, st);
    else launch_vn_gather<T, ALG, 8>(c, c2v, prior_in, prior_out, marg, live, xbits, srcmap, g, st);
}

template <typename T, int ALG>
int dispatch_cn(const Code* c, T* c2v, const T* src, const u64* live, const Geometry& g, int first, hipStream_t st) {
    const bool regular = c->min_dc == c->max_dc;
    if (regular && c->max_dc == 6) {
        launch_cn<T, ALG, 6, 6>(c, c2v, src, live, g, first, st);
        return 0;
    }
    if (c->max_dc > 4 && c->max_dc <= 6) {  // check degrees up to 6 (the rho = x^5 ensembles): no lines fetched for positions that never exist
        launch_cn<T, ALG, 6, 0>(c, c2v, src, live, g, first, st);
        return 0;
    }
    switch (pick_pow2_ge(c->max_dc, 4, 64)) {
        case 4: launch_cn<T, ALG, 4, 0>(c, c2v, src, live, g, first, st); break;
        case 8: launch_cn<T, ALG, 8, 0>(c, c2v, src, live, g, first, st); break;
        case 16: launch_cn<T, ALG, 16, 0>(c, c2v, src, live, g, first, st); break;
        case 32: launch_cn<T, ALG, 32, 0>(c, c2v, src, live, g, first, st); break;
        default: launch_cn<T, ALG, 64, 0>(c, c2v, src, live, g, first, st); break;
    }
    return 0;
}

template <typename T, int ALG>
int dispatch_vn(const Code* c, const T* c2v, const T* prior, T* marg, const u64* live, u64* xbits, const Geometry& g, hipStream_t st) {
    if (c->min_dv == c->max_dv && c->max_dv == 3) {  // (3, r)-regular codes
        launch_vn<T, ALG, 3, 3>(c, c2v, prior, marg, live, xbits, g, st);
        return 0;
    }
    if (c->min_dv == c->max_dv && c->max_dv == 4) {
        launch_vn<T, ALG, 4, 4>(c, c2v, prior, marg, live, xbits, g, st);
        return 0;
    }
    switch (pick_pow2_ge(c->max_dv, 4, 64)) {
        case 4: launch_vn<T, ALG, 4>(c, c2v, prior, marg, live, xbits, g, st); break;
        case 8: launch_vn<T, ALG, 8>(c, c2v, prior, marg, live, xbits, g, st); break;
        case 16: launch_vn<T, ALG, 16>(c, c2v, prior, marg, live, xbits, g, st); break;
        case 32: launch_vn<T, ALG, 32>(c, c2v, prior, marg, live, xbits, g, st); break;
        default: launch_vn<T, ALG, 64>(c, c2v, prior, marg, live, xbits, g, st); break;
    }
    return 0;
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// One poll of the live counters, in flight: the syndrome kernels of a check point add the live tiles / frames into `slot` of the
// device ring, a copy brings them to the pinned ring, an event marks it.  The host reads poll k only after it has enqueued the work
// up to poll k + 1, so the GPU never waits for the host inside the sweep loop.
struct PendingPoll {
    int slot, it, sweeps, tiles;
    bool tiling_current;  // false once a repack has been enqueued after this poll: its tile count no longer describes the state
};
constexpr int POLL_RING = 4;

template <typename T, int ALG>
int run(Decoder* d, const void* priors_v, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags_in, uint8_t* xhat,
        int32_t* iters, void* soft_out, hipStream_t st, const SimSource* sim = nullptr) {
    const Code* c = d->code;
    const int n = c->n, m = c->m;
    const int64_t E = c->E;
    const int tiles = (int)((B + 63) / 64);
    if (c->max_dc > 64 || c->max_dv > 64) {
        set_error("streaming backend supports node degrees up to 64 (max_dc=%d, max_dv=%d)", c->max_dc, c->max_dv);
        return LDPC_E_UNSUPPORTED;
    }
    const bool early = !(flags_in & FLAG_NO_EARLY_EXIT);
    // Frame repack (see k_repack): LLR decoders without soft output.  Policy: at a poll, when the live frames would fill less than
    // `fill` of the tiles that still hold one, gather them into dense tiles -- a repack moves (E + 2n) lines per tile once, a sweep
    // moves about (3E + 3n), so it pays as soon as about one more sweep follows.
    bool repack_ok = early && soft_out == nullptr && tiles >= 2;
    double repack_fill = 0.75;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK")) repack_ok = repack_ok && atoi(e) != 0;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK_FILL")) repack_fill = atof(e);
    if (repack_fill > 0.95) repack_fill = 0.95;

    // every buffer of the decode is reserved here, before the first sweep (no allocation -- a device synchronisation -- in the loop)
    LDPC_TRY(d->msg.reserve((size_t)tiles * E * 64 * sizeof(T)));
    LDPC_TRY(d->marg.reserve((size_t)tiles * n * 64 * sizeof(T)));
    LDPC_TRY(d->prior.reserve((size_t)tiles * n * 64 * sizeof(T)));
    LDPC_TRY(d->xbits.reserve(plane_words(tiles, n) * 8));
    LDPC_TRY(d->live.reserve((size_t)tiles * 8));
    LDPC_TRY(d->flags.reserve((size_t)tiles * 16 + 64 + POLL_RING * 16));
    if (repack_ok) {
        // second state set: the first repack fires at <= fill * 64 live frames per tile, later ones only shrink
        const size_t nt = (size_t)(repack_fill * tiles) + 2;
        if (d->msg2.reserve(nt * E * 64 * sizeof(T)) || d->marg2.reserve(nt * n * 64 * sizeof(T)) ||
            d->prior2.reserve(nt * n * 64 * sizeof(T)) || d->xbits2.reserve(plane_words((int)nt, n) * 8) || d->live2.reserve(nt * 8) ||
            d->fmap2.reserve(nt * 64 * sizeof(int32_t)) || d->fmap.reserve(nt * 64 * sizeof(int32_t)) ||
            d->rbase.reserve(((size_t)tiles + 1) * sizeof(int32_t)) || d->rmap.reserve(nt * 64 * sizeof(int32_t)))
            repack_ok = false;  // no room for a second set: decode without repacking
    }
    // the repack folded into the sweep behind it (k_repack_map + GATHER passes) where those passes are built; LDPC_STREAM_REPACK_FOLD=0: the
    // separate copy kernel (k_repack), kept for the degrees beyond and as the A/B reference
    const bool fold_repack = gather_passes_built(c) && env_int("LDPC_STREAM_REPACK_FOLD", 1) != 0;
    T* msg = (T*)d->msg.p;
    T* marg = (T*)d->marg.p;
    T* prior = (T*)d->prior.p;
    u64* xbits = (u64*)d->xbits.p;
    u64* live = (u64*)d->live.p;
    u64* tflags = (u64*)d->flags.p;                                        // [tiles][2]
    int* poll_dev = (int*)((char*)d->flags.p + (size_t)tiles * 16 + 64);  // [POLL_RING][4] ints: live tiles, live frames
    volatile int* poll_host = (volatile int*)d->pinned;                    // [POLL_RING][4] page-locked
    hipEvent_t poll_ev[POLL_RING];
    for (int i = 0; i < POLL_RING; ++i) LDPC_TRY(prof_event(d, i, &poll_ev[i]));
    size_t ev_next = POLL_RING;

    Geometry g;
    g.tiles = tiles;
    g.freeze = soft_out != nullptr ? 1 : 0;
    // Nodes per wave.  The marginal lines a check pass gathers are re-used dv times; the fewer tiles are in flight at once, the
    // more of those re-reads hit on chip -- so a tile is cut into MANY short wave tasks (tile-major task order).  Measured on one
    // MI355X (sweep of 32 768 frames of the (3,6) n = 64 800 shape, profiles/r03_stream_chunking.txt): 64 checks per wave 20.9 ms,
    // 16: 19.9, 8: 19.4, 2-4 with 16 variables per wave: 19.1-19.3 ms; n = 1200 and the n = 10 000 ensemble agree.  Longer runs only
    // beyond 2^22 tasks per launch.
    auto per_wave = [&](int nodes, int base) {
        long v = base < 1 ? 1 : base;
        while ((long)((nodes + v - 1) / v) * tiles > (1L << 22) && v < 256) v *= 2;
        return (int)(v > nodes ? ((nodes + 3) / 4 * 4) : v);
    };
    g.cpw = per_wave(m, 4);
    g.cn_chunks = (m + g.cpw - 1) / g.cpw;
    g.vpw = per_wave(n, 16);
    g.vn_chunks = (n + g.vpw - 1) / g.vpw;
    // XCD-aware task order (task_of) where a tile's marginal rows fit one XCD's 4 MB L2: measured (profiles/r03_stream_xcd.txt) the check
    // pass then fetches E + n lines per tile -- its compulsory minimum -- instead of ~2E (n = 1200: 1.78 -> 1.24 GB per launch of 65 536
    // frames, n = 10 000: 3.63 -> 2.68 GB), at unchanged time (the re-reads were being served by the Infinity Cache); at n = 64 800
    // (16.6 MB of marginals per tile) nothing is re-used either way and the plain order is 2.5 % faster
    g.xcd_aware = (size_t)n * 64 * sizeof(T) <= ((size_t)4 << 20) ? 1 : 0;

    hipEvent_t e_begin = nullptr, e_end = nullptr;
    if (d->profile) {
        LDPC_TRY(prof_event(d, ev_next++, &e_begin));
        LDPC_TRY(prof_event(d, ev_next++, &e_end));
        LDPC_HIP_TRY(hipEventRecord(e_begin, st));
    }
    LDPC_HIP_TRY(hipMemsetAsync(xbits, 0, plane_words(tiles, n) * 8, st));
    LDPC_HIP_TRY(hipMemsetAsync(tflags, 0, (size_t)tiles * 16 + 64 + POLL_RING * 16, st));
    LDPC_HIP_TRY(hipMemsetAsync(iters, 0, (size_t)B * sizeof(int32_t), st));
    if (soft_out) LDPC_HIP_TRY(hipMemsetAsync(marg, 0, (size_t)tiles * n * 64 * sizeof(T), st));  // frames that never sweep report 0
    if (sim) {  // device Monte-Carlo over BI-AWGN: the noise goes straight into the tile layout
        const int bpf = (n + 3) / 4, bpw = 16;
        hipLaunchKernelGGL((k_biawgn_tile<T>), dim3(((bpf + bpw - 1) / bpw + 3) / 4, tiles), dim3(64, 4), 0, st, *sim, B, n, bpf, bpw, prior);
    }
    if (!sim)
        hipLaunchKernelGGL((k_load_tile<T>), dim3((n + 63) / 64, tiles), dim3(256), 0, st, (const T*)priors_v, y0, B, n, prior, xbits);
    hipLaunchKernelGGL(k_init_live, dim3((tiles + 255) / 256), dim3(256), 0, st, live, B, tiles);

    const int cap = max_iter > 0 ? max_iter : 100000;  // max_iter <= 0 == unlimited upstream (src/bpa.py:28); bounded here
    // how often the live counters are polled: about every 300 us of streaming work
    const double iter_us = 15.0 + (double)tiles * 64.0 * sizeof(T) * (4.0 * E + n) / 5.0e6;
    int poll_every = (int)(300.0 / iter_us);
    poll_every = poll_every < 1 ? 1 : (poll_every > 16 ? 16 : poll_every);
    DevBuf* set_msg[2] = {&d->msg, &d->msg2};
    DevBuf* set_marg[2] = {&d->marg, &d->marg2};
    DevBuf* set_prior[2] = {&d->prior, &d->prior2};
    DevBuf* set_xbits[2] = {&d->xbits, &d->xbits2};
    DevBuf* set_live[2] = {&d->live, &d->live2};
    DevBuf* set_fmap[2] = {&d->fmap, &d->fmap2};
    int cur = 0;                 // which buffer set holds the state
    int32_t* fmap = nullptr;     // frame index of (tile, lane); null = identity (never repacked)
    int cur_tiles = tiles;
    int repacks = 0;
    int sweeps = 0;
    int polls = 0;
    const T *gather_msg = nullptr, *gather_marg = nullptr, *gather_prior = nullptr;  // non-null: the next sweep carries a folded repack out of that set
    std::vector<PendingPoll> pending;
    std::vector<ProfSpan> spans;
    bool all_left = false;
    for (int it = 0; it < cap && !all_left; ++it) {
        const bool check = early && (it > 0 || y0 != nullptr);
        if (check) {
            const bool poll = (it % poll_every) == 0 || max_iter <= 0;
            int* slot_dev = nullptr;
            int slot = 0;
            if (poll) {
                slot = polls % POLL_RING;
                slot_dev = poll_dev + 4 * slot;
                LDPC_HIP_TRY(hipMemsetAsync(slot_dev, 0, 2 * sizeof(int), st));
            }
            {
                // groups of eight tiles x chunks of the checks: enough blocks to fill the chip (about 4 per CU), at least 1024 checks each
                const int groups = (cur_tiles + 7) / 8;
                int sblocks = (1024 + groups - 1) / groups;
                const int smax = (m + 1023) / 1024;
                sblocks = sblocks < 1 ? 1 : (sblocks > smax ? smax : sblocks);
                const int cpb = (m + sblocks - 1) / sblocks;
                hipLaunchKernelGGL(k_syndrome_part, dim3(sblocks, groups), dim3(256), 0, st, c->d_row_ptr, c->d_edge_var, xbits, live, tflags, m, n,
                                   cur_tiles, cpb);
                hipLaunchKernelGGL(k_syndrome_fin, dim3(cur_tiles), dim3(64), 0, st, tflags, live, iters, slot_dev, B, sweeps, fmap);
            }
            if (poll) {
                LDPC_HIP_TRY(hipMemcpyAsync((void*)(poll_host + 4 * slot), slot_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
                LDPC_HIP_TRY(hipEventRecord(poll_ev[slot], st));
                pending.push_back({slot, it, sweeps, cur_tiles, true});
                ++polls;
                // Read the OLDEST poll once a newer one is enqueued behind it (its event has long fired: a block of sweeps lies in
                // between); with a single-tile batch, or an unbounded run, there is nothing to overlap and the newest is read at once.
                // LONG sweeps (more than ~1.5 ms of streaming work: n = 64 800 x 512 tiles is 19 ms) also read the newest at once: the
                // host round trip is a percent of one sweep, and the repack decision then rests on the live counts of THIS check point
                // instead of those of a sweep ago -- while frames leave by the thousand per sweep (profiles/r05_config5_timeline.txt).
                const double sweep_us_now = (double)cur_tiles * 64.0 * sizeof(T) * (4.0 * E + n) / 5.0e6;
                const bool eager = max_iter <= 0 || cur_tiles < 2 || sweep_us_now > 1500.0;
                while (!pending.empty() && (pending.size() >= 2 || eager)) {
                    const PendingPoll pp = pending.front();
                    pending.erase(pending.begin());
                    LDPC_HIP_TRY(hipEventSynchronize(poll_ev[pp.slot]));
                    const int lt = poll_host[4 * pp.slot], lf = poll_host[4 * pp.slot + 1];
                    if (lt == 0) {  // every frame had left at that check point; what was enqueued since found no live tile
                        all_left = true;
                        sweeps = pp.sweeps;
                        break;
                    }
                    // When to repack: the live frames fill at most `repack_fill` (0.75) of the live tiles.  (A "rent or buy" schedule -- repack once
                    // the tile-sweeps spent on departed lanes reach the cost of a repack -- was measured within the run-to-run spread of this
                    // rule in round 5 and removed in round 6: HISTORY.md.)
                    const bool want_repack = (double)lf <= repack_fill * 64.0 * lt;
                    if (repack_ok && pp.tiling_current && pp.it > 0 && lt >= 2 && want_repack && it + 1 < cap) {
                        const int nt = (lf + 63) / 64;
                        const int nx = 1 - cur;
                        // the decisions of every frame of the old tiles (those that left keep them; the moved ones overwrite theirs later)
                        launch_unpack(d, xbits, xhat, B, n, cur_tiles, fmap, st);
                        hipLaunchKernelGGL(k_repack_plan, dim3(1), dim3(1024), 0, st, live, cur_tiles, (int32_t*)d->rbase.p);
                        if (fold_repack) {
                            // only the map now; the sweep enqueued below moves the state (GATHER passes: old set in, new set out)
                            hipLaunchKernelGGL(k_repack_map, dim3(nt), dim3(64), 0, st, live, (u64*)set_live[nx]->p, (const int32_t*)d->rbase.p, fmap,
                                               (int32_t*)set_fmap[nx]->p, (int32_t*)d->rmap.p, cur_tiles);
                            gather_msg = msg;
                            gather_marg = marg;
                            gather_prior = prior;
                        } else {
                            const int rows_per_wave = 128;
                            const int chunks = (int)((E + n + rows_per_wave - 1) / rows_per_wave);
                            hipLaunchKernelGGL((k_repack<T>), dim3((chunks + 3) / 4, nt), dim3(64, 4), 0, st, msg, (T*)set_msg[nx]->p, marg,
                                               (T*)set_marg[nx]->p, prior, (T*)set_prior[nx]->p, xbits, (u64*)set_xbits[nx]->p, live,
                                               (u64*)set_live[nx]->p, (const int32_t*)d->rbase.p, fmap, (int32_t*)set_fmap[nx]->p, cur_tiles, n, E,
                                               rows_per_wave);
                        }
                        cur = nx;
                        msg = (T*)set_msg[cur]->p;
                        marg = (T*)set_marg[cur]->p;
                        prior = (T*)set_prior[cur]->p;
                        xbits = (u64*)set_xbits[cur]->p;
                        live = (u64*)set_live[cur]->p;
                        fmap = (int32_t*)set_fmap[cur]->p;
                        cur_tiles = nt;
                        g.tiles = nt;
                        ++repacks;
                        for (PendingPoll& q : pending) q.tiling_current = false;
                    }
                }
                if (all_left) break;
            }
        }
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (d->profile) {
            LDPC_TRY(prof_event(d, ev_next++, &e0));
            LDPC_TRY(prof_event(d, ev_next++, &e1));
            LDPC_TRY(prof_event(d, ev_next++, &e2));
            LDPC_HIP_TRY(hipEventRecord(e0, st));
        }
        if (gather_msg) {  // the sweep that carries a repack: state read from the old set through the map, written densely into the new one
            dispatch_cn_gather<T, ALG>(c, msg, gather_msg, gather_marg, live, (const int32_t*)d->rmap.p, g, st);
            if (d->profile) LDPC_HIP_TRY(hipEventRecord(e1, st));
            dispatch_vn_gather<T, ALG>(c, msg, gather_prior, prior, marg, live, xbits, (const int32_t*)d->rmap.p, g, st);
            gather_msg = gather_marg = gather_prior = nullptr;
        } else {
            dispatch_cn<T, ALG>(c, msg, it == 0 ? prior : marg, live, g, it == 0 ? 1 : 0, st);
            if (d->profile) LDPC_HIP_TRY(hipEventRecord(e1, st));
            dispatch_vn<T, ALG>(c, msg, prior, marg, live, xbits, g, st);
        }
        if (d->profile) {
            LDPC_HIP_TRY(hipEventRecord(e2, st));
            spans.push_back({0, e0, e1});
            spans.push_back({1, e1, e2});
        }
        ++sweeps;
    }
    hipLaunchKernelGGL(k_finish_iters, dim3(cur_tiles), dim3(64), 0, st, live, iters, B, sweeps, fmap);
    launch_unpack(d, xbits, xhat, B, n, cur_tiles, fmap, st);
    if (soft_out)
        hipLaunchKernelGGL(k_soft_out<T>, dim3((n + 3) / 4, tiles), dim3(256), 0, st, marg, (T*)soft_out, B, n);
    LDPC_HIP_TRY(hipGetLastError());
    // polls still in flight copy into the pinned ring; the next decode of this handle may run on another stream and reuse the slots:
    // let the last copy land first (everything of this decode is enqueued by now, the GPU is not waiting for the host)
    if (!pending.empty()) LDPC_HIP_TRY(hipEventSynchronize(poll_ev[pending.back().slot]));
    if (d->profile) {
        LDPC_HIP_TRY(hipEventRecord(e_end, st));
        LDPC_HIP_TRY(hipStreamSynchronize(st));
        LDPC_TRY(prof_collect(d, spans));
        float total = 0.f;
        LDPC_HIP_TRY(hipEventElapsedTime(&total, e_begin, e_end));
        d->prof_ms[3] += total;  // everything the decode enqueued: the two passes + load, syndrome, repack, unpack kernels
        d->prof_launches[3] += 1;
    }
    d->last_repacks = repacks;
    d->last_sweeps = sweeps;
    d->last_backend = BK_STREAM;
    return LDPC_OK;
}

}  // namespace

// =====================================================================================================================================
// fp16 STORAGE mode (LDPC_DTYPE_F16): the check -> variable messages -- the E-sized part of the sweep's traffic -- are kept as fp16, all
// arithmetic and the marginals stay fp32.  A lane must still move at least 4 bytes per access to use the memory system (2-byte lanes
// would halve the bytes AND the rate), so this mode works on PAIR-TILES of 128 frames: lane l holds frames l and 64 + l of the pair
// (= the 64-frame tiles 2P and 2P + 1 of the planes / live words, which keep their layout), a message line is 64 x half2 = 256 B, a
// prior line 64 x float2 = 512 B.  The sweep is the two-array form SURVEY 8(d) prices (v2c written by the variable pass and read back by the
// check pass): with 2-byte messages every E-sized line is moved exactly once per pass and nothing is re-read -- the marginal-resident form
// of the fp32 passes would gather 4-byte marginal lines dv times each, which at half-size messages becomes half of the traffic.  Bytes
// per frame-sweep: check pass 2E + 2E, variable pass 2E + 2E + 4n: 8E + 4n = 1.81 MB at n = 64 800 against 12E + 12n = 3.11 MB in fp32
// (SURVEY 8(d)'s all-fp16 figure 2(4E + n) differs only by the fp32 priors).
// A throughput mode, NOT the parity mode: messages are rounded to 11 significant bits each sweep (and saturate at +-65504 where the fp32
// mode would carry larger finite values); tests hold it to a stated per-sweep tolerance against the fp32 kernels and to the published
// curves.  Frame repack on pair-tiles (k_repack16); with a soft output there is none, and the soft output is that of the last sweep of the
// BATCH (frames that have left keep evolving).
namespace {

// the load and the conversion are separate so that a kernel can issue all its line loads before it touches the first result
__device__ __forceinline__ uint32_t msg16_ld_raw(const __half2* p) { return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)); }
__device__ __forceinline__ float2 msg16_cvt(uint32_t raw) { return __half22float2(*reinterpret_cast<const __half2*>(&raw)); }
__device__ __forceinline__ float2 msg16_ld(const __half2* p) { return msg16_cvt(msg16_ld_raw(p)); }
template <int ALG>
__device__ __forceinline__ float sat16(float v) {
    // finite values beyond the fp16 range saturate (min-sum messages of a trapped frame grow without bound); sum-product keeps its +-inf
    // and NaN (src/bpa.py:38 relies on them)
    if constexpr (ALG == ALG_SPA) return (__builtin_fabsf(v) <= 65504.0f || !(__builtin_fabsf(v) < __builtin_huge_valf())) ? v : __builtin_copysignf(65504.0f, v);
    else return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
}
template <int ALG>
__device__ __forceinline__ void msg16_st(__half2* p, float x, float y) {
    const __half2 h = __floats2half2_rn(sat16<ALG>(x), sat16<ALG>(y));
    __builtin_nontemporal_store(*reinterpret_cast<const uint32_t*>(&h), reinterpret_cast<uint32_t*>(p));
}

// priors [B,n] fp32 -> pair-tile layout [P][n][64] float2; optional hard word y0 -> decision planes of the two 64-frame tiles
__global__ __launch_bounds__(256) void k_load_tile16(const float* __restrict__ priors, const uint8_t* __restrict__ y0, int64_t B, int n,
                                                     float2* __restrict__ prior_t, u64* __restrict__ xbits) {
    __shared__ float sp[64][65];
    __shared__ uint8_t sy[64][68];
    const int P = blockIdx.y, v0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int sub = 0; sub < 2; ++sub) {
        const int tile = 2 * P + sub;
        __syncthreads();
        for (int f = ty; f < 64; f += 4) {
            const int64_t fr = (int64_t)tile * 64 + f;
            const int v = v0 + tx;
            float val = 0.0f;
            uint8_t yy = 0;
            if (fr < B && v < n) {
                if (y0) yy = y0[fr * n + v];
                val = priors[fr * n + v];
            }
            sp[f][tx] = val;
            sy[f][tx] = yy;
        }
        __syncthreads();
        for (int vv = ty; vv < 64; vv += 4) {
            const int v = v0 + vv;
            if (v < n) {
                reinterpret_cast<float*>(prior_t + ((int64_t)P * n + v) * 64 + tx)[sub] = sp[tx][vv];
                if (y0) {
                    const u64 one = __ballot(sy[tx][vv] != 0);
                    if (tx == 0) xbits[plane_at(tile, v, n)] = one;
                }
            }
        }
    }
}

// BI-AWGN channel + LLR straight into the pair-tile layout: the expressions of k_biawgn<float> (bit-identical priors)
__global__ __launch_bounds__(256) void k_biawgn_tile16(SimSource s, int64_t B, int n, int bpf, int blocks_per_wave, float2* __restrict__ prior_t) {
    const int lane = threadIdx.x, P = blockIdx.y;
    const int64_t fa = (int64_t)P * 128 + lane, fb = fa + 64;
    if (fa >= B) return;
    const int j0 = (blockIdx.x * 4 + threadIdx.y) * blocks_per_wave;
    const float sg = (float)s.sigma, k = (float)s.inv_var2, mean = (float)(2 * s.codeword - 1);
    float2* pt = prior_t + (int64_t)P * n * 64 + lane;
    for (int j = j0; j < min(bpf, j0 + blocks_per_wave); ++j) {
        const Philox4 pa = philox_word_block(s.seed, s.stream, s.frame0 + (uint64_t)fa, (uint32_t)j);
        const Philox4 pb = philox_word_block(s.seed, s.stream, s.frame0 + (uint64_t)fb, (uint32_t)j);  // (beyond the batch: never read back)
        float za[4], zb[4];
        box_muller<float>(pa.w[0], pa.w[1], za[0], za[1]);
        box_muller<float>(pa.w[2], pa.w[3], za[2], za[3]);
        box_muller<float>(pb.w[0], pb.w[1], zb[0], zb[1]);
        box_muller<float>(pb.w[2], pb.w[3], zb[2], zb[3]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * j + q < n) pt[(int64_t)(4 * j + q) * 64] = make_float2(-(k * (mean + sg * za[q])), -(k * (mean + sg * zb[q])));
    }
}

// Check pass: per check, GATHER the variable -> check lines of its edges (each line exactly once per sweep: no re-read, unlike the
// marginal-resident fp32 passes), apply the rule, STREAM the check -> variable lines out.  First sweep: v2c = prior (src/bpa.py:19).
// FIRST is a template parameter: as a run-time flag the compiler kept a branch per line and waited for every gathered line before issuing the next
template <int ALG, int DCMAX, int FIXED_DC, int UNR, bool FIRST>
__global__ __launch_bounds__(256) void k_cn16(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_var, const int32_t* __restrict__ edge_vpos,
                                              __half2* __restrict__ c2v, const __half2* __restrict__ v2c, const float2* __restrict__ prior_t,
                                              const u64* __restrict__ live, int m, int n, int64_t E, int pairs, int tiles, int chunks, int cpw) {
    const int lane = threadIdx.x;
    int P, chunk;
    if (!task_of(pairs, chunks, 0, &P, &chunk)) return;
    if ((live[2 * P] | (2 * P + 1 < tiles ? live[2 * P + 1] : 0ull)) == 0) return;
    __half2* ct = c2v + (int64_t)P * E * 64 + lane;
    const __half2* vt = v2c + (int64_t)P * E * 64 + lane;
    const float2* pt = prior_t + (int64_t)P * n * 64 + lane;
    const int c_end = min(m, (chunk + 1) * cpw);
    for (int c = chunk * cpw; c < c_end; c += UNR) {
        float vx[UNR][DCMAX], vy[UNR][DCMAX];
        int k0[UNR], deg[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int cc = c + u;
            if (cc < c_end) {
                k0[u] = FIXED_DC > 0 ? cc * FIXED_DC : row_ptr[cc];
                deg[u] = FIXED_DC > 0 ? FIXED_DC : row_ptr[cc + 1] - k0[u];
            } else {
                k0[u] = 0;
                deg[u] = 0;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int j = 0; j < DCMAX; ++j) {
                // a short row re-reads its last edge (an empty row edge 0): no branch per line
                const int kk = FIXED_DC > 0 ? k0[u] + j : (deg[u] > 0 ? k0[u] + (j < deg[u] ? j : deg[u] - 1) : 0);
                float2 v;
                if constexpr (FIRST) v = pt[(int64_t)edge_var[kk] * 64];
                else v = msg16_ld(vt + (int64_t)edge_vpos[kk] * 64);
                vx[u][j] = v.x;
                vy[u][j] = v.y;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            cn_rule<float, ALG, DCMAX>(vx[u], deg[u]);
            cn_rule<float, ALG, DCMAX>(vy[u], deg[u]);
#pragma unroll
            for (int j = 0; j < DCMAX; ++j)
                if (j < deg[u]) msg16_st<ALG>(ct + (int64_t)(k0[u] + j) * 64, vx[u][j], vy[u][j]);
        }
    }
}

// Variable pass: per variable, gather its check -> variable lines (each read once), marginal = prior + ordered sum (src/bpa.py:35), decision
// bits, and v2c_j = marginal - c2v_j (src/bpa.py:37) STREAMED out in variable-major order (line p0 + j).  The marginal itself is written
// only when a soft output was asked for.
template <int ALG, int DVMAX, int UNR, int FIXED_DV>
__global__ __launch_bounds__(256) void k_vn16(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ col_edge, const __half2* __restrict__ c2v,
                                              __half2* __restrict__ v2c, const float2* __restrict__ prior_t, float2* __restrict__ marg_t,
                                              const u64* __restrict__ live, u64* __restrict__ xbits, int n, int64_t E, int pairs, int tiles, int chunks, int vpw) {
    const int lane = threadIdx.x;
    int P, chunk;
    if (!task_of(pairs, chunks, 0, &P, &chunk)) return;
    const bool has_b = 2 * P + 1 < tiles;
    const u64 lva = live[2 * P], lvb = has_b ? live[2 * P + 1] : 0ull;
    if ((lva | lvb) == 0) return;
    const __half2* ct = c2v + (int64_t)P * E * 64 + lane;
    __half2* vt = v2c + (int64_t)P * E * 64 + lane;
    const float2* pt = prior_t + (int64_t)P * n * 64 + lane;
    float2* mt = marg_t ? marg_t + (int64_t)P * n * 64 + lane : nullptr;
    u64* xa = xbits + plane_at(2 * P, 0, n);
    u64* xb = xbits + plane_at(2 * P + 1, 0, n);  // (only touched when the pair has a second tile)
    const int v_end = min(n, (chunk + 1) * vpw);
    for (int vbase = chunk * vpw; vbase < v_end; vbase += UNR) {
        float2 c[UNR][DVMAX], pr[UNR];
        uint32_t raw[UNR][DVMAX];
        u64 olda[UNR], oldb[UNR];
        int deg[UNR], p0[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int vv = vbase + u < v_end ? vbase + u : v_end - 1;  // past the end: the last variable once more, result unused
            p0[u] = FIXED_DV > 0 ? vv * FIXED_DV : col_ptr[vv];
            deg[u] = vbase + u < v_end ? (FIXED_DV > 0 ? FIXED_DV : col_ptr[vv + 1] - p0[u]) : -1;
            pr[u] = pt[(int64_t)vv * 64];
            // decisions of frames that have left, read with the lines (a read behind the stores of this variable would wait for them)
            if (lva != ~0ull) olda[u] = uniform_ld64(xa + 8 * vv);
            if (has_b && lvb != ~0ull) oldb[u] = uniform_ld64(xb + 8 * vv);
            // lines of an irregular variable sit behind a wave-uniform branch each; nothing in the branch but the load (no default value, no
            // conversion), so that no load waits for the one before it
#pragma unroll
            for (int j = 0; j < DVMAX; ++j) {
                if (FIXED_DV > 0 || j < deg[u]) raw[u][j] = msg16_ld_raw(ct + (int64_t)col_edge[p0[u] + j] * 64);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int j = 0; j < DVMAX; ++j) c[u][j] = (FIXED_DV > 0 || j < deg[u]) ? msg16_cvt(raw[u][j]) : make_float2(0.0f, 0.0f);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (deg[u] < 0) continue;  // wave-uniform
            float sx = 0.0f, sy = 0.0f;  // ordered sum from +0.0, prior last (src/bpa.py:35, src/math_utils.py:7)
#pragma unroll
            for (int j = 0; j < DVMAX; ++j)
                if (FIXED_DV > 0 || j < deg[u]) {
                    sx += c[u][j].x;
                    sy += c[u][j].y;
                }
            const float2 marg = make_float2(pr[u].x + sx, pr[u].y + sy);
            if (mt) {  // soft output (diagnostic path): a frame that has left keeps the marginal of ITS last sweep, as the fp32 / fp64 kernels do
                float2 w = marg;
                if (lva != ~0ull || lvb != ~0ull) {
                    const float2 old = mt[(int64_t)(vbase + u) * 64];
                    if (!((lva >> lane) & 1ull)) w.x = old.x;
                    if (!((lvb >> lane) & 1ull)) w.y = old.y;
                }
                mt[(int64_t)(vbase + u) * 64] = w;
            }
#pragma unroll
            for (int j = 0; j < DVMAX; ++j)
                if (FIXED_DV > 0 || j < deg[u]) msg16_st<ALG>(vt + (int64_t)(p0[u] + j) * 64, marg.x - c[u][j].x, marg.y - c[u][j].y);
            const u64 onea = __ballot(marg.x < 0.0f), oneb = __ballot(marg.y < 0.0f);  // NaN marginal -> 0 (src/bpa.py:38,62)
            const int vv = vbase + u;
            u64 ma = onea, mb = oneb;
            if (lva != ~0ull) ma = (olda[u] & ~lva) | (onea & lva);  // frames that have left keep their decisions
            if (has_b && lvb != ~0ull) mb = (oldb[u] & ~lvb) | (oneb & lvb);
            if (lane == 0) {
                xa[8 * vv] = ma;
                if (has_b) xb[8 * vv] = mb;
            }
        }
    }
}

// marginal pair-tile [n][64] float2 -> [B,n] (diagnostic / tolerance tests; the MESSAGES of a frame that has left keep evolving in this
// mode, its stored marginal and decisions do not: k_vn16 masks both by the live words)
__global__ void k_soft_out16(const float2* __restrict__ soft_t, float* __restrict__ out, int64_t B, int n) {
    const int P = blockIdx.y, lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n) return;
    const float2 m = soft_t[((int64_t)P * n + v) * 64 + lane];
    const int64_t fa = (int64_t)P * 128 + lane, fb = fa + 64;
    if (fa < B) out[fa * n + v] = m.x;
    if (fb < B) out[fb * n + v] = m.y;
}

// Frame repack of the fp16 storage mode (the policy and the ranking of ldpc_repack.hpp, as k_repack above).  The two halves of a pair-tile are
// two 64-frame tiles (2P: the x components, 2P + 1: the y components), so a destination lane draws its two frames from two unrelated
// (tile, lane) sources: component (tile & 1) of lane `lane` of pair tile >> 1.  What has to move after a variable pass: the variable -> check
// lines, the priors, the decision planes, the live words and the frame map; the check -> variable lines are rewritten by the next check pass.
__global__ __launch_bounds__(256) void k_repack16(const __half2* __restrict__ v2c_src, __half2* __restrict__ v2c_dst, const float2* __restrict__ prior_src,
                                                  float2* __restrict__ prior_dst, const u64* __restrict__ xb_src, u64* __restrict__ xb_dst,
                                                  const u64* __restrict__ live_src, u64* __restrict__ live_dst, const int32_t* __restrict__ base,
                                                  const int32_t* __restrict__ frame_src, int32_t* __restrict__ frame_dst, int tiles_src, int n, int64_t E,
                                                  int rows_per_wave) {
    const int lane = threadIdx.x;
    const int dp = blockIdx.y;  // destination pair
    int sa, la, sb, lb;
    const bool ha = repack_source(base, live_src, tiles_src, (2 * dp) * 64 + lane, &sa, &la);
    const bool hb = repack_source(base, live_src, tiles_src, (2 * dp + 1) * 64 + lane, &sb, &lb);
    const int chunk = blockIdx.x * 4 + threadIdx.y;
    const int64_t rows = E + n;
    const int64_t r0 = (int64_t)chunk * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    // a message line is 64 half2 = 128 halves, a prior line 64 float2 = 128 floats
    const __half* va = reinterpret_cast<const __half*>(v2c_src + ((int64_t)(sa >> 1) * E * 64 + la)) + (sa & 1);
    const __half* vb = reinterpret_cast<const __half*>(v2c_src + ((int64_t)(sb >> 1) * E * 64 + lb)) + (sb & 1);
    const float* pa = reinterpret_cast<const float*>(prior_src + ((int64_t)(sa >> 1) * n * 64 + la)) + (sa & 1);
    const float* pb = reinterpret_cast<const float*>(prior_src + ((int64_t)(sb >> 1) * n * 64 + lb)) + (sb & 1);
    __half2* vd = v2c_dst + (int64_t)dp * E * 64 + lane;
    float2* pd = prior_dst + (int64_t)dp * n * 64 + lane;
    const __half hz = __float2half(0.0f);
    for (int64_t r = r0; r < r1; ++r) {
        if (r < E) {
            vd[r * 64] = __halves2half2(ha ? va[r * 128] : hz, hb ? vb[r * 128] : hz);
        } else {
            const int64_t v = r - E;
            pd[v * 64] = make_float2(ha ? pa[v * 128] : 0.0f, hb ? pb[v * 128] : 0.0f);
            const u64 wa = ha ? xb_src[plane_at(sa, v, n)] : 0ull, wb = hb ? xb_src[plane_at(sb, v, n)] : 0ull;
            const u64 plane_a = __ballot(ha && ((wa >> la) & 1ull)), plane_b = __ballot(hb && ((wb >> lb) & 1ull));
            if (lane == 0) {
                xb_dst[plane_at(2 * dp, v, n)] = plane_a;
                xb_dst[plane_at(2 * dp + 1, v, n)] = plane_b;
            }
        }
    }
    if (chunk == 0) {
        frame_dst[(int64_t)(2 * dp) * 64 + lane] = ha ? (frame_src ? frame_src[(int64_t)sa * 64 + la] : sa * 64 + la) : -1;
        frame_dst[(int64_t)(2 * dp + 1) * 64 + lane] = hb ? (frame_src ? frame_src[(int64_t)sb * 64 + lb] : sb * 64 + lb) : -1;
        const u64 lva = __ballot(ha), lvb = __ballot(hb);
        if (lane == 0) {
            live_dst[2 * dp] = lva;
            live_dst[2 * dp + 1] = lvb;
        }
    }
}

template <int ALG>
int run16(Decoder* d, const float* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags_in, uint8_t* xhat, int32_t* iters,
          float* soft_out, hipStream_t st, const SimSource* sim) {
    const Code* c = d->code;
    const int n = c->n, m = c->m;
    const int64_t E = c->E;
    if (c->max_dc > 64 || c->max_dv > 64) {
        set_error("streaming backend supports node degrees up to 64 (max_dc=%d, max_dv=%d)", c->max_dc, c->max_dv);
        return LDPC_E_UNSUPPORTED;
    }
    const int tiles0 = (int)((B + 63) / 64), pairs0 = (tiles0 + 1) / 2;
    int tiles = tiles0, pairs = pairs0;  // shrink when the live frames are repacked into dense pair-tiles
    const bool early = !(flags_in & FLAG_NO_EARLY_EXIT);
    LDPC_TRY(d->msg.reserve((size_t)pairs * E * 64 * sizeof(__half2)));
    LDPC_TRY(d->msg2.reserve((size_t)pairs * E * 64 * sizeof(__half2)));  // variable -> check lines, variable-major
    // Frame repack (k_repack16; policy of the fp32 passes): at a poll, when the live frames would fill less than `fill` of the tiles that
    // still hold one, they are gathered into dense pair-tiles.  Second state set: variable -> check lines, priors, planes, live words, map.
    bool repack_ok = early && soft_out == nullptr && tiles0 >= 4;
    double repack_fill = 0.75;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK")) repack_ok = repack_ok && atoi(e) != 0;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK_FILL")) repack_fill = atof(e);
    if (repack_fill > 0.95) repack_fill = 0.95;
    if (repack_ok) {
        const size_t np2 = ((size_t)(repack_fill * tiles0) + 2 + 1) / 2;  // pairs of the second set: the first repack fires at <= fill * tiles
        if (d->marg2.reserve(np2 * E * 64 * sizeof(__half2)) || d->prior2.reserve(np2 * n * 64 * sizeof(float2)) ||
            d->xbits2.reserve(plane_words((int)(2 * np2), n) * 8) || d->live2.reserve(2 * np2 * 8) || d->fmap2.reserve(2 * np2 * 64 * sizeof(int32_t)) ||
            d->fmap.reserve(2 * np2 * 64 * sizeof(int32_t)) || d->rbase.reserve(((size_t)tiles0 + 2) * sizeof(int32_t)))
            repack_ok = false;  // no room for a second set: decode without repacking
    }
    if (soft_out) LDPC_TRY(d->marg.reserve((size_t)pairs * n * 64 * sizeof(float2)));
    if (!d->scratch.p) {  // row-major edge k -> its position in the CSC edge list (where the variable pass writes its v2c line)
        std::vector<int32_t> vpos((size_t)E);
        for (int64_t p = 0; p < E; ++p) vpos[(size_t)c->col_edge[(size_t)p]] = (int32_t)p;
        LDPC_TRY(d->scratch.reserve((size_t)E * sizeof(int32_t) + 16));
        LDPC_HIP_TRY(hipMemcpy(d->scratch.p, vpos.data(), (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const int32_t* edge_vpos = (const int32_t*)d->scratch.p;
    LDPC_TRY(d->prior.reserve((size_t)pairs * n * 64 * sizeof(float2)));
    LDPC_TRY(d->xbits.reserve(plane_words(2 * pairs, n) * 8));
    LDPC_TRY(d->live.reserve((size_t)2 * pairs * 8));
    LDPC_TRY(d->flags.reserve((size_t)2 * pairs * 16 + 64));
    __half2* msg = (__half2*)d->msg.p;
    __half2* v2c = (__half2*)d->msg2.p;
    float2* marg = soft_out ? (float2*)d->marg.p : nullptr;
    float2* prior = (float2*)d->prior.p;
    u64* xbits = (u64*)d->xbits.p;
    u64* live = (u64*)d->live.p;
    u64* tflags = (u64*)d->flags.p;
    int* live_tiles = (int*)((char*)d->flags.p + (size_t)2 * pairs * 16);
    volatile int* poll_host = (volatile int*)d->pinned;
    LDPC_HIP_TRY(hipMemsetAsync(xbits, 0, plane_words(2 * pairs, n) * 8, st));
    LDPC_HIP_TRY(hipMemsetAsync(tflags, 0, (size_t)2 * pairs * 16 + 64, st));
    LDPC_HIP_TRY(hipMemsetAsync(live, 0, (size_t)2 * pairs * 8, st));
    LDPC_HIP_TRY(hipMemsetAsync(iters, 0, (size_t)B * sizeof(int32_t), st));
    if (marg) LDPC_HIP_TRY(hipMemsetAsync(marg, 0, (size_t)pairs * n * 64 * sizeof(float2), st));  // frames that never sweep report 0
    if (sim) {
        const int bpf = (n + 3) / 4, bpw = 16;
        hipLaunchKernelGGL(k_biawgn_tile16, dim3(((bpf + bpw - 1) / bpw + 3) / 4, pairs), dim3(64, 4), 0, st, *sim, B, n, bpf, bpw, prior);
    } else {
        hipLaunchKernelGGL(k_load_tile16, dim3((n + 63) / 64, pairs), dim3(256), 0, st, priors, y0, B, n, prior, xbits);
    }
    hipLaunchKernelGGL(k_init_live, dim3((tiles + 255) / 256), dim3(256), 0, st, live, B, tiles);
    DevBuf* set_v2c[2] = {&d->msg2, &d->marg2};
    DevBuf* set_prior[2] = {&d->prior, &d->prior2};
    DevBuf* set_xbits[2] = {&d->xbits, &d->xbits2};
    DevBuf* set_live[2] = {&d->live, &d->live2};
    DevBuf* set_fmap[2] = {&d->fmap, &d->fmap2};
    int32_t* fmap = nullptr;  // frame of (tile, lane); null = identity (never repacked)
    int cur = 0, repacks = 0;
    const int cpw = 4, vpw = 16;
    const int cn_chunks = (m + cpw - 1) / cpw, vn_chunks = (n + vpw - 1) / vpw;
    const int cap = max_iter > 0 ? max_iter : 100000;
    // the live counters are read (synchronously) every fourth sweep; every sweep where a sweep is more than ~1.5 ms of streaming work
    // (the round trip is then a percent of it, and frames leave by the thousand per sweep: the repack should not wait three sweeps)
    const int poll_every = (double)tiles * 64.0 * (8.0 * E + 4.0 * n) / 5.0e6 > 1500.0 ? 1 : 4;
    const bool reg36 = c->min_dc == c->max_dc && c->max_dc == 6;
    const bool dv3 = c->min_dv == c->max_dv && c->max_dv == 3;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    std::vector<ProfSpan> spans;
    size_t ev_next = 0;
    int sweeps = 0;
    bool all_left = false;
    for (int it = 0; it < cap && !all_left; ++it) {
        if (early && (it > 0 || y0 != nullptr)) {
            const bool poll = (it % poll_every) == 0;
            if (poll) LDPC_HIP_TRY(hipMemsetAsync(live_tiles, 0, 2 * sizeof(int), st));
            const int groups = (tiles + 7) / 8;
            int sblocks = (1024 + groups - 1) / groups;
            const int smax = (m + 1023) / 1024;
            sblocks = sblocks < 1 ? 1 : (sblocks > smax ? smax : sblocks);
            hipLaunchKernelGGL(k_syndrome_part, dim3(sblocks, groups), dim3(256), 0, st, c->d_row_ptr, c->d_edge_var, xbits, live, tflags, m, n, tiles,
                               (m + sblocks - 1) / sblocks);
            hipLaunchKernelGGL(k_syndrome_fin, dim3(tiles), dim3(64), 0, st, tflags, live, iters, poll ? live_tiles : nullptr, B, sweeps, (const int32_t*)fmap);
            if (poll) {
                LDPC_HIP_TRY(hipMemcpyAsync((void*)poll_host, live_tiles, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
                LDPC_HIP_TRY(hipStreamSynchronize(st));
                const int lt = poll_host[0], lf = poll_host[1];  // tiles that still hold a live frame, live frames
                if (lt == 0) {
                    all_left = true;
                    break;
                }
                const bool want_repack = (double)lf <= repack_fill * 64.0 * lt;
                if (repack_ok && it > 0 && lt >= 2 && want_repack && it + 1 < cap) {
                    const int nt = (lf + 63) / 64, np = (nt + 1) / 2, nx = 1 - cur;
                    // the decisions of every frame of the old tiles (those that left keep them; the moved ones overwrite theirs at the end)
                    launch_unpack(d, xbits, xhat, B, n, tiles, fmap, st);
                    hipLaunchKernelGGL(k_repack_plan, dim3(1), dim3(1024), 0, st, live, tiles, (int32_t*)d->rbase.p);
                    const int rows_per_wave = 128;
                    const int chunks = (int)((E + n + rows_per_wave - 1) / rows_per_wave);
                    hipLaunchKernelGGL(k_repack16, dim3((chunks + 3) / 4, np), dim3(64, 4), 0, st, v2c, (__half2*)set_v2c[nx]->p, prior, (float2*)set_prior[nx]->p,
                                       xbits, (u64*)set_xbits[nx]->p, live, (u64*)set_live[nx]->p, (const int32_t*)d->rbase.p, (const int32_t*)fmap,
                                       (int32_t*)set_fmap[nx]->p, tiles, n, E, rows_per_wave);
                    cur = nx;
                    v2c = (__half2*)set_v2c[cur]->p;
                    prior = (float2*)set_prior[cur]->p;
                    xbits = (u64*)set_xbits[cur]->p;
                    live = (u64*)set_live[cur]->p;
                    fmap = (int32_t*)set_fmap[cur]->p;
                    pairs = np;
                    tiles = 2 * np;  // (an odd tile count leaves the last pair's second tile without a live frame)
                    ++repacks;
                }
            }
        }
        if (d->profile) {
            LDPC_TRY(prof_event(d, ev_next++, &e0));
            LDPC_TRY(prof_event(d, ev_next++, &e1));
            LDPC_TRY(prof_event(d, ev_next++, &e2));
            LDPC_HIP_TRY(hipEventRecord(e0, st));
        }
        const bool first = it == 0;
        const dim3 cgrid(task_blocks(pairs, cn_chunks, 0)), vgrid(task_blocks(pairs, vn_chunks, 0)), blk(64, 4);
#define LDPC_CN16_LAUNCH(DCM, FDC, UNR, FIRST) \
    hipLaunchKernelGGL((k_cn16<ALG, DCM, FDC, UNR, FIRST>), cgrid, blk, 0, st, c->d_row_ptr, c->d_edge_var, edge_vpos, msg, v2c, prior, live, m, n, E, pairs, tiles, cn_chunks, cpw)
#define LDPC_CN16(DCM, FDC, UNR)                       \
    do {                                               \
        if (first) LDPC_CN16_LAUNCH(DCM, FDC, UNR, true); \
        else LDPC_CN16_LAUNCH(DCM, FDC, UNR, false);      \
    } while (0)
        if (reg36) LDPC_CN16(6, 6, 2);
        else if (c->max_dc <= 4) LDPC_CN16(4, 0, 2);
        else if (c->max_dc <= 6) LDPC_CN16(6, 0, 2);
        else if (c->max_dc <= 8) LDPC_CN16(8, 0, 1);
        else if (c->max_dc <= 16) LDPC_CN16(16, 0, 1);
        else if (c->max_dc <= 32) LDPC_CN16(32, 0, 1);
        else LDPC_CN16(64, 0, 1);
#undef LDPC_CN16
#undef LDPC_CN16_LAUNCH
        if (d->profile) LDPC_HIP_TRY(hipEventRecord(e1, st));
#define LDPC_VN16(DVM, UNR, FDV) \
    hipLaunchKernelGGL((k_vn16<ALG, DVM, UNR, FDV>), vgrid, blk, 0, st, c->d_col_ptr, c->d_col_edge, msg, v2c, prior, marg, live, xbits, n, E, pairs, tiles, vn_chunks, vpw)
        if (dv3) LDPC_VN16(3, 4, 3);
        else if (c->max_dv <= 4) LDPC_VN16(4, 4, 0);
        else if (c->max_dv <= 8) LDPC_VN16(8, 2, 0);
        else if (c->max_dv <= 16) LDPC_VN16(16, 1, 0);
        else if (c->max_dv <= 32) LDPC_VN16(32, 1, 0);
        else LDPC_VN16(64, 1, 0);
#undef LDPC_VN16
        if (d->profile) {
            LDPC_HIP_TRY(hipEventRecord(e2, st));
            spans.push_back({0, e0, e1});
            spans.push_back({1, e1, e2});
        }
        ++sweeps;
    }
    hipLaunchKernelGGL(k_finish_iters, dim3(tiles), dim3(64), 0, st, live, iters, B, sweeps, (const int32_t*)fmap);
    launch_unpack(d, xbits, xhat, B, n, tiles, fmap, st);
    if (soft_out) hipLaunchKernelGGL(k_soft_out16, dim3((n + 3) / 4, pairs), dim3(256), 0, st, marg, soft_out, B, n);  // (no repack with a soft output)
    LDPC_HIP_TRY(hipGetLastError());
    if (d->profile) {
        LDPC_HIP_TRY(hipStreamSynchronize(st));
        LDPC_TRY(prof_collect(d, spans));
    }
    d->last_repacks = repacks;
    d->last_sweeps = sweeps;
    d->last_backend = BK_STREAM;
    return LDPC_OK;
}

}  // namespace

// fp16-storage mode entry points (LDPC_DTYPE_F16): priors are fp32
static int stream_decode_f16(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags, uint8_t* xhat, int32_t* iters,
                             float* soft_out, hipStream_t st, const SimSource* sim) {
    if (B <= 0) return LDPC_OK;
    if (B > (int64_t)65535 * 64) {
        set_error("batch of %lld frames exceeds one launch (max %d); split the call", (long long)B, 65535 * 64);
        return LDPC_E_ARG;
    }
    if (d->alg == ALG_MSA) return run16<ALG_MSA>(d, (const float*)priors, y0, B, max_iter, flags, xhat, iters, soft_out, st, sim);
    if (d->alg == ALG_SPA) return run16<ALG_SPA>(d, (const float*)priors, y0, B, max_iter, flags, xhat, iters, soft_out, st, sim);
    set_error("fp16 storage: LLR decoders (the erasure decoder moves 2 bits per message already)");
    return LDPC_E_UNSUPPORTED;
}


// ldpc_simulate on the streaming kernels, BI-AWGN with the all-`codeword` word: channel + LLR generated into the tiles, then the decode
int stream_simulate_biawgn(Decoder* d, double param, int codeword, uint64_t seed, uint64_t stream_id, uint64_t frame0, int64_t B,
                           int32_t max_iter, uint32_t flags, uint8_t* xhat, int32_t* iters, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    if (d->alg == ALG_BEC || B > (int64_t)65535 * 64) {
        set_error("stream_simulate_biawgn: LLR decoders, at most %d frames per call", 65535 * 64);
        return LDPC_E_ARG;
    }
    const double var = pow(10.0, -param / 10.0);  // src/biawgn.py:10 -- the host arithmetic of channel_generate()
    const SimSource s{sqrt(var), 2.0 / var, codeword, seed, frame0, (uint32_t)stream_id};
    if (d->dtype == DT_F16) return stream_decode_f16(d, nullptr, nullptr, B, max_iter, flags, xhat, iters, nullptr, st, &s);
    if (d->alg == ALG_MSA)
        return d->dtype == DT_F64 ? run<double, ALG_MSA>(d, nullptr, nullptr, B, max_iter, flags, xhat, iters, nullptr, st, &s)
                                  : run<float, ALG_MSA>(d, nullptr, nullptr, B, max_iter, flags, xhat, iters, nullptr, st, &s);
    return d->dtype == DT_F64 ? run<double, ALG_SPA>(d, nullptr, nullptr, B, max_iter, flags, xhat, iters, nullptr, st, &s)
                              : run<float, ALG_SPA>(d, nullptr, nullptr, B, max_iter, flags, xhat, iters, nullptr, st, &s);
}

int stream_decode(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                  uint8_t* xhat, int32_t* iters, void* soft_out, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    if (B > (int64_t)65535 * 64) {
        set_error("batch of %lld frames exceeds one launch (max %d); split the call", (long long)B, 65535 * 64);
        return LDPC_E_ARG;
    }
    if (d->alg == ALG_BEC) {
        if (!y0) {
            set_error("erasure decoder needs the received symbols (y0)");
            return LDPC_E_ARG;
        }
        return becs_stream_decode(d, y0, B, max_iter, flags, xhat, iters, st);  // bit-sliced: ldpc_bec_stream.hip
    }
    if (!priors) {
        set_error("priors pointer is null");
        return LDPC_E_ARG;
    }
    if (d->dtype == DT_F16) return stream_decode_f16(d, priors, y0, B, max_iter, flags, xhat, iters, (float*)soft_out, st, nullptr);
    if (d->alg == ALG_MSA) {
        return d->dtype == DT_F64 ? run<double, ALG_MSA>(d, priors, y0, B, max_iter, flags, xhat, iters, soft_out, st)
                                  : run<float, ALG_MSA>(d, priors, y0, B, max_iter, flags, xhat, iters, soft_out, st);
    }
    return d->dtype == DT_F64 ? run<double, ALG_SPA>(d, priors, y0, B, max_iter, flags, xhat, iters, soft_out, st)
                              : run<float, ALG_SPA>(d, priors, y0, B, max_iter, flags, xhat, iters, soft_out, st);
}

}  // namespace ldpc
