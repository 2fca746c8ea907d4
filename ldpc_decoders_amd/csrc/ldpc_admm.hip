// ADMM LP decoding (Barman, Liu, Draper, Recht: "Decomposition methods for large scale LP decoding") on the GPU -- the
// batched counterpart of the reference's ADMM class:
//
//   iteration, stopping rule, iteration counter ... reference src/admm.py:42-69, 17-23
//   projection onto the parity polytope ........... reference src/parity_polytope/projection.cpp:30-249 (its only native code)
//
// Layout as in the streaming BP backend: tiles of 64 frames, lane == frame, every array [tile][index][64] in HBM, every
// graph index wave-uniform.  All arithmetic is fp64 in the reference's operation order -- column sums from +0.0 in ascending
// edge order (scipy COO), divisions kept as divisions, the two stopping sums in numpy's pairwise order (np_sum below), the
// projection with the same stable decreasing sort and water-filling steps -- so estimates and iteration counts are
// bit-identical to the reference's (golden vectors, tests/test_gpu_admm.py).  oracle/admm_oracle.c is the CPU statement.
#include "ldpc_common.hpp"
#include "ldpc_repack.hpp"

#include <algorithm>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

namespace ldpc {

struct AdmmDecoder {
    Code* code = nullptr;
    DevBuf z, lam, d1, d2, x, gam, live, flags, part;
    DevBuf z2, lam2, x2, gam2, live2, fmap, fmap2, rbase;  // second state set + frame maps of the repack (ldpc_repack.hpp)
    int last_repacks = 0;
    int32_t* d_leaf_off = nullptr;  // numpy's summation blocks of a length-E vector
    int32_t* d_leaf_len = nullptr;
    int32_t* d_prog = nullptr;      // stack program folding the block sums (see k_admm_test)
    int leaves = 0, prog_len = 0;
    void* pinned = nullptr;
    int last_iters = 0;
    // LDS-resident kernel (k_admm_lds): the split tree of numpy's pairwise sum as a level schedule -- node i (leaves first, then the
    // additions) = node lvl_a[i] + node lvl_b[i]; the additions of level l are [lvl_start[l], lvl_start[l + 1])
    int32_t* d_lvl_a = nullptr;
    int32_t* d_lvl_b = nullptr;
    int32_t* d_lvl_start = nullptr;
    int8_t* d_fold_partner = nullptr;  // the same tree as lane-to-lane additions inside one wave (k_admm_lds; <= 32 blocks)
    int levels = 0, nodes = 0;
    unsigned long long* d_ticket = nullptr;  // frame dispenser of the LDS-resident kernel
    int last_backend = 0;                    // 0: streaming kernels, 1: LDS-resident kernel
};

namespace {

using u64 = unsigned long long;
constexpr int PP_MAX = 16;  // std::sort is a stable insertion sort up to 16 elements

__device__ __forceinline__ double clamp01(double x) {
    const double lo = (x < 0.0) ? 0.0 : x;
    return (1.0 < lo) ? 1.0 : lo;
}

// Small per-lane array living in the LDS: element i of this lane at p[i * STRIDE].  The projection indexes its work arrays
// with data-dependent indices; in registers every such access expands into a select chain over the whole array, in the LDS it
// is one ds_read / ds_write (lanes sit on distinct banks for any i: i * STRIDE * sizeof(T) is a multiple of 256 bytes).
template <typename T, int STRIDE>
struct LdsArr {
    T* p;
    __device__ __forceinline__ T& operator[](int i) const { return p[i * STRIDE]; }
};

// Euclidean projection onto the parity polytope; v is overwritten with the result.  Same steps as oracle_pp_project.
// v, s, c, bp: double arrays, who, bp_who: int arrays -- raw pointers to local arrays or LdsArr accessors.
template <typename AD, typename AI>
__device__ void pp_project(AD v, AD s, AD c, AD bp, AI who, AI bp_who, int len) {
    bool none_positive = true, all_above_one = true;
    for (int i = 0; i < len; ++i) {
        if (v[i] > 0) none_positive = false;
        if (v[i] <= 1) all_above_one = false;
    }
    if (none_positive) {
        for (int i = 0; i < len; ++i) v[i] = 0;
        return;
    }
    if (all_above_one && len % 2 == 0) {
        for (int i = 0; i < len; ++i) v[i] = 1;
        return;
    }
    for (int i = 0; i < len; ++i) {  // stable insertion sort, decreasing
        const double val = v[i];
        int j = i;
        while (j > 0 && val > s[j - 1]) {
            s[j] = s[j - 1];
            who[j] = who[j - 1];
            --j;
        }
        s[j] = val;
        who[j] = i;
    }
    double mass = 0;
    for (int i = 0; i < len; ++i) {
        c[i] = clamp01(s[i]);
        mass += c[i];
    }
    int r = (int)floor(mass);
    if (r & 1) --r;
    double facet = 0;
    for (int i = 0; i < r + 1 && i < len; ++i) facet += c[i];  // r == len: upstream reads c[len] (projection.cpp:79-80); taken as 0 here
    for (int i = r + 1; i < len; ++i) facet -= c[i];
    if (facet <= r) {
        for (int i = 0; i < len; ++i) v[who[i]] = c[i];
        return;
    }
    const double beta_cap = (r + 2 <= len) ? (s[r] - s[r + 1]) / 2 : s[r];
    {
        int L = r, R = r + 1, k = 0;
        while (k < len) {
            if (L < 0) {
                for (; k < len; ++k, ++R) { bp_who[k] = R; bp[k] = -s[R]; }
                break;
            }
            if (R >= len) {
                for (; k < len; ++k, --L) { bp_who[k] = L; bp[k] = s[L] - 1; }
                break;
            }
            const double a = s[L] - 1, b = -s[R];
            if (a > b) { bp_who[k] = R; bp[k] = b; ++R; } else { bp_who[k] = L; bp[k] = a; --L; }
            ++k;
        }
    }
    const double tol = 1e-10;
    int clip = -1, zero = 0, first = 0, last = -1;
    for (int i = 0; i < len; ++i) {
        if (s[i] > 1) ++clip;
        if (s[i] >= 0 - tol) ++zero;
        if (bp[i] < 0 + tol) ++first;
        if (bp[i] < beta_cap) ++last;
    }
    double active = 0;
    for (int i = 0; i < len; ++i) {
        if (i > clip && i <= r) active += s[i];
        if (i > r && i < zero) active -= s[i];
    }
    double total = active + clip + 1;
    int prev_clip = clip, prev_zero = zero;
    bool fresh = true;
    double prev_active = active, beta = 0;
    for (int i = first; i <= last; ++i) {
        if (fresh) {
            prev_clip = clip;
            prev_zero = zero;
            prev_active = active;
        }
        fresh = false;
        beta = bp[i];
        if (bp_who[i] <= r) {
            --clip;
            active += s[bp_who[i]];
        } else {
            ++zero;
            active -= s[bp_who[i]];
        }
        if (i < len - 1) {
            if (beta != bp[i + 1]) {
                total = (clip + 1) + active - beta * (zero - clip - 1);
                fresh = true;
                if (total < r) break;
            }
        } else if (i == len - 1) {
            total = (clip + 1) + active - beta * (zero - clip - 1);
            fresh = true;
        }
    }
    if (total > r)
        beta = -(r - clip - 1 - active) / (zero - clip - 1);
    else
        beta = -(r - prev_clip - 1 - prev_active) / (prev_zero - prev_clip - 1);
    for (int i = 0; i < len; ++i) v[who[i]] = clamp01(i <= r ? s[i] - beta : s[i] + beta);
}

// The same projection for a check of exactly L edges, entirely in registers: every loop is unrolled over the L positions and
// predicated, the sort is an odd-even transposition network (adjacent compare-exchanges with a strict comparison never reorder
// equal values, i.e. the same stable decreasing order as the insertion sort above), the two data-dependent reads of the merge
// are select chains.  No LDS arrays -> the kernel's occupancy is set by its registers alone (k_admm_z_fixed).  Same arithmetic,
// operation by operation, as pp_project: results are bit-identical.
template <int L>
__device__ __forceinline__ double dyn_get(const double (&a)[L], int idx) {
    // a select chain that STAYS one: left to itself the optimiser turns the chain back into a[idx], i.e. a private array with a dynamic
    // index -- promoted to the LDS where the kernel's LDS size is static (k_admm_z_fixed: one ds_read per access), but placed in SCRATCH
    // under a dynamic LDS allocation (k_admm_lds: 14 dependent scratch loads per projection, 60 % of the kernel's time spent waiting)
    double r = a[0];
#pragma unroll
    for (int i = 1; i < L; ++i) {
        r = (idx == i) ? a[i] : r;
        asm("" : "+v"(r));
    }
    return r;
}

template <int L>
__device__ __forceinline__ void pp_project_fixed(double (&v)[L]) {
    bool none_positive = true, all_above_one = true;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (v[i] > 0) none_positive = false;
        if (v[i] <= 1) all_above_one = false;
    }
    if (none_positive) {
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = 0;
        return;
    }
    if (L % 2 == 0 && all_above_one) {
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = 1;
        return;
    }
    double s[L];
    int who[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        s[i] = v[i];
        who[i] = i;
    }
#pragma unroll
    for (int round = 0; round < L; ++round) {
#pragma unroll
        for (int i = round & 1; i + 1 < L; i += 2) {
            const bool sw = s[i + 1] > s[i];
            const double hi = sw ? s[i + 1] : s[i], lo = sw ? s[i] : s[i + 1];
            const int wh = sw ? who[i + 1] : who[i], wl = sw ? who[i] : who[i + 1];
            s[i] = hi; s[i + 1] = lo;
            who[i] = wh; who[i + 1] = wl;
        }
    }
    double mass = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) mass += clamp01(s[i]);
    int r = (int)floor(mass);
    if (r & 1) --r;
    double facet = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) facet = (i <= r) ? facet + clamp01(s[i]) : facet - clamp01(s[i]);
    // r == L (even L, every entry clips to 1): upstream reads one element past its arrays here (projection.cpp:79-80); the
    // all-ones point is a vertex of the polytope, and with that element taken as 0 the facet test returns it
    if (facet <= r) {
#pragma unroll
        for (int i = 0; i < L; ++i) v[i] = clamp01(v[i]);  // == v[who[i]] = clamp01(s[i])
        return;
    }
    // members of the first r+1 sorted positions, as a bit mask over the original positions
    unsigned inset = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) inset |= (i <= r) ? (1u << who[i]) : 0u;
    const double s_r = dyn_get<L>(s, r);
    const double beta_cap = (r + 2 <= L) ? (s_r - dyn_get<L>(s, r + 1 < L ? r + 1 : L - 1)) / 2 : s_r;
    // break points, ascending: s[i] - 1 for i = r..0 merged with -s[i] for i = r+1..L-1 (ties: the left sequence first)
    double bp[L], bs[L];  // bs[k] = s[bp_who[k]]
    bool bleft[L];        // bp_who[k] <= r
    {
        int lp = r, rp = r + 1;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const double sl = dyn_get<L>(s, lp < 0 ? 0 : lp), sr = dyn_get<L>(s, rp >= L ? L - 1 : rp);
            const double a = sl - 1, b = -sr;
            const bool take_right = lp < 0 || (rp < L && a > b);
            bp[k] = take_right ? b : a;
            bs[k] = take_right ? sr : sl;
            bleft[k] = !take_right;
            if (take_right) ++rp; else --lp;
        }
    }
    const double tol = 1e-10;
    int clip = -1, zero = 0, first = 0, last = -1;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (s[i] > 1) ++clip;
        if (s[i] >= 0 - tol) ++zero;
        if (bp[i] < 0 + tol) ++first;
        if (bp[i] < beta_cap) ++last;
    }
    double active = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (i > clip && i <= r) active += s[i];
        if (i > r && i < zero) active -= s[i];
    }
    double total = active + clip + 1;
    int prev_clip = clip, prev_zero = zero;
    bool fresh = true, running = true;
    double prev_active = active, beta = 0;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        if (running && i >= first && i <= last) {
            if (fresh) {
                prev_clip = clip;
                prev_zero = zero;
                prev_active = active;
            }
            fresh = false;
            beta = bp[i];
            if (bleft[i]) {
                --clip;
                active += bs[i];
            } else {
                ++zero;
                active -= bs[i];
            }
            if (i < L - 1) {
                if (beta != bp[i + 1 < L ? i + 1 : i]) {
                    total = (clip + 1) + active - beta * (zero - clip - 1);
                    fresh = true;
                    if (total < r) running = false;
                }
            } else {
                total = (clip + 1) + active - beta * (zero - clip - 1);
                fresh = true;
            }
        }
    }
    if (total > r)
        beta = -(r - clip - 1 - active) / (zero - clip - 1);
    else
        beta = -(r - prev_clip - 1 - prev_active) / (prev_zero - prev_clip - 1);
#pragma unroll
    for (int i = 0; i < L; ++i) v[i] = clamp01(((inset >> i) & 1u) ? v[i] - beta : v[i] + beta);
}

// gamma [B,n] -> gam[tile][n][64]; state: z = 0.5, lambda = 0 (src/admm.py:44)
__global__ __launch_bounds__(256) void k_admm_init(const double* __restrict__ gamma, int64_t B, int n, int64_t E, double* __restrict__ gam,
                                                   double* __restrict__ z, double* __restrict__ lam, double* __restrict__ x) {
    const int tile = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t fr = (int64_t)tile * 64 + lane;
    for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < E; k += (int64_t)gridDim.x * 4) {
        z[((int64_t)tile * E + k) * 64 + lane] = 0.5;
        lam[((int64_t)tile * E + k) * 64 + lane] = 0.0;
    }
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < n; v += gridDim.x * 4) {
        gam[((int64_t)tile * n + v) * 64 + lane] = fr < B ? gamma[fr * n + v] : 0.0;
        x[((int64_t)tile * n + v) * 64 + lane] = 0.0;
    }
}

__global__ void k_admm_live(u64* __restrict__ live, int64_t B, int tiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const int64_t rem = B - (int64_t)t * 64;
    live[t] = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
}

// x update (src/admm.py:54-55): clip((sum_cols(z - lambda/mu) - gamma/mu) / var_deg, 0, 1)
__global__ __launch_bounds__(256) void k_admm_x(const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ col_edge,
                                                const double* __restrict__ z, const double* __restrict__ lam, const double* __restrict__ gam,
                                                double* __restrict__ x, const u64* __restrict__ live, int n, int64_t E, int tiles, double mu) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    const u64 lv = live[tile];
    if (lv == 0 || !((lv >> lane) & 1ull)) return;
    const double* zt = z + (int64_t)tile * E * 64 + lane;
    const double* lt = lam + (int64_t)tile * E * 64 + lane;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < n; v += gridDim.x * 4) {
        const int p0 = col_ptr[v], p1 = col_ptr[v + 1];
        double s = 0.0;
        for (int p = p0; p < p1; ++p) {
            const int64_t k = col_edge[p];
            s += zt[k * 64] - lt[k * 64] / mu;
        }
        const int64_t o = ((int64_t)tile * n + v) * 64 + lane;
        x[o] = clamp01((s - gam[o] / mu) / (double)(p1 - p0));
    }
}

// z and lambda updates (src/admm.py:58-63) + the two squared-distance vectors of the stopping test (src/admm.py:18-19)
template <int DCM, bool LDSARR>
__global__ __launch_bounds__(LDSARR ? 128 : 256) void k_admm_z(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ edge_var,
                                                               double* __restrict__ z, double* __restrict__ lam, const double* __restrict__ x,
                                                               double* __restrict__ d1, double* __restrict__ d2, const u64* __restrict__ live,
                                                               int m, int n, int64_t E, double mu) {
    constexpr int TPB = LDSARR ? 128 : 256;
    __shared__ double sh_d[LDSARR ? 4 : 1][LDSARR ? DCM : 1][LDSARR ? TPB : 1];
    __shared__ int sh_i[LDSARR ? 2 : 1][LDSARR ? DCM : 1][LDSARR ? TPB : 1];
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    const u64 lv = live[tile];
    if (lv == 0 || !((lv >> lane) & 1ull)) return;
    const int64_t eb = (int64_t)tile * E * 64 + lane;
    const double* xt = x + (int64_t)tile * n * 64 + lane;
    for (int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); c < m; c += gridDim.x * (TPB / 64)) {
        const int k0 = row_ptr[c], len = row_ptr[c + 1] - k0;
        double xs[DCM], lm[DCM];
        if constexpr (LDSARR) {
            const LdsArr<double, TPB> v{&sh_d[0][0][threadIdx.x]}, s{&sh_d[1][0][threadIdx.x]}, cc{&sh_d[2][0][threadIdx.x]}, bp{&sh_d[3][0][threadIdx.x]};
            const LdsArr<int, TPB> who{&sh_i[0][0][threadIdx.x]}, bp_who{&sh_i[1][0][threadIdx.x]};
            for (int j = 0; j < len; ++j) {
                xs[j] = xt[(int64_t)edge_var[k0 + j] * 64];
                lm[j] = lam[eb + (int64_t)(k0 + j) * 64];
                v[j] = xs[j] + lm[j] / mu;
            }
            pp_project(v, s, cc, bp, who, bp_who, len);
            for (int j = 0; j < len; ++j) {
                const int64_t o = eb + (int64_t)(k0 + j) * 64;
                const double zo = z[o], zn = v[j];
                lam[o] = lm[j] + mu * (xs[j] - zn);
                const double a = xs[j] - zn, b = zo - zn;
                d1[o] = a * a;
                d2[o] = b * b;
                z[o] = zn;
            }
        } else {
            double v[DCM], s[DCM], cc[DCM], bp[DCM];
            int who[DCM], bp_who[DCM];
            for (int j = 0; j < len; ++j) {
                xs[j] = xt[(int64_t)edge_var[k0 + j] * 64];
                lm[j] = lam[eb + (int64_t)(k0 + j) * 64];
                v[j] = xs[j] + lm[j] / mu;
            }
            pp_project((double*)v, (double*)s, (double*)cc, (double*)bp, (int*)who, (int*)bp_who, len);
            for (int j = 0; j < len; ++j) {
                const int64_t o = eb + (int64_t)(k0 + j) * 64;
                const double zo = z[o];
                lam[o] = lm[j] + mu * (xs[j] - v[j]);
                const double a = xs[j] - v[j], b = zo - v[j];
                d1[o] = a * a;
                d2[o] = b * b;
                z[o] = v[j];
            }
        }
    }
}

// The same for codes whose checks all have exactly L edges: register-only projection, edge k0 + j of check c at c * L + j
template <int L>
__global__ __launch_bounds__(256) void k_admm_z_fixed(const int32_t* __restrict__ edge_var, double* __restrict__ z, double* __restrict__ lam,
                                                      const double* __restrict__ x, double* __restrict__ d1, double* __restrict__ d2,
                                                      const u64* __restrict__ live, int m, int n, int64_t E, double mu) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    const u64 lv = live[tile];
    if (lv == 0 || !((lv >> lane) & 1ull)) return;
    const int64_t eb = (int64_t)tile * E * 64 + lane;
    const double* xt = x + (int64_t)tile * n * 64 + lane;
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < m; c += gridDim.x * 4) {
        const int k0 = c * L;
        double xs[L], lm[L], zo[L], v[L];
#pragma unroll
        for (int j = 0; j < L; ++j) {
            xs[j] = xt[(int64_t)edge_var[k0 + j] * 64];
            lm[j] = lam[eb + (int64_t)(k0 + j) * 64];
            zo[j] = z[eb + (int64_t)(k0 + j) * 64];
        }
#pragma unroll
        for (int j = 0; j < L; ++j) v[j] = xs[j] + lm[j] / mu;
        pp_project_fixed<L>(v);
#pragma unroll
        for (int j = 0; j < L; ++j) {
            const int64_t o = eb + (int64_t)(k0 + j) * 64;
            lam[o] = lm[j] + mu * (xs[j] - v[j]);
            const double a = xs[j] - v[j], b = zo[j] - v[j];
            d1[o] = a * a;
            d2[o] = b * b;
            z[o] = v[j];
        }
    }
}

// numpy's pairwise sum of the E squared distances (np.add.reduce of a contiguous float64 vector): the vector is split
// recursively (n2 = n/2 rounded down to a multiple of 8) down to blocks of at most 128 elements, each block is summed with 8
// strided accumulators, and the block sums are added back up the split tree.  The blocks are independent, so they are summed
// by one wave each (k_admm_leaves); the tree is then folded by a tiny stack program built on the host (k_admm_test):
// op >= 0: push the sum of block `op`; op == -1: pop b, pop a, push a + b.
__device__ double np_block(const double* a, int n) {  // n <= 128, element stride 64 doubles
    double res;
    if (n < 8) {
        res = -0.0;
        for (int i = 0; i < n; ++i) res += a[(int64_t)i * 64];
    } else {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = a[(int64_t)k * 64];
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += a[(int64_t)(i + k) * 64];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[(int64_t)i * 64];
    }
    return res;
}

__global__ __launch_bounds__(256) void k_admm_leaves(const double* __restrict__ d1, const double* __restrict__ d2,
                                                     const int32_t* __restrict__ leaf_off, const int32_t* __restrict__ leaf_len, int leaves,
                                                     double* __restrict__ part, const u64* __restrict__ live, int64_t E) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    const u64 lv = live[tile];
    if (lv == 0 || !((lv >> lane) & 1ull)) return;
    const int leaf = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (leaf >= leaves) return;
    const int64_t base = ((int64_t)tile * E + leaf_off[leaf]) * 64 + lane;
    double* o = part + ((int64_t)tile * leaves + leaf) * 128 + lane;
    o[0] = np_block(d1 + base, leaf_len[leaf]);
    o[64] = np_block(d2 + base, leaf_len[leaf]);
}

// stopping test (src/admm.py:21, 65) and the max_iter exit of the next loop head (src/admm.py:51); one wave per tile
__global__ __launch_bounds__(64) void k_admm_test(const double* __restrict__ part, const int32_t* __restrict__ prog, int prog_len, int leaves,
                                                  u64* __restrict__ live, int32_t* __restrict__ iters, uint8_t* __restrict__ converged,
                                                  int* __restrict__ live_tiles, int64_t B, double thresh, int it, int max_iter,
                                                  const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.x, lane = threadIdx.x;
    const u64 lv = live[tile];
    if (lv == 0) return;
    const bool on = (lv >> lane) & 1ull;
    bool close = false;
    if (on) {
        const double* p = part + (int64_t)tile * leaves * 128 + lane;
        double s1[40], s2[40];
        int sp = 0;
        for (int i = 0; i < prog_len; ++i) {
            const int op = prog[i];
            if (op >= 0) {
                s1[sp] = p[(int64_t)op * 128];
                s2[sp] = p[(int64_t)op * 128 + 64];
                ++sp;
            } else {
                --sp;
                s1[sp - 1] = s1[sp - 1] + s1[sp];
                s2[sp - 1] = s2[sp - 1] + s2[sp];
            }
        }
        const double aa1 = 0.0 + s1[0], aa2 = 0.0 + s2[0];
        close = aa1 < thresh && aa2 < thresh;
    }
    const bool capped = max_iter > 0 && it + 1 >= max_iter;
    const int64_t fr = frame_of ? (int64_t)frame_of[(int64_t)tile * 64 + lane] : (int64_t)tile * 64 + lane;
    if (on && fr >= 0 && fr < B) {
        if (close) {
            iters[fr] = it;
            if (converged) converged[fr] = 1;
        } else if (capped) {
            iters[fr] = it + 1;
        }
    }
    const u64 leave = capped ? lv : __ballot(on && close);
    const u64 stay = lv & ~leave;
    if (lane == 0) {
        live[tile] = stay;
        if (stay && live_tiles) {
            atomicAdd(live_tiles, 1);
            atomicAdd(live_tiles + 1, __popcll(stay));
        }
    }
}

__global__ void k_admm_out(const double* __restrict__ x, double* __restrict__ out, int64_t B, int n, const int32_t* __restrict__ frame_of) {
    const int tile = blockIdx.y, lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t fr = frame_of ? (int64_t)frame_of[(int64_t)tile * 64 + lane] : (int64_t)tile * 64 + lane;
    if (v < n && fr >= 0 && fr < B) out[fr * n + v] = x[((int64_t)tile * n + v) * 64 + lane];
}

// live frames -> dense tiles: z, lambda (E lines each) and x, gamma (n lines each) of the frame of rank j move to (j / 64, j % 64)
__global__ __launch_bounds__(256) void k_admm_repack(const double* __restrict__ z_s, double* __restrict__ z_d, const double* __restrict__ l_s,
                                                     double* __restrict__ l_d, const double* __restrict__ x_s, double* __restrict__ x_d,
                                                     const double* __restrict__ g_s, double* __restrict__ g_d, const u64* __restrict__ live_src,
                                                     u64* __restrict__ live_dst, const int32_t* __restrict__ base,
                                                     const int32_t* __restrict__ frame_src, int32_t* __restrict__ frame_dst, int tiles_src, int n,
                                                     int64_t E, int rows_per_wave) {
    const int lane = threadIdx.x;
    const int dt = blockIdx.y;
    int st, sl;
    const bool has = repack_source(base, live_src, tiles_src, dt * 64 + lane, &st, &sl);
    const int chunk = blockIdx.x * 4 + threadIdx.y;
    const int64_t rows = E + n;
    const int64_t r0 = (int64_t)chunk * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    if (has) {
        const int64_t es = (int64_t)st * E * 64 + sl, ed = (int64_t)dt * E * 64 + lane;
        const int64_t vs = (int64_t)st * n * 64 + sl, vd = (int64_t)dt * n * 64 + lane;
        for (int64_t r = r0; r < r1; ++r) {
            if (r < E) {
                z_d[ed + r * 64] = z_s[es + r * 64];
                l_d[ed + r * 64] = l_s[es + r * 64];
            } else {
                x_d[vd + (r - E) * 64] = x_s[vs + (r - E) * 64];
                g_d[vd + (r - E) * 64] = g_s[vs + (r - E) * 64];
            }
        }
    }
    if (chunk == 0) {
        frame_dst[(int64_t)dt * 64 + lane] = has ? (frame_src ? frame_src[(int64_t)st * 64 + sl] : st * 64 + sl) : -1;
        const u64 lv = __ballot(has);
        if (lane == 0) live_dst[dt] = lv;
    }
}


// =====================================================================================================================================
// LDS-resident ADMM (round 6): ONE workgroup owns ONE frame for all its iterations; z, lambda, lambda / mu, the two squared-distance
// vectors of the stopping test and x live in the LDS of the CU (n = 1200 (3,6): 5 x 28.8 KB + 9.6 KB), only gamma in / the estimate out
// touch HBM.  The streaming kernels above move 8 (9E + 2n) bytes per frame-iteration through HBM and keep every tile in lockstep with
// its slowest frame; here a frame leaves at its own iteration (src/admm.py:65-66) and the workgroup takes the next one from a
// dispenser.  Same arithmetic, operation by operation (pp_project_fixed; the ordered column sum; numpy's pairwise stopping sums):
// estimates and iteration counts are bit-identical to the streaming kernels' and the reference's.
//
// Codes whose checks all have L edges (edge k = L c + j).  Edge arrays are kept POSITION-major in the LDS -- element (c, j) at
// j * m + c -- so that the check phase (lane == check) reads and writes them lane-contiguously; the variable phase gathers.
//   phase A  x update (src/admm.py:54-55), lane == variable (VPL variables per lane): s = sum over its edges in ascending edge order
//            of z - lambda / mu; x = clip((s - gamma / mu) / deg).  lambda / mu is kept from the check phase (the same division,
//            done once), gamma / mu once per frame.
//   phase B  z / lambda update (src/admm.py:58-63), lane == check: projection in registers (pp_project_fixed), d1 = (x - z')^2,
//            d2 = (z - z')^2 (src/admm.py:18-19).
//   phase C  stopping test (src/admm.py:21,65): numpy's pairwise sums of d1 and d2 -- per block of <= 128 elements eight strided
//            accumulators (one lane per accumulator chain), the block sums folded up the split tree level by level (one wave).
struct AdmmLdsArgs {
    const double* gamma;
    double* x_out;
    int32_t* iters;
    uint8_t* converged;
    const int32_t *col_ptr, *col_edge, *edge_var, *leaf_off, *leaf_len, *lvl_a, *lvl_b, *lvl_start;
    const int8_t* fold_partner;  // [levels][64]: lane arr * 32 + leaf adds the value of that lane at that level (-1: none); null: more than 32 blocks
    int m, n, E, leaves, levels, nodes, val_doubles;  // val_doubles: size of the `val` area (>= 2 * nodes, >= 64)
    long long B;
    double mu, thresh;
    int max_iter, cap;
    unsigned long long* ticket;
};

template <int L, int DVMAX, int VPL, int NW, int CPL>
__global__ __launch_bounds__(64 * NW) void k_admm_lds(const AdmmLdsArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, T = 64 * NW;
    const int m = A.m, n = A.n, E = A.E;
    double* const z = reinterpret_cast<double*>(smem);
    double* const lam = z + E;
    double* const q = lam + E;    // lambda / mu
    double* const d1 = q + E;
    double* const d2 = d1 + E;
    double* const x = d2 + E;
    double* const racc = x + n;                        // [2][leaves][8] accumulator chains
    double* const val = racc + 2 * A.leaves * 8;        // [2][nodes] block sums and the tree above them
    volatile int* const word = reinterpret_cast<volatile int*>(val + A.val_doubles);  // [0] frame hand-out (lo), [1] (hi), [2] verdict
    // the schedule of the stopping sums, copied into the LDS once per workgroup (a global load per tree level on the one wave that
    // folds the tree was a third of the iteration): additions of the tree (lvl_a, lvl_b per node), level boundaries, blocks (offset, length)
    int* const s_lvl_a = const_cast<int*>(reinterpret_cast<volatile int*>(word)) + 16;
    int* const s_lvl_b = s_lvl_a + A.nodes;
    int* const s_lvl_start = s_lvl_b + A.nodes;
    int* const s_leaf_off = s_lvl_start + A.levels + 1;
    int* const s_leaf_len = s_leaf_off + A.leaves;
    const double mu = A.mu;
    auto eidx = [&](int k) { return (k % L) * m + k / L; };  // canonical edge index -> LDS position
    for (int i = tid; i < A.nodes; i += T) {
        s_lvl_a[i] = A.lvl_a[i];
        s_lvl_b[i] = A.lvl_b[i];
    }
    for (int i = tid; i <= A.levels; i += T) s_lvl_start[i] = A.lvl_start[i];
    for (int i = tid; i < A.leaves; i += T) {
        s_leaf_off[i] = A.leaf_off[i];
        s_leaf_len[i] = A.leaf_len[i];
    }
    // graph indices of this lane, resident for the whole launch: LDS positions of the edges of its variables (ascending edge order),
    // variables of its check's edges, its accumulator chain of the stopping sums
    int vk[VPL][DVMAX], vdeg[VPL];
#pragma unroll
    for (int r = 0; r < VPL; ++r) {
        const int v = tid + r * T;
        const int p0 = v < n ? A.col_ptr[v] : 0, p1 = v < n ? A.col_ptr[v + 1] : 0;
        vdeg[r] = p1 - p0;
#pragma unroll
        for (int j = 0; j < DVMAX; ++j) vk[r][j] = eidx(A.col_edge[p0 + j < p1 ? p0 + j : p0]);
    }
    int ev[CPL][L];  // CPL checks per lane: check tid + cp * T
#pragma unroll
    for (int cp = 0; cp < CPL; ++cp)
#pragma unroll
        for (int j = 0; j < L; ++j) ev[cp][j] = tid + cp * T < m ? A.edge_var[(tid + cp * T) * L + j] : 0;
    // chain `tid`: array (d1 / d2), block, accumulator; at most one per lane (2 * 8 * leaves chains, leaves ~ E / 112, T >= E / L)
    const int chain_arr = tid / (A.leaves * 8), chain_rem = tid - chain_arr * (A.leaves * 8);
    const bool has_chain = tid < 2 * A.leaves * 8;
    const int chain_off = has_chain ? A.leaf_off[chain_rem >> 3] + (chain_rem & 7) : 0;
    const int chain_len = has_chain ? A.leaf_len[chain_rem >> 3] : 0;
    // wave 0: the lanes this lane adds in at each level of the split tree (lane arr * 32 + block holds that block's / subtree's sum)
    int fold_p[8];
#pragma unroll
    for (int lv = 0; lv < 8; ++lv) fold_p[lv] = (A.fold_partner && tid < 64 && lv < A.levels) ? (int)A.fold_partner[lv * 64 + tid] : -1;

    for (;;) {
        // ---- next frame
        __syncthreads();  // everybody is done with the previous frame's LDS state and hand-out words (first trip: the tables above are written)
        if (tid == 0) {
            const unsigned long long t = atomicAdd(A.ticket, 1ull);
            word[0] = (int)(unsigned)(t & 0xffffffffull);
            word[1] = (int)(unsigned)(t >> 32);
        }
        __syncthreads();
        const long long fr = (long long)(((unsigned long long)(unsigned)word[1] << 32) | (unsigned)word[0]);
        if (fr >= A.B) break;
        // ---- state: z = 0.5, lambda = 0 (src/admm.py:44); gamma / mu in registers
        for (int k = tid; k < E; k += T) {
            z[k] = 0.5;
            lam[k] = 0.0;
            q[k] = 0.0 / mu;
        }
        double gq[VPL];
#pragma unroll
        for (int r = 0; r < VPL; ++r) {
            const int v = tid + r * T;
            gq[r] = v < n ? A.gamma[fr * n + v] / mu : 0.0;
        }
        __syncthreads();
        int it = 0, result_iters = 0;
        bool conv = false;
        for (;; ++it) {
            // ---- phase A: x update
#pragma unroll
            for (int r = 0; r < VPL; ++r) {
                const int v = tid + r * T;
                if (v < n) {
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < DVMAX; ++j) {
                        const double t = z[vk[r][j]] - q[vk[r][j]];
                        s = (j < vdeg[r]) ? s + t : s;
                    }
                    x[v] = clamp01((s - gq[r]) / (double)vdeg[r]);
                }
            }
            __syncthreads();
            // ---- phase B: z / lambda update, lane == check (CPL passes: T lanes take the checks T at a time)
#pragma unroll 1
            for (int cp = 0; cp < CPL; ++cp) {
                const int c = tid + cp * T;
                if (c < m) {
                    double xs[L], lm[L], zo[L], v[L];
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        const int o = j * m + c;
                        xs[j] = x[CPL == 1 ? ev[0][j] : (cp == 0 ? ev[0][j] : ev[CPL - 1][j])];
                        lm[j] = lam[o];
                        zo[j] = z[o];
                        v[j] = xs[j] + q[o];
                    }
                    pp_project_fixed<L>(v);
#pragma unroll
                    for (int j = 0; j < L; ++j) {
                        const int o = j * m + c;
                        const double a = xs[j] - v[j], b = zo[j] - v[j];
                        const double ln = lm[j] + mu * a;
                        lam[o] = ln;
                        q[o] = ln / mu;
                        d1[o] = a * a;
                        d2[o] = b * b;
                        z[o] = v[j];
                    }
                }
            }
            __syncthreads();
            // ---- phase C1: the eight strided accumulators of every block (np_block), one lane per chain
            double rsum = 0.0;
            if (has_chain && chain_len >= 8) {
                // a block has at most 128 elements, 16 per accumulator: every load of the chain is issued before the first add (the adds stay in
                // np_block's order); element positions past the chain read the chain's first element and are not added
                const double* a = chain_arr ? d2 : d1;
                const int nel = (chain_len - (chain_len % 8)) >> 3;  // elements of this chain, >= 1
                double e[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) e[t] = a[eidx(chain_off + (t < nel ? 8 * t : 0))];
                rsum = e[0];
#pragma unroll
                for (int t = 1; t < 16; ++t) rsum = t < nel ? rsum + e[t] : rsum;
                if (!A.fold_partner) racc[tid] = rsum;
            }
            if (A.fold_partner) {
                // the eight accumulators of a block sit on eight consecutive lanes of one wave: ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) by three
                // lane shifts, then the block's tail (np_block) on the lane of accumulator 0, which hands the block sum to wave 0
                const double p1 = rsum + __shfl_down(rsum, 1, 64);
                const double p2 = p1 + __shfl_down(p1, 2, 64);
                const double p4 = p2 + __shfl_down(p2, 4, 64);
                if (has_chain && (chain_rem & 7) == 0) {
                    const double* a = chain_arr ? d2 : d1;
                    const int i0 = chain_len < 8 ? 0 : chain_len - (chain_len % 8), ntail = chain_len - i0;  // at most 7 elements, added one by one
                    double tail[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) tail[t] = a[eidx(chain_off + (t < ntail ? i0 + t : 0))];
                    double res = chain_len < 8 ? -0.0 : p4;
#pragma unroll
                    for (int t = 0; t < 7; ++t) res = t < ntail ? res + tail[t] : res;
                    val[chain_arr * 32 + (chain_rem >> 3)] = res;
                }
            }
            __syncthreads();
            // ---- phase C2 + C3 (one wave): block sums, then the additions of the split tree level by level
            if (tid < 64 && A.fold_partner) {
                // <= 32 blocks: lane arr * 32 + block holds the block sum; the tree is folded lane to lane (two ds_bpermute per level instead
                // of an LDS round trip per index, value and result)
                const int leaf = tid & 31;
                double res = leaf < A.leaves ? val[tid] : 0.0;
#pragma unroll
                for (int lv = 0; lv < 8; ++lv) {
                    if (lv < A.levels) {  // wave-uniform
                        const double other = __shfl(res, fold_p[lv] >= 0 ? fold_p[lv] : tid, 64);
                        res = fold_p[lv] >= 0 ? res + other : res;
                    }
                }
                const bool far = (leaf == 0) && !((0.0 + res) < A.thresh);  // lanes 0 and 32 hold the two totals (src/admm.py:21)
                const bool any_far = __ballot(far) != 0ull;
                if (tid == 0) word[2] = any_far ? 0 : 1;
            } else if (tid < 64) {
                for (int idx = tid; idx < 2 * A.leaves; idx += 64) {
                    const int arr = idx / A.leaves, leaf = idx - arr * A.leaves;
                    const int off = s_leaf_off[leaf], len = s_leaf_len[leaf];
                    const double* a = arr ? d2 : d1;
                    const int i0 = len < 8 ? 0 : len - (len % 8), ntail = len - i0;  // np_block's tail: at most 7 elements, added one by one
                    double tail[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) tail[t] = a[eidx(off + (t < ntail ? i0 + t : 0))];
                    double res;
                    if (len < 8) {
                        res = -0.0;
                    } else {
                        const double* r8 = racc + (arr * A.leaves + leaf) * 8;
                        res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                    }
#pragma unroll
                    for (int t = 0; t < 7; ++t) res = t < ntail ? res + tail[t] : res;
                    val[arr * A.nodes + leaf] = res;
                }
                __builtin_amdgcn_wave_barrier();
                for (int lv = 0; lv < A.levels; ++lv) {
                    const int s0 = s_lvl_start[lv], s1 = s_lvl_start[lv + 1];
                    for (int idx = tid; idx < 2 * (s1 - s0); idx += 64) {
                        const int arr = idx / (s1 - s0), node = s0 + idx - arr * (s1 - s0);
                        val[arr * A.nodes + node] = val[arr * A.nodes + s_lvl_a[node]] + val[arr * A.nodes + s_lvl_b[node]];
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (tid == 0) {
                    const double aa1 = 0.0 + val[A.nodes - 1], aa2 = 0.0 + val[2 * A.nodes - 1];
                    word[2] = (aa1 < A.thresh && aa2 < A.thresh) ? 1 : 0;
                }
            }
            __syncthreads();
            const bool close = word[2] != 0;
            const bool capped = (A.max_iter > 0 && it + 1 >= A.max_iter) || it + 1 >= A.cap;
            if (close) {
                conv = true;
                result_iters = it;
                break;
            }
            if (capped) {
                result_iters = it + 1;
                break;
            }
        }
        // ---- estimate out (the x of the iteration that ended the loop, src/admm.py:66-69)
#pragma unroll
        for (int r = 0; r < VPL; ++r) {
            const int v = tid + r * T;
            if (v < n) A.x_out[fr * n + v] = x[v];
        }
        if (tid == 0) {
            A.iters[fr] = result_iters;
            if (A.converged) A.converged[fr] = conv ? 1 : 0;
        }
    }
}

}  // namespace

int admm_create(Code* code, AdmmDecoder** out) {
    if (!code || !out) return LDPC_E_ARG;
    if (code->max_dc > PP_MAX) {
        set_error("ADMM: check degree %d above %d (the projection's sort order is only defined up to there)", code->max_dc, PP_MAX);
        return LDPC_E_UNSUPPORTED;
    }
    AdmmDecoder* d = new (std::nothrow) AdmmDecoder();
    if (!d) return LDPC_E_NOMEM;
    d->code = code;
    std::vector<int32_t> off, len, prog;
    struct Split {
        static void run(int64_t o, int64_t n, std::vector<int32_t>& off, std::vector<int32_t>& len, std::vector<int32_t>& prog) {
            if (n <= 128) {
                prog.push_back((int32_t)off.size());
                off.push_back((int32_t)o);
                len.push_back((int32_t)n);
                return;
            }
            int64_t n2 = n / 2;
            n2 -= n2 % 8;
            run(o, n2, off, len, prog);
            run(o + n2, n - n2, off, len, prog);
            prog.push_back(-1);
        }
    };
    Split::run(0, code->E, off, len, prog);
    d->leaves = (int)off.size();
    d->prog_len = (int)prog.size();
    // the same tree as a level schedule (k_admm_lds): replay the stack program, every addition becomes a node one level above its deeper child
    std::vector<int32_t> lvl_a, lvl_b, lvl_start;
    std::vector<int8_t> fold;
    {
        const int leaves = d->leaves;
        std::vector<int> stack, node_level((size_t)leaves, 0), ea, eb;  // additions in program order
        for (int32_t op : prog) {
            if (op >= 0) {
                stack.push_back(op);
            } else {
                const int b = stack.back(); stack.pop_back();
                const int a = stack.back(); stack.pop_back();
                const int id = leaves + (int)ea.size();
                ea.push_back(a);
                eb.push_back(b);
                node_level.push_back(1 + std::max(node_level[a], node_level[b]));
                stack.push_back(id);
            }
        }
        // renumber the additions level by level (the root last), children before parents
        int maxl = 0;
        for (int v : node_level) maxl = std::max(maxl, v);
        std::vector<int> newid((size_t)leaves + ea.size());
        for (int i = 0; i < leaves; ++i) newid[i] = i;
        int next = leaves;
        lvl_a.assign((size_t)leaves + ea.size(), 0);
        lvl_b.assign((size_t)leaves + ea.size(), 0);
        for (int l = 1; l <= maxl; ++l) {
            lvl_start.push_back(next);
            for (size_t i = 0; i < ea.size(); ++i)
                if (node_level[leaves + i] == l) {
                    newid[leaves + i] = next;
                    lvl_a[next] = newid[ea[i]];
                    lvl_b[next] = newid[eb[i]];
                    ++next;
                }
        }
        lvl_start.push_back(next);
        d->levels = maxl;
        d->nodes = next;
        // and as a fold inside ONE wave: a node's value lives in the lane of its leftmost block (lane arr * 32 + block); at the node's level
        // that lane adds the value of its right child's lane
        if (leaves <= 32 && maxl <= 8) {
            std::vector<int> home((size_t)leaves + ea.size());
            for (int i = 0; i < leaves; ++i) home[i] = i;
            for (size_t i = 0; i < ea.size(); ++i) home[leaves + i] = home[ea[i]];
            fold.assign((size_t)std::max(maxl, 1) * 64, (int8_t)-1);
            for (size_t i = 0; i < ea.size(); ++i)
                for (int arr = 0; arr < 2; ++arr) fold[(size_t)(node_level[leaves + i] - 1) * 64 + arr * 32 + home[ea[i]]] = (int8_t)(arr * 32 + home[eb[i]]);
        }
    }
    hipError_t e = hipSetDevice(code->device);
    if (e == hipSuccess) e = hipHostMalloc(&d->pinned, 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_leaf_off, off.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_leaf_len, len.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_prog, prog.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d->d_leaf_off, off.data(), off.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_leaf_len, len.data(), len.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_prog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_lvl_a, lvl_a.size() * 4 + 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_lvl_b, lvl_b.size() * 4 + 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_lvl_start, lvl_start.size() * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d->d_ticket, 64);
    if (e == hipSuccess && !fold.empty()) e = hipMalloc((void**)&d->d_fold_partner, fold.size());
    if (e == hipSuccess && !fold.empty()) e = hipMemcpy(d->d_fold_partner, fold.data(), fold.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_lvl_a, lvl_a.data(), lvl_a.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_lvl_b, lvl_b.data(), lvl_b.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->d_lvl_start, lvl_start.data(), lvl_start.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("admm_create: %s", hipGetErrorString(e));
        admm_destroy(d);  // frees whatever was allocated
        return LDPC_E_HIP;
    }
    *out = d;
    return LDPC_OK;
}

int admm_last_repacks(const AdmmDecoder* d) { return d ? d->last_repacks : 0; }
int admm_last_backend(const AdmmDecoder* d) { return d ? d->last_backend : 0; }

void admm_destroy(AdmmDecoder* d) {
    if (!d) return;
    for (DevBuf* b : {&d->z, &d->lam, &d->d1, &d->d2, &d->x, &d->gam, &d->live, &d->flags, &d->part, &d->z2, &d->lam2, &d->x2, &d->gam2, &d->live2, &d->fmap, &d->fmap2,
                      &d->rbase})
        b->release();
    for (void* q : {(void*)d->d_leaf_off, (void*)d->d_leaf_len, (void*)d->d_prog, (void*)d->d_lvl_a, (void*)d->d_lvl_b, (void*)d->d_lvl_start, (void*)d->d_ticket, (void*)d->d_fold_partner})
        if (q) (void)hipFree(q);
    if (d->pinned) (void)hipHostFree(d->pinned);
    delete d;
}

// LDS-resident path of admm_decode: returns 1 where the code is not eligible (checks of unequal degree, a frame beyond the LDS, a code
// too small to fill a workgroup -- the streaming kernels serve those), else LDPC_OK / an error.  LDPC_ADMM_BACKEND=stream forces the
// streaming kernels (A/B and the parity tests of both).
template <int L, int DVMAX, int VPL, int NW, int CPL>
static int admm_launch_lds(const AdmmLdsArgs& a, size_t lds_bytes, int grid, hipStream_t st) {
    static_assert(CPL <= 2, "phase B selects between the first and the last pass's index registers");
    LDPC_HIP_TRY(hipFuncSetAttribute((const void*)k_admm_lds<L, DVMAX, VPL, NW, CPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((k_admm_lds<L, DVMAX, VPL, NW, CPL>), dim3(grid), dim3(64 * NW), lds_bytes, st, a);
    return LDPC_OK;
}

static int admm_decode_lds(AdmmDecoder* d, const double* gamma, int64_t B, double mu, double eps, int32_t max_iter, double* x_out, int32_t* iters,
                           uint8_t* converged, hipStream_t st) {
    const Code* c = d->code;
    if (const char* e = std::getenv("LDPC_ADMM_BACKEND"))
        if (std::string(e) == "stream") return 1;
    if (c->min_dc != c->max_dc || c->max_dc != 6 || c->max_dv > 3 || c->min_dv < 1) return 1;  // built: (3,6)-type codes (every check six edges, variables up to three)
    if (c->m < 128) return 1;  // a frame must fill a workgroup: tiny codes keep the streaming kernels (64 frames per wave)
    // waves per frame: 4 or 8 with one check per lane (m <= 256 / 512); beyond (n = 1200: m = 600) eight waves take the checks in TWO passes
    // -- ten waves at one check each were measured no faster (three wave-projections per SIMD either way) and leave the compiler 170
    // registers where the projection wants 200 (49 spilled)
    const int rows = (c->m + 63) / 64;
    const int nw = rows <= 4 ? 4 : 8, cpl = rows <= 8 ? 1 : 2;
    if (rows > 16 || c->n > (cpl == 1 ? 2 : 3) * 64 * nw) return 1;
    if (2 * d->leaves * 8 > 64 * nw) return 1;  // one accumulator chain of the stopping sums per lane
    const int val_doubles = std::max(2 * d->nodes, 64);
    const size_t lds_bytes = ((size_t)5 * c->E + c->n + (size_t)2 * d->leaves * 8 + (size_t)val_doubles) * 8 + 64 +
                             ((size_t)2 * d->nodes + d->levels + 1 + (size_t)2 * d->leaves) * 4 + 16;  // + the schedule tables (ints)
    if (lds_bytes > (size_t)160 * 1024) return 1;
    AdmmLdsArgs a;
    a.gamma = gamma; a.x_out = x_out; a.iters = iters; a.converged = converged;
    a.col_ptr = c->d_col_ptr; a.col_edge = c->d_col_edge; a.edge_var = c->d_edge_var;
    a.leaf_off = d->d_leaf_off; a.leaf_len = d->d_leaf_len; a.lvl_a = d->d_lvl_a; a.lvl_b = d->d_lvl_b; a.lvl_start = d->d_lvl_start;
    a.fold_partner = d->d_fold_partner;
    a.m = c->m; a.n = c->n; a.E = (int)c->E; a.leaves = d->leaves; a.levels = d->levels; a.nodes = d->nodes; a.val_doubles = val_doubles;
    a.B = B; a.mu = mu; a.thresh = (eps * eps) * (double)c->E;
    a.max_iter = max_iter; a.cap = max_iter > 0 ? max_iter : 100000;
    a.ticket = d->d_ticket;
    LDPC_HIP_TRY(hipMemsetAsync(d->d_ticket, 0, 8, st));
    hipDeviceProp_t prop;
    LDPC_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    const int per_cu = std::max(1, (int)((size_t)160 * 1024 / lds_bytes));
    const int64_t want = (int64_t)prop.multiProcessorCount * per_cu;
    const int grid = (int)(B < want ? B : want);
    int rc = LDPC_OK;
    if (nw == 4) rc = admm_launch_lds<6, 3, 2, 4, 1>(a, lds_bytes, grid, st);
    else if (cpl == 1) rc = admm_launch_lds<6, 3, 2, 8, 1>(a, lds_bytes, grid, st);
    else rc = admm_launch_lds<6, 3, 3, 8, 2>(a, lds_bytes, grid, st);
    if (rc) return rc;
    LDPC_HIP_TRY(hipGetLastError());
    d->last_repacks = 0;
    d->last_backend = 1;
    return LDPC_OK;
}

int admm_decode(AdmmDecoder* d, const double* gamma, int64_t B, double mu, double eps, int32_t max_iter, double* x_out, int32_t* iters,
                uint8_t* converged, hipStream_t st) {
    if (B <= 0) return LDPC_OK;
    const Code* c = d->code;
    const int n = c->n, m = c->m;
    const int64_t E = c->E;
    if (B > (int64_t)65535 * 64) {
        set_error("ADMM: at most %d frames per call", 65535 * 64);
        return LDPC_E_ARG;
    }
    if (!(mu > 0.0)) {
        set_error("ADMM: mu must be positive");
        return LDPC_E_ARG;
    }
    LDPC_HIP_TRY(hipSetDevice(c->device));
    {
        const int rc = admm_decode_lds(d, gamma, B, mu, eps, max_iter, x_out, iters, converged, st);
        if (rc != 1) return rc;  // 1: not eligible for the LDS-resident kernel -> the streaming kernels below
    }
    d->last_backend = 0;
    const int tiles = (int)((B + 63) / 64);
    const size_t es = (size_t)tiles * E * 64 * sizeof(double), vs = (size_t)tiles * n * 64 * sizeof(double);
    LDPC_TRY(d->z.reserve(es));
    LDPC_TRY(d->lam.reserve(es));
    LDPC_TRY(d->d1.reserve(es));
    LDPC_TRY(d->d2.reserve(es));
    LDPC_TRY(d->x.reserve(vs));
    LDPC_TRY(d->gam.reserve(vs));
    LDPC_TRY(d->live.reserve((size_t)tiles * 8));
    LDPC_TRY(d->flags.reserve(64));
    LDPC_TRY(d->part.reserve((size_t)tiles * d->leaves * 128 * sizeof(double)));
    double *z = (double*)d->z.p, *lam = (double*)d->lam.p, *d1 = (double*)d->d1.p, *d2 = (double*)d->d2.p, *x = (double*)d->x.p,
           *gam = (double*)d->gam.p;
    u64* live = (u64*)d->live.p;
    int* live_tiles = (int*)d->flags.p;
    int* h_poll = (int*)d->pinned;
    LDPC_HIP_TRY(hipMemsetAsync(iters, 0, (size_t)B * sizeof(int32_t), st));
    if (converged) LDPC_HIP_TRY(hipMemsetAsync(converged, 0, (size_t)B, st));
    const unsigned gx = 256;
    hipLaunchKernelGGL(k_admm_init, dim3(gx, tiles), dim3(256), 0, st, gamma, B, n, E, gam, z, lam, x);
    hipLaunchKernelGGL(k_admm_live, dim3((tiles + 255) / 256), dim3(256), 0, st, live, B, tiles);
    const double thresh = (eps * eps) * (double)E;  // (eps ** 2) * parity_mtx.sum()   (src/admm.py:14)
    const int cap = max_iter > 0 ? max_iter : 100000;  // max_iter <= 0: no cap upstream (src/admm.py:51); bounded here
    const unsigned gv = (unsigned)((n + 3) / 4 < 512 ? (n + 3) / 4 : 512), gc = (unsigned)((m + 3) / 4 < 512 ? (m + 3) / 4 : 512);
    // frame repack (ldpc_repack.hpp): frames leave one by one (src/admm.py:65-66) while the frames that never converge run into the
    // iteration cap -- at 2.2 dB a frame needs 64 iterations on average, 4.5 % of them all 300, and nearly every tile holds one
    bool repack_ok = true;
    double repack_fill = 0.75;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK")) repack_ok = atoi(e) != 0;
    if (const char* e = std::getenv("LDPC_STREAM_REPACK_FILL")) repack_fill = atof(e);
    DevBuf* set_z[2] = {&d->z, &d->z2};
    DevBuf* set_l[2] = {&d->lam, &d->lam2};
    DevBuf* set_x[2] = {&d->x, &d->x2};
    DevBuf* set_g[2] = {&d->gam, &d->gam2};
    DevBuf* set_live[2] = {&d->live, &d->live2};
    DevBuf* set_fmap[2] = {&d->fmap, &d->fmap2};
    int cur = 0, cur_tiles = tiles, repacks = 0;
    int32_t* fmap = nullptr;
    int done = 0;
    for (int it = 0; it < cap; ++it) {
        hipLaunchKernelGGL(k_admm_x, dim3(gv, cur_tiles), dim3(256), 0, st, c->d_col_ptr, c->d_col_edge, z, lam, gam, x, live, n, E, cur_tiles, mu);
        const int Lfix = (c->min_dc == c->max_dc && c->max_dc >= 2 && c->max_dc <= 8) ? c->max_dc : 0;
        if (Lfix) {  // every check has the same degree: the register-only projection
#define LDPC_ADMM_FIXED_CASE(LL) \
    case LL: hipLaunchKernelGGL((k_admm_z_fixed<LL>), dim3(gc, cur_tiles), dim3(256), 0, st, c->d_edge_var, z, lam, x, d1, d2, live, m, n, E, mu); break;
            switch (Lfix) {
                LDPC_ADMM_FIXED_CASE(2) LDPC_ADMM_FIXED_CASE(3) LDPC_ADMM_FIXED_CASE(4) LDPC_ADMM_FIXED_CASE(5)
                LDPC_ADMM_FIXED_CASE(6) LDPC_ADMM_FIXED_CASE(7) LDPC_ADMM_FIXED_CASE(8)
            }
#undef LDPC_ADMM_FIXED_CASE
        } else if (c->max_dc <= 8) {  // unequal check degrees: the projection's work arrays in the LDS (data-dependent indices)
            const unsigned gcl = (unsigned)((m + 1) / 2 < 1024 ? (m + 1) / 2 : 1024);
            hipLaunchKernelGGL((k_admm_z<8, true>), dim3(gcl, cur_tiles), dim3(128), 0, st, c->d_row_ptr, c->d_edge_var, z, lam, x, d1, d2, live, m, n, E, mu);
        } else {
            hipLaunchKernelGGL((k_admm_z<16, false>), dim3(gc, cur_tiles), dim3(256), 0, st, c->d_row_ptr, c->d_edge_var, z, lam, x, d1, d2, live, m, n, E, mu);
        }
        const bool poll = (it % 8) == 7 || it + 1 == cap;
        if (poll) LDPC_HIP_TRY(hipMemsetAsync(live_tiles, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(k_admm_leaves, dim3((d->leaves + 3) / 4, cur_tiles), dim3(256), 0, st, d1, d2, d->d_leaf_off, d->d_leaf_len, d->leaves,
                           (double*)d->part.p, live, E);
        hipLaunchKernelGGL(k_admm_test, dim3(cur_tiles), dim3(64), 0, st, (const double*)d->part.p, d->d_prog, d->prog_len, d->leaves, live, iters,
                           converged, poll ? live_tiles : nullptr, B, thresh, it, it + 1 == cap ? it + 1 : max_iter, fmap);
        done = it + 1;
        if (poll) {
            LDPC_HIP_TRY(hipMemcpyAsync(h_poll, live_tiles, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
            LDPC_HIP_TRY(hipStreamSynchronize(st));
            const int lt = h_poll[0], lf = h_poll[1];
            if (lt == 0) break;
            if (repack_ok && lt >= 2 && (double)lf <= repack_fill * 64.0 * lt && it + 1 < cap) {
                const int nt = (lf + 63) / 64, nx = 1 - cur;
                const size_t es2 = (size_t)nt * E * 64 * sizeof(double), vs2 = (size_t)nt * n * 64 * sizeof(double);
                LDPC_TRY(set_z[nx]->reserve(es2));
                LDPC_TRY(set_l[nx]->reserve(es2));
                LDPC_TRY(set_x[nx]->reserve(vs2));
                LDPC_TRY(set_g[nx]->reserve(vs2));
                LDPC_TRY(set_live[nx]->reserve((size_t)nt * 8));
                LDPC_TRY(set_fmap[nx]->reserve((size_t)nt * 64 * sizeof(int32_t)));
                LDPC_TRY(d->rbase.reserve(((size_t)cur_tiles + 1) * sizeof(int32_t)));
                // x_hat of every frame of the old tiles as it stands (frames that left keep it; the moved ones overwrite theirs at the end)
                hipLaunchKernelGGL(k_admm_out, dim3((n + 3) / 4, cur_tiles), dim3(256), 0, st, x, x_out, B, n, fmap);
                hipLaunchKernelGGL(k_repack_plan, dim3(1), dim3(1024), 0, st, live, cur_tiles, (int32_t*)d->rbase.p);
                const int rows_per_wave = 128;
                const int chunks = (int)((E + n + rows_per_wave - 1) / rows_per_wave);
                hipLaunchKernelGGL(k_admm_repack, dim3((chunks + 3) / 4, nt), dim3(64, 4), 0, st, z, (double*)set_z[nx]->p, lam, (double*)set_l[nx]->p, x,
                                   (double*)set_x[nx]->p, gam, (double*)set_g[nx]->p, live, (u64*)set_live[nx]->p, (const int32_t*)d->rbase.p, fmap,
                                   (int32_t*)set_fmap[nx]->p, cur_tiles, n, E, rows_per_wave);
                cur = nx;
                z = (double*)set_z[cur]->p;
                lam = (double*)set_l[cur]->p;
                x = (double*)set_x[cur]->p;
                gam = (double*)set_g[cur]->p;
                live = (u64*)set_live[cur]->p;
                fmap = (int32_t*)set_fmap[cur]->p;
                cur_tiles = nt;
                ++repacks;
            }
        }
    }
    hipLaunchKernelGGL(k_admm_out, dim3((n + 3) / 4, cur_tiles), dim3(256), 0, st, x, x_out, B, n, fmap);
    d->last_repacks = repacks;
    LDPC_HIP_TRY(hipGetLastError());
    d->last_iters = done;
    return LDPC_OK;
}

}  // namespace ldpc
