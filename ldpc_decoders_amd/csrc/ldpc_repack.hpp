// Frame repack shared by the tile-based backends (streaming BP, ADMM): both keep 64 frames per tile (lane == frame) and let a
// frame leave on its own (src/bpa.py:28-29, src/admm.py:65-66); a tile keeps streaming its whole state for as long as one of its
// frames is live, so when a batch thins out the live frames are gathered into dense tiles.  k_repack_plan ranks the live frames
// (prefix sums of the tiles' live counts); repack_source() gives destination rank j its source (tile, lane).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {

static __global__ __launch_bounds__(1024) void k_repack_plan(const unsigned long long* __restrict__ live, int tiles, int32_t* __restrict__ base) {
    // base[t] = number of live frames in tiles [0, t); base[tiles] = total.  One workgroup, tiles <= 65535.
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (tiles + 1023) / 1024;
    int sum = 0;
    for (int i = t * per; i < min(tiles, (t + 1) * per); ++i) sum += __popcll(live[i]);
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int i = t * per; i < min(tiles, (t + 1) * per); ++i) {
        base[i] = run;
        run += __popcll(live[i]);
    }
    if (t == 1023) base[tiles] = part[1023];
}

__device__ __forceinline__ int nth_set_bit(unsigned long long x, int k) {  // position of the k-th (0-based) set bit of x
    for (int i = 0; i < k; ++i) x &= x - 1;
    return __ffsll((long long)x) - 1;
}

// source (tile, lane) of the live frame of rank j; false if j is beyond the live frames
__device__ __forceinline__ bool repack_source(const int32_t* __restrict__ base, const unsigned long long* __restrict__ live_src, int tiles_src,
                                              int j, int* st, int* sl) {
    *st = 0;
    *sl = 0;
    if (j >= base[tiles_src]) return false;
    int lo = 0, hi = tiles_src - 1;
    while (lo < hi) {  // the tile whose rank interval holds j
        const int mid = (lo + hi + 1) >> 1;
        if (base[mid] <= j) lo = mid; else hi = mid - 1;
    }
    *st = lo;
    *sl = nth_set_bit(live_src[lo], j - base[lo]);
    return true;
}

}  // namespace ldpc
