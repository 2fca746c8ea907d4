// HBM access-pattern ceilings for the streaming BP passes (round 3): what a read+write pass over a message array can reach on
// MI355X, by pattern.  Stand-alone: hipcc -O3 --offload-arch=gfx950 tools/microbench/hbm_patterns.hip -o /tmp/hbm_patterns
//
//   copy          a -> b, grid-stride, 16 B per lane                 (the 6.3 TB/s copy ceiling of MI355X_MICROARCH.md)
//   rmw-stride    a -> a in place, grid-stride
//   rmw-chunk     a -> a in place, every wave owns a contiguous run (the check pass), K vectors in flight
//   line-inplace  every wave updates lines at random positions in place (the variable pass), line = 256 B ... 4 KiB
//   gather-write  random-line reads from a, contiguous writes to b   (c2v gathered in CSC order, v2c written as a stream)
//   read-scatter  contiguous reads from a, random-line writes to b
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_stride(const f4* __restrict__ a, f4* __restrict__ b, long nvec) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride * 4) {
        f4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i + j * stride < nvec) v[j] = a[i + j * stride];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i + j * stride < nvec) st<NT>(b + i + j * stride, v[j] + 1.0f);
    }
}

// one workgroup sweeps a contiguous window; its 4 waves interleave line by line (window = 4 * K lines per step)
template <int K, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const f4* __restrict__ a, f4* __restrict__ b, long nvec, int lines_per_wave, int interleave) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 4 + w;
    for (int l = 0; l < lines_per_wave; l += K) {
        f4 v[K];
        long idx[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long line = interleave ? ((long)blockIdx.x * 4 * lines_per_wave + (long)(l + j) * 4 + w) : (wave * lines_per_wave + l + j);
            idx[j] = line * 64 + lane;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) if (idx[j] < nvec) v[j] = a[idx[j]];
#pragma unroll
        for (int j = 0; j < K; ++j) if (idx[j] < nvec) st<NT>(b + idx[j], v[j] + 1.0f);
    }
}

// random lines of LV consecutive 1-KiB vectors-lines (LV = 1: 1 KiB, 2: 2 KiB, 4: 4 KiB); MODE 0 in place, 1 gather a -> stream b, 2 stream a -> scatter b
template <int LV, int K, int MODE, bool NT>
__global__ __launch_bounds__(256) void k_lines(const f4* __restrict__ a, f4* __restrict__ b, const int* __restrict__ perm, long nlines, int lines_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = wave * lines_per_wave;
    for (int l = 0; l < lines_per_wave; l += K) {
        f4 v[K][LV];
        long rd[K], wr[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long i = base + l + j;
            const long p = i < nlines ? (long)perm[i] : -1;
            rd[j] = MODE == 2 ? (i < nlines ? i : -1) : p;
            wr[j] = MODE == 1 ? (i < nlines ? i : -1) : p;
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int q = 0; q < LV; ++q) if (rd[j] >= 0) v[j][q] = a[(rd[j] * LV + q) * 64 + lane];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int q = 0; q < LV; ++q) if (wr[j] >= 0) st<NT>(b + (wr[j] * LV + q) * 64 + lane, v[j][q] + 1.0f);
    }
}

// 256-byte lines (4 B per lane): today's tile
template <int K, int MODE>
__global__ __launch_bounds__(256) void k_lines4(const float* __restrict__ a, float* __restrict__ b, const int* __restrict__ perm, long nlines, int lines_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = wave * lines_per_wave;
    for (int l = 0; l < lines_per_wave; l += K) {
        float v[K];
        long rd[K], wr[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const long i = base + l + j;
            const long p = i < nlines ? (long)perm[i] : -1;
            rd[j] = MODE == 2 ? (i < nlines ? i : -1) : p;
            wr[j] = MODE == 1 ? (i < nlines ? i : -1) : p;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) if (rd[j] >= 0) v[j] = a[rd[j] * 64 + lane];
#pragma unroll
        for (int j = 0; j < K; ++j) if (wr[j] >= 0) b[wr[j] * 64 + lane] = v[j] + 1.0f;
    }
}

static hipEvent_t e0, e1;
template <typename F> static double timed(F launch) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms / 5;
}
static void report(const char* name, double ms, long bytes) { printf("%-58s %8.3f ms  %.2f TB/s\n", name, ms, 2.0 * bytes / (ms * 1e-3) / 1e12); fflush(stdout); }

static int* make_perm(long nlines, long region) {
    std::vector<int> perm(nlines);
    for (long i = 0; i < nlines; ++i) perm[i] = (int)i;
    std::mt19937 rng(1);
    for (long r = 0; r < nlines; r += region) std::shuffle(perm.begin() + r, perm.begin() + std::min(nlines, r + region), rng);
    int* d;
    hipMalloc(&d, nlines * 4);
    hipMemcpy(d, perm.data(), nlines * 4, hipMemcpyHostToDevice);
    return d;
}

template <int LV, int K, int MODE, bool NT>
static void run_lines(const char* name, f4* a, f4* b, long bytes, long region_bytes) {
    const long nlines = bytes / (1024L * LV);
    int* perm = make_perm(nlines, std::max(1L, region_bytes / (1024L * LV)));
    const int lpw = 12;
    const long waves = (nlines + lpw - 1) / lpw;
    char buf[128];
    snprintf(buf, sizeof buf, "%s line=%d B K=%d region=%ld MB%s", name, 1024 * LV, K, region_bytes >> 20, NT ? " nt-store" : "");
    report(buf, timed([&] { hipLaunchKernelGGL((k_lines<LV, K, MODE, NT>), dim3((waves + 3) / 4), dim3(256), 0, 0, a, MODE == 0 ? a : b, perm, nlines, lpw); }), bytes);
    hipFree(perm);
}
template <int K, int MODE>
static void run_lines4(const char* name, f4* a, f4* b, long bytes, long region_bytes) {
    const long nlines = bytes / 256;
    int* perm = make_perm(nlines, std::max(1L, region_bytes / 256));
    const int lpw = 48;
    const long waves = (nlines + lpw - 1) / lpw;
    char buf[128];
    snprintf(buf, sizeof buf, "%s line=256 B K=%d region=%ld MB", name, K, region_bytes >> 20);
    report(buf, timed([&] { hipLaunchKernelGGL((k_lines4<K, MODE>), dim3((waves + 3) / 4), dim3(256), 0, 0, (const float*)a, (float*)(MODE == 0 ? a : b), perm, nlines, lpw); }), bytes);
    hipFree(perm);
}

int main() {
    const long bytes = 1L << 31;  // 2 GiB each way (>> the 256 MiB Infinity Cache)
    f4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long nvec = bytes / 16;
    for (int grid : {2048, 8192, 32768}) {
        char buf[96];
        snprintf(buf, sizeof buf, "copy a->b grid-stride grid=%d", grid);
        report(buf, timed([&] { hipLaunchKernelGGL((k_stride<false>), dim3(grid), dim3(256), 0, 0, a, b, nvec); }), bytes);
        snprintf(buf, sizeof buf, "copy a->b grid-stride nt-store grid=%d", grid);
        report(buf, timed([&] { hipLaunchKernelGGL((k_stride<true>), dim3(grid), dim3(256), 0, 0, a, b, nvec); }), bytes);
        snprintf(buf, sizeof buf, "rmw in place grid-stride grid=%d", grid);
        report(buf, timed([&] { hipLaunchKernelGGL((k_stride<false>), dim3(grid), dim3(256), 0, 0, a, a, nvec); }), bytes);
        snprintf(buf, sizeof buf, "rmw in place grid-stride nt-store grid=%d", grid);
        report(buf, timed([&] { hipLaunchKernelGGL((k_stride<true>), dim3(grid), dim3(256), 0, 0, a, a, nvec); }), bytes);
    }
    for (int lpw : {24, 96}) {
        const long waves = (nvec / 64 + lpw - 1) / lpw;
        for (int il : {0, 1}) {
            char buf[96];
            snprintf(buf, sizeof buf, "rmw in place wave-chunk K=6 lpw=%d %s", lpw, il ? "wg-interleaved" : "per-wave runs");
            report(buf, timed([&] { hipLaunchKernelGGL((k_chunk<6, false>), dim3((waves + 3) / 4), dim3(256), 0, 0, a, a, nvec, lpw, il); }), bytes);
            snprintf(buf, sizeof buf, "rmw in place wave-chunk K=6 lpw=%d nt %s", lpw, il ? "wg-interleaved" : "per-wave runs");
            report(buf, timed([&] { hipLaunchKernelGGL((k_chunk<6, true>), dim3((waves + 3) / 4), dim3(256), 0, 0, a, a, nvec, lpw, il); }), bytes);
            snprintf(buf, sizeof buf, "rmw in place wave-chunk K=12 lpw=%d nt %s", lpw, il ? "wg-interleaved" : "per-wave runs");
            report(buf, timed([&] { hipLaunchKernelGGL((k_chunk<12, true>), dim3((waves + 3) / 4), dim3(256), 0, 0, a, a, nvec, lpw, il); }), bytes);
        }
    }
    // the message block of one tile of the n = 64 800 shape: 194 400 lines (50 MB at 256 B per line, 199 MB at 1 KiB)
    for (long region : {50L << 20, 200L << 20, 2048L << 20}) {
        run_lines4<12, 0>("line-inplace", a, b, bytes, region);
        run_lines<1, 3, 0, false>("line-inplace", a, b, bytes, region);
        run_lines<1, 6, 0, false>("line-inplace", a, b, bytes, region);
        run_lines<1, 6, 0, true>("line-inplace", a, b, bytes, region);
        run_lines<1, 12, 0, false>("line-inplace", a, b, bytes, region);
        run_lines<2, 3, 0, false>("line-inplace", a, b, bytes, region);
        run_lines<2, 6, 0, false>("line-inplace", a, b, bytes, region);
        run_lines<4, 3, 0, false>("line-inplace", a, b, bytes, region);
    }
    const long region = 200L << 20;
    run_lines4<12, 1>("gather-read a, stream-write b", a, b, bytes, region);
    run_lines4<12, 2>("stream-read a, scatter-write b", a, b, bytes, region);
    run_lines<1, 6, 1, false>("gather-read a, stream-write b", a, b, bytes, region);
    run_lines<1, 6, 1, true>("gather-read a, stream-write b", a, b, bytes, region);
    run_lines<1, 6, 2, false>("stream-read a, scatter-write b", a, b, bytes, region);
    run_lines<1, 6, 2, true>("stream-read a, scatter-write b", a, b, bytes, region);
    run_lines<2, 6, 1, true>("gather-read a, stream-write b", a, b, bytes, region);
    run_lines<2, 6, 2, true>("stream-read a, scatter-write b", a, b, bytes, region);
    return 0;
}
