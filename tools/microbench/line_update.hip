#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
typedef float f4 __attribute__((ext_vector_type(4)));
// VN-like: each wave updates K lines at random positions (a permutation of all lines) in place, plus streams a prior line
template <typename V, int K>
__global__ __launch_bounds__(256) void k(V* __restrict__ msg, const int* __restrict__ perm, long nlines, int lines_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = wave * lines_per_wave;
    for (int l = 0; l < lines_per_wave; l += K) {
        V v[K]; long idx[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { long i = base + l + j; idx[j] = i < nlines ? (long)perm[i] : -1; }
#pragma unroll
        for (int j = 0; j < K; ++j) if (idx[j] >= 0) v[j] = msg[idx[j] * 64 + lane];
#pragma unroll
        for (int j = 0; j < K; ++j) if (idx[j] >= 0) msg[idx[j] * 64 + lane] = v[j] + (V)(1.0f);
    }
}
template <typename V, int K>
void run(const char* name, void* a, long bytes, int lpw, long region_lines) {
    long nlines = bytes / (sizeof(V) * 64);
    // permutation that is random inside regions of `region_lines` lines (a tile's message array), regions in order
    std::vector<int> perm(nlines); for (long i = 0; i < nlines; ++i) perm[i] = (int)i;
    std::mt19937 rng(1);
    for (long r = 0; r < nlines; r += region_lines) std::shuffle(perm.begin() + r, perm.begin() + std::min(nlines, r + region_lines), rng);
    int* dperm; hipMalloc(&dperm, nlines * 4); hipMemcpy(dperm, perm.data(), nlines * 4, hipMemcpyHostToDevice);
    long waves = (nlines + lpw - 1) / lpw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<V, K>), dim3((waves + 3) / 4), dim3(256), 0, 0, (V*)a, dperm, nlines, lpw);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-22s lpw=%3d region=%7ld lines  %.3f ms  %.2f TB/s\n", name, lpw, region_lines, ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
    hipFree(dperm);
}
int main() {
    const long bytes = 1L << 31;
    void* a; hipMalloc(&a, bytes); hipMemset(a, 0, bytes);
    // n=1200 code: E=3600 lines per tile (64 frames) -> region 3600 ; with 256-frame tiles: 3600 lines of 1 KB
    run<float, 12>("256B lines K=12", a, bytes, 48, 3600);
    run<f4, 3>("1KB lines  K=3", a, bytes, 12, 3600);
    run<f4, 6>("1KB lines  K=6", a, bytes, 12, 3600);
    run<float, 12>("256B lines K=12", a, bytes, 48, 194400);
    run<f4, 3>("1KB lines  K=3", a, bytes, 12, 194400);
    run<f4, 6>("1KB lines  K=6", a, bytes, 12, 194400);
    run<float, 12>("256B lines K=12", a, bytes, 48, 1L << 40);
    run<f4, 6>("1KB lines  K=6", a, bytes, 12, 1L << 40);
    return 0;
}
