// How fast can a pass read lines of 256 / 512 / 1024 bytes at RANDOM places (each line exactly once) and stream the same amount out?
// -- the access pattern of the fp16-storage check pass (gather the v2c lines of a check, stream its c2v lines).  One wave moves 6 lines per
// step (a degree-6 check); the permutation is a fixed odd multiplier modulo the (power-of-two) line count.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/gather_lines tools/microbench/gather_lines.hip && /tmp/gather_lines
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__global__ __launch_bounds__(256) void k(const T* __restrict__ src, T* __restrict__ dst, uint32_t nlines_mask, int64_t nlines) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t base = wave * 6; base + 6 <= nlines; base += waves * 6) {
        T v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint32_t line = (uint32_t)(((uint64_t)(base + j) * 2654435761ull) & nlines_mask);
            v[j] = __builtin_nontemporal_load(src + (int64_t)line * 64 + lane);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) __builtin_nontemporal_store(v[j], dst + (base + j) * 64 + lane);
    }
}
template <typename T>
void run(const char* name, void* a, void* b, size_t bytes) {
    const int64_t nlines = (int64_t)(bytes / (64 * sizeof(T)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T><<<256 * 16, 256>>>((const T*)a, (T*)b, (uint32_t)(nlines - 1), nlines);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<T><<<256 * 16, 256>>>((const T*)a, (T*)b, (uint32_t)(nlines - 1), nlines);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %6.2f TB/s (read + write, %zu MiB each way)\n", name, 2.0 * 5 * bytes / (ms * 1e-3) / 1e12, bytes >> 20);
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    run<uint32_t>("256-B lines (4 B per lane)", a, b, bytes);
    run<u32x2>("512-B lines (8 B per lane)", a, b, bytes);
    run<u32x4>("1024-B lines (16 B per lane)", a, b, bytes);
    return 0;
}
