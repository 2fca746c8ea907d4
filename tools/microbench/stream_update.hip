#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
// k_cn-like: each wave owns a contiguous run; K lines in flight, then stored back in place (or to dst)
template <typename V, int K, bool NTS>
__global__ __launch_bounds__(256) void k(const V* __restrict__ src, V* __restrict__ dst, long nvec, int lines_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long base = wave * lines_per_wave * 64;
    for (int l = 0; l < lines_per_wave; l += K) {
        V v[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { long i = base + (long)(l + j) * 64 + lane; if (i < nvec) v[j] = src[i]; }
#pragma unroll
        for (int j = 0; j < K; ++j) { long i = base + (long)(l + j) * 64 + lane; if (i < nvec) { if (NTS) __builtin_nontemporal_store(v[j], dst + i); else dst[i] = v[j]; } }
    }
}
template <typename V, int K, bool NTS>
void run(const char* name, void* a, void* b, long bytes, int lpw) {
    long nvec = bytes / sizeof(V);
    long waves = (nvec / 64 + lpw - 1) / lpw;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<V, K, NTS>), dim3((waves + 3) / 4), dim3(256), 0, 0, (const V*)a, (V*)b, nvec, lpw);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s lpw=%3d  %.3f ms/launch  %.2f TB/s\n", name, lpw, ms / 5, 2.0 * bytes / (ms / 5 * 1e-3) / 1e12);
}
int main() {
    const long bytes = 1L << 31;  // 2 GiB each way (>> MALL)
    void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int lpw : {24, 96}) {
        run<float, 24, false>("dword  K=24 plain", a, b, bytes, lpw);
        run<float, 24, true>("dword  K=24 nt-store", a, b, bytes, lpw);
        run<float, 24, true>("dword  K=24 nt-store inplace", a, a, bytes, lpw);
        run<f4, 6, false>("dwordx4 K=6 plain", a, b, bytes, lpw);
        run<f4, 6, true>("dwordx4 K=6 nt-store", a, b, bytes, lpw);
        run<f4, 6, true>("dwordx4 K=6 nt-store inplace", a, a, bytes, lpw);
        run<f4, 12, true>("dwordx4 K=12 nt-store inplace", a, a, bytes, lpw);
    }
    return 0;
}
