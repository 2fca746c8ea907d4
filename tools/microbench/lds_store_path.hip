// Cycles per wave-instruction of the LDS operations the fp64 min-sum kernel is built from -- and of the one a proposed variant would use
// (ordered ds_add_f64 accumulation of the check messages into the marginals, VERDICT r3 task 3): every wave issues a stream of ONE kind of
// DS instruction on conflict-free addresses, 16 waves per CU on every CU.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/lds_store_path tools/microbench/lds_store_path.hip && /tmp/lds_store_path
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    extern __shared__ unsigned char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)smem + w * 8192;
    // lane-contiguous address and a conflict-free "gathered" one (a permutation of the lanes inside each half-wave)
    const uint32_t a_lin = base + lane * 8, a_perm = base + ((lane & 32) | ((lane * 5 + 3) & 31)) * 8;
    double v = (double)lane, acc = 0.0;
    for (int i = lane; i < 1024; i += 64) reinterpret_cast<double*>(smem + w * 8192)[i] = 0.0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (KIND == 0) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lin), "v"(v), "n"(r * 512) : "memory");
            if (KIND == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_perm), "v"(v), "n"(r * 512) : "memory");
            if (KIND == 2) asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(a_perm), "v"(v), "n"(r * 512) : "memory");
            if (KIND == 3) { double t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(a_perm), "n"(r * 512) : "memory"); acc += t; }
            if (KIND == 4) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a_perm), "v"((float)v), "n"(r * 512) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + reinterpret_cast<double*>(smem + w * 8192)[lane];
}
template <int KIND>
double run(double* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    k<KIND><<<256 * 4, 256, 32768>>>(out, 10);
    hipEventRecord(e0);
    k<KIND><<<256 * 4, 256, 32768>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)iters * 16 * 16);  // seconds per wave-instruction per CU (16 waves per CU)
}
int main() {
    double* out; hipMalloc(&out, 256 * 4 * 256 * 8);
    const int iters = 20000;
    const char* names[] = {"ds_write_b64 lane-contiguous", "ds_write_b64 permuted", "ds_add_f64 permuted", "ds_read_b64 permuted", "ds_write_b32 permuted"};
    double t[5] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters)};
    for (int i = 0; i < 5; ++i) printf("%-30s %.2f ns per wave-instruction per CU = %.2f cycles at 2.4 GHz\n", names[i], t[i] * 1e9, t[i] * 2.4e9);
    return 0;
}
