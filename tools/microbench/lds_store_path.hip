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
    typedef double d2 __attribute__((ext_vector_type(2)));
    const uint32_t a_lin2 = base + lane * 16;
    const uint32_t a_p64 = base + ((lane * 37 + 11) & 63) * 4;
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
    double v = (double)lane, acc = 0.0, v2 = v + 1.0;
    d2 q = {v, v2};
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
            // two rows per instruction (round 4, second half): counted per INSTRUCTION, i.e. per two rows of 64 doubles
            if (KIND == 5 && (r & 1) == 0) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a_perm), "v"(v), "v"(v2), "n"(r), "n"(r + 1) : "memory");
            if (KIND == 6 && (r & 1) == 0) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a_lin2), "v"(v), "v"(v2), "n"((r & 2) * 64), "n"((r & 2) * 64 + 1) : "memory");
            if (KIND == 7 && (r & 1) == 0) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a_lin2), "v"(q), "n"(r * 512) : "memory");
            if (KIND == 8 && (r & 1) == 0) { d2 t; asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(a_perm), "n"(r), "n"(r + 1) : "memory"); acc += t.x + t.y; }
            if (KIND == 9 && (r & 1) == 0) { d2 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(a_lin2), "n"(r * 512) : "memory"); acc += t.x + t.y; }
            // split planes (lo / hi dwords of a double in two 256-byte rows): one ds_read2_b32 gathers a double, two ds_write_addtid_b32 store a row of
            // doubles without an address register (the fp32 kernel's store).  a_p64: a permutation of all 64 lanes (64 distinct banks)
            if (KIND == 11) { double t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(a_p64), "n"((r & 1) * 128), "n"((r & 1) * 128 + 64) : "memory"); acc += t; }
            if (KIND == 12) { double t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(a_p64), "n"((r & 1) * 128), "n"((r & 1) * 128 + 65) : "memory"); acc += t; }
            if (KIND == 13) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"((float)v), "n"(r * 256) : "memory");
            if (KIND == 14) {  // the would-be kernel's mix per 7 slots: 3 gathers of doubles + 4 half-row stores
                if ((r % 7) < 3) { double t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(a_p64), "n"(0), "n"(64) : "memory"); acc += t; }
                else if (r < 14) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"((float)v), "n"(r * 256) : "memory");
            }
            if (KIND == 15) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(a_p64), "n"(r * 256) : "memory"); acc += t; }
            // the kernel's mix: are loads and stores additive on the path?  per group of 4 slots: 2 gathers + 1 row store (3 instructions)
            if (KIND == 10 && (r & 3) != 3) {
                if ((r & 3) == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lin), "v"(v), "n"(r * 512) : "memory");
                else { double t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(a_perm), "n"(r * 512) : "memory"); acc += t; }
            }
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
    const char* names[] = {"ds_write_b64 lane-contiguous", "ds_write_b64 permuted", "ds_add_f64 permuted", "ds_read_b64 permuted", "ds_write_b32 permuted",
                           "ds_write2st64_b64 (2 rows)", "ds_write2_b64 (16 B per lane)", "ds_write_b128 (16 B per lane)", "ds_read2st64_b64 (2 rows)", "ds_read_b128 (16 B per lane)",
                           "mix: 2 ds_read_b64 + 1 ds_write_b64", "ds_read2_b32 rows 256 B apart (same bank)", "ds_read2_b32 rows 260 B apart", "ds_write_addtid_b32",
                           "mix: 3 ds_read2_b32 + 4 ds_write_addtid_b32", "ds_read_b32 permuted (64 banks)"};
    const double per16[] = {16, 16, 16, 16, 16, 8, 8, 8, 8, 8, 12, 16, 16, 16, 14, 16};  // instructions issued per 16 slots
    double t[16] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters),
                    run<6>(out, iters), run<7>(out, iters), run<8>(out, iters), run<9>(out, iters), run<10>(out, iters), run<11>(out, iters),
                    run<12>(out, iters), run<13>(out, iters), run<14>(out, iters), run<15>(out, iters)};
    for (int i = 0; i < 16; ++i) {
        const double ti = t[i] * 16.0 / per16[i];
        printf("%-36s %.2f ns per wave-instruction per CU = %.2f cycles at 2.4 GHz\n", names[i], ti * 1e9, ti * 2.4e9);
    }
    return 0;
}
