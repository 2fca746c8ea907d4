// Issue cost of the integer VALU instructions the Philox rounds and the bit-sliced erasure rules are made of, relative to v_xor_b32:
// every wave runs a dependent-free stream of N instructions of one kind (8 independent accumulators), 16 waves per CU on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    uint64_t d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; d[i] = a[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if (KIND == 4) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 5) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                if (KIND == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if (KIND == 7) asm volatile("v_or_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
                if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : );
            }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i] ^ (uint32_t)d[i] ^ (uint32_t)(d[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
double run(uint32_t* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256 * 4, 256>>>(out, 10, 1);
    hipEventRecord(e0);
    k<KIND><<<256 * 4, 256>>>(out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD: 4 blocks/CU x 4 waves / 4 SIMDs = 4 waves per SIMD, each iters*32 instructions
    return ms * 1e-3 / ((double)iters * 32 * 4);  // seconds per wave-instruction per SIMD
}
int main() {
    uint32_t* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    const int iters = 20000;
    const char* names[] = {"v_xor_b32", "v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_bitop3_b32", "v_bfi_b32", "v_mul_u32_u24", "v_or_b32_dpp", "v_cndmask_b32"};
    double t[9] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters), run<6>(out, iters), run<7>(out, iters), run<8>(out, iters)};
    for (int i = 0; i < 9; ++i) printf("%-16s %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz (x%.2f of v_xor)\n", names[i], t[i] * 1e9, t[i] * 2.4e9, t[i] / t[0]);
    return 0;
}
