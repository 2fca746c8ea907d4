// Which LDS instructions does SQ_LDS_BANK_CONFLICT charge although their addresses are conflict-free by the bank model of
// MI355X_MICROARCH.md?  One kernel per access form, each issuing REPS wave-instructions per wave; run under
//   rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace ...
// and divide by the instruction count.   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_conflict_probe.hip -o build/ab/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int REPS = 4096;
#define KERNEL(name, body)                                                                   \
    __global__ __launch_bounds__(128) void name(float* out, const int* perm) {                \
        extern __shared__ __attribute__((aligned(256))) unsigned char smem[];                 \
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;                              \
        float* base = reinterpret_cast<float*>(smem) + w * 4096;                              \
        for (int i = lane; i < 4096; i += 64) base[i] = (float)i;                             \
        __syncthreads();                                                                      \
        const int p = perm[lane];                                                             \
        float acc = 0.f;                                                                      \
        body;                                                                                 \
        out[blockIdx.x * 128 + threadIdx.x] = acc;                                            \
    }
// (a) lane-contiguous ds_read_b32
KERNEL(k_read_row, for (int r = 0; r < REPS; ++r) { acc += *(volatile float*)(base + (r & 31) * 64 + lane); })
// (b) ds_read_b32, lanes permuted inside each half-wave over distinct banks (conflict-free gather)
KERNEL(k_read_perm, for (int r = 0; r < REPS; ++r) { acc += *(volatile float*)(base + (r & 31) * 64 + p); })
// (c) ds_read_b32, every lane a different ROW, bank == lane % 32 (conflict-free, addresses far apart)
KERNEL(k_read_far, for (int r = 0; r < REPS; ++r) { acc += *(volatile float*)(base + ((lane * 7 + r) & 63) * 64 + (lane & 31) + (lane & 32)); })
// (d) ds_write_b32 lane-contiguous
KERNEL(k_write_row, for (int r = 0; r < REPS; ++r) { *(volatile float*)(base + (r & 31) * 64 + lane) = (float)r; })
// (e) ds_write_addtid_b32 rows
KERNEL(k_write_addtid, {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base);
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0v) : "memory");
    for (int r = 0; r < REPS; r += 4) {
        asm volatile("ds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %0 offset:256\n\tds_write_addtid_b32 %0 offset:512\n\tds_write_addtid_b32 %0 offset:768" ::"v"((float)r) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
})
// (f) two lanes per half-wave share an ADDRESS (broadcast), the rest distinct banks
KERNEL(k_read_bcast, for (int r = 0; r < REPS; ++r) { acc += *(volatile float*)(base + (r & 31) * 64 + ((lane & 31) == 5 ? (lane & 32) + 4 : p)); })
// (g) ds_read_b64 lane-contiguous / (h) ds_write_b64 lane-contiguous
KERNEL(k_read64_row, for (int r = 0; r < REPS; ++r) { acc += (float)*(volatile double*)(reinterpret_cast<double*>(base) + (r & 15) * 64 + lane); })
KERNEL(k_write64_row, for (int r = 0; r < REPS; ++r) { *(volatile double*)(reinterpret_cast<double*>(base) + (r & 15) * 64 + lane) = (double)r; })

int main() {
    float* out;
    int* perm;
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = (l & 32) + ((l * 13 + 5) & 31);  // a permutation inside each half-wave: distinct banks
    hipMalloc(&out, 256 * 128 * 4);
    hipMalloc(&perm, 256);
    hipMemcpy(perm, h, 256, hipMemcpyHostToDevice);
#define RUN(k) hipLaunchKernelGGL(k, dim3(256), dim3(128), 32768, 0, out, perm); hipDeviceSynchronize();
    RUN(k_read_row) RUN(k_read_perm) RUN(k_read_far) RUN(k_write_row) RUN(k_write_addtid) RUN(k_read_bcast) RUN(k_read64_row) RUN(k_write64_row)
    printf("done: %d wave-instructions per wave, 512 waves per kernel\n", REPS);
    return 0;
}
