// Counter-based RNG for the on-device channel kernels: Philox4x32-10
// (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// Keyed by (seed); counter = (block j within the frame, stream id, global frame index lo, hi), so the
// noise of a frame depends only on (seed, stream, frame index): results are identical for any sharding
// of the frame range over GPUs.  oracle/bp_oracle.py::philox4x32 is the bit-exact CPU statement.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {

struct Philox4 {
    uint32_t w[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                          uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// words 4j .. 4j+3 of frame `frame` in stream `stream`
__host__ __device__ __forceinline__ Philox4 philox_word_block(uint64_t seed, uint32_t stream, uint64_t frame, uint32_t j) {
    return philox4x32_10(j, stream, (uint32_t)frame, (uint32_t)(frame >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Box-Muller on a pair of Philox words: (w_a, w_b) -> two independent N(0,1).  Shared by the stand-alone channel kernel and
// the fused simulate kernel so that both produce bit-identical noise.
template <typename T>
__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, T& z0, T& z1);

template <>
__device__ __forceinline__ void box_muller<float>(uint32_t wa, uint32_t wb, float& z0, float& z1) {
    // u = (w + 0.5) / 2^32 in (0,1); radius from the full 32 bits (tail to 6.7 sigma), angle from 32 bits.
    // Everything on the transcendental unit: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
    // REVOLUTIONS, so sin(2 pi u2) needs no range reduction).  ~1e-6 absolute accuracy, checked against an fp64 model.
    const float u1 = fmaf((float)wa, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float u2 = fmaf((float)wb, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1), ln = ln2 * log2
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

template <>
__device__ __forceinline__ void box_muller<double>(uint32_t wa, uint32_t wb, double& z0, double& z1) {
    const double u1 = ((double)wa + 0.5) * 2.3283064365386963e-10;
    const double u2 = ((double)wb + 0.5) * 2.3283064365386963e-10;
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincospi(2.0 * u2, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

}  // namespace ldpc
