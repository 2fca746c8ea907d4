// Bit-plane arithmetic of the bit-sliced erasure decoder (shared by the LDS-resident kernels, ldpc_bec_kernels.hpp, and the streaming
// kernels, ldpc_bec_stream.hip): a message of the ternary alphabet {-1, +1, 0} of src/bec.py:70-125 is two bits, k (known) and
// v (value; v implies k); a plane word holds one of them for 32 frames.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {
namespace {

constexpr int BEC_SLAB = 32;  // frames per slab == bits of a plane word
constexpr int BECS_FLUSH_MAX = 4096;  // Monte-Carlo kernel: frames a counter slot may accumulate before its packed 16-bit sums are flushed

struct P2 {
    uint32_t k, v;
};
// Any boolean function of three planes is ONE instruction (v_bitop3_b32); its 8-bit truth table is the function applied to the constants
// 0xF0, 0xCC, 0xAA (first, second, third operand).  B3(a, b, c, expression in X0, X1, X2) spells it where the compiler's own matching
// of and/or/not trees was measured to fall short (9 instructions for the 4 of an edge's message rebuild).
#define B3(a, b, c, EXPR) \
    __builtin_amdgcn_bitop3_b32((a), (b), (c), (unsigned)([] { constexpr unsigned X0 = 0xF0u, X1 = 0xCCu, X2 = 0xAAu; (void)X0; (void)X1; (void)X2; return (EXPR) & 0xFFu; }()))
__device__ __forceinline__ uint32_t plane_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
// v[LANE] = s (a wave-uniform value): one v_writelane_b32
template <int LANE>
__device__ __forceinline__ void write_lane(uint32_t& v, uint32_t s) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(s)), "n"(LANE));
}
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return B3(a, b, c, (X0 & X1) | (X2 & (X0 | X1))); }
__device__ __forceinline__ uint32_t mux(uint32_t s, uint32_t a, uint32_t b) { return B3(s, a, b, (X0 & X1) | (~X0 & X2)); }  // s ? a : b, bitwise

// bits needed for a count in [0, N]
template <int N>
struct BitsFor {
    static constexpr int value = (N < 2) ? 1 : (N < 4) ? 2 : (N < 8) ? 3 : (N < 16) ? 4 : (N < 32) ? 5 : (N < 64) ? 6 : (N < 128) ? 7 : 8;
    static_assert(N < 256, "plane counters of up to eight bits");
};
// S = number of set planes among in[0..N), as bit planes S[0] (weight 1) ... -- column compression: full adders (x^y^z, majority) take
// three planes of one weight to one plane of that weight and one of the next, half adders two.  Every index is a compile-time constant
// after unrolling (checked on the ISA: no scratch, 14 instructions for N = 8).
template <int N>
__device__ __forceinline__ void plane_count(const uint32_t (&in)[N], uint32_t (&S)[BitsFor<N>::value]) {
    constexpr int NB = BitsFor<N>::value;
    uint32_t col[NB][2 * N];
    int head[NB], tail[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) head[b] = tail[b] = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) col[0][tail[0]++] = in[i];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int step = 0; step < N; ++step) {
            const int len = tail[b] - head[b];
            if (len >= 3) {
                const uint32_t x = col[b][head[b]], y = col[b][head[b] + 1], z = col[b][head[b] + 2];
                head[b] += 3;
                col[b][tail[b]++] = plane_xor3(x, y, z);
                if (b + 1 < NB) col[b + 1][tail[b + 1]++] = maj3(x, y, z);
            } else if (len == 2) {
                const uint32_t x = col[b][head[b]], y = col[b][head[b] + 1];
                head[b] += 2;
                col[b][tail[b]++] = x ^ y;
                if (b + 1 < NB) col[b + 1][tail[b + 1]++] = x & y;
            }
        }
        S[b] = (tail[b] - head[b]) ? col[b][head[b]] : 0u;
    }
}
// plane of [S >= c], c a compile-time constant after inlining: from the low bit up, ge = c_b ? (S_b & ge) : (S_b | ge)
template <int NB>
__device__ __forceinline__ uint32_t plane_ge(const uint32_t (&S)[NB], int c) {
    if (c <= 0) return ~0u;
    if (c >= (1 << NB)) return 0u;
    uint32_t ge = ~0u;
#pragma unroll
    for (int b = 0; b < NB; ++b) ge = ((c >> b) & 1) ? (S[b] & ge) : (S[b] | ge);
    return ge;
}

}  // namespace
}  // namespace ldpc
