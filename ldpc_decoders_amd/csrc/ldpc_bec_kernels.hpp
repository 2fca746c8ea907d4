// Bit-sliced erasure decoder on the LDS -- device code, included by ldpc_fused_shapes_bec.hip.
//
// The decoder behind the reference's `bec` selector (src/bec.py:70-125) passes messages from a THREE-symbol alphabet
// {-1 (bit 0), +1 (bit 1), 0 (erased)}.  Here a message is two bits -- `k` (known) and `v` (value; v implies k) -- and one LDS element
// holds those two bits for a SLAB of 32 frames: an 8-byte element = {k plane, v plane}, bit f = frame f of the slab.  One workgroup of
// NW wavefronts owns one slab for all its sweeps; every rule of the decoder becomes a handful of bitwise instructions that serve 32
// frames at once, and the LDS moves 2 bits per message instead of the 32 the float-carried form moved ("2-bit messages").
//
// Layout of a slab in the LDS (rows of 64 eight-byte elements, one per lane):
//   v2c rows   [NW * VNK]  variable -> check messages, VARIABLE-major: row (w, g) = gather position g of wave w's variable phase, written
//                          lane-contiguously by the variable's owner (compile-time addresses), gathered by the checks;
//   summaries  [CR]        one element per check slot: the check's whole answer in two planes (A, B) -- see below;
//   system row             {0,0} (what a missing edge of a variable reads), {~0,0} (a known 0: what a missing edge of a short check row
//                          reads), the slab ticket, the verdict words of the waves.
// The first VR rows double as the staging area in which a slab's received word is transposed into bit planes before the first sweep
// and its decisions are transposed back after the last one.
//
// Check rule (src/bec.py:99-112).  With e = number of erased incoming messages and p = parity of the incoming +1s, the outgoing message
// on edge j is: e = 0 -> the incoming message itself ("echo"); e = 1 -> the parity on the erased edge, 0 elsewhere; e > 1 -> 0.  That is a
// function of the check's (e == 0, e == 1, p) and of the message the variable itself sent, so the check does not store dc messages: it
// stores ONE element, A = (e == 1), B = (e == 0) | (A & p), and the variable rebuilds its incoming message from (A, B) and its own
// last outgoing (k, v):   c_k = (B & ~A) | (A & ~k),   c_v = B & (A ? ~k : v).   dc stores per check become one.
//
// Variable rule (src/bec.py:115-119): marginal = prior + sum of incoming, v2c_j = sign(marginal - c_j), decision = sign(marginal).
// Bit-sliced: every term t in {-1,0,+1} contributes [t > 0] + [t >= 0] to a counter S (column compression with full adders), so
// marginal = S - (D + 1) for D incoming messages; the five facts the outputs need (marginal >= 2, >= 1, >= 0, <= -1, <= -2) are
// comparisons of S with constants, and sign(marginal - c_j) is a selection among them by (c_k, c_v).  Exact for ANY received word,
// contradictory ones included (the marginal is a true integer sum, not a "known / unknown" flag).
//
// Exits are per FRAME as upstream (src/bec.py:96-97,120): a mask of live frames gates the decision update, so a frame that has left
// keeps the word it left with while the rest of its slab goes on; a slab is finished when its last frame has left.
#pragma once
#include "ldpc_fused_kernels.hpp"

namespace ldpc {
namespace {

constexpr int BEC_SLAB = 32;  // frames per slab == bits of a plane word

struct P2 {
    uint32_t k, v;
};
__device__ __forceinline__ P2 lds_ld2(const unsigned char* base, uint32_t byte_off) {
    const uint2 t = *reinterpret_cast<const uint2*>(base + byte_off);
    return P2{t.x, t.y};
}
// one row element, 8 bytes, ds_write_b64 with an immediate offset (inline asm for the reason given at lds_st64)
template <int OFF>
__device__ __forceinline__ void lds_st2(uint32_t vaddr, uint32_t k, uint32_t v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    const u64 pair = ((u64)v << 32) | k;
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(vaddr), "v"(pair), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_st2_dyn(uint32_t vaddr, uint32_t k, uint32_t v) {
    const u64 pair = ((u64)v << 32) | k;
    asm volatile("ds_write_b64 %0, %1" ::"v"(vaddr), "v"(pair) : "memory");
}

__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
__device__ __forceinline__ uint32_t mux(uint32_t s, uint32_t a, uint32_t b) { return (s & a) | (~s & b); }  // s ? a : b, bitwise (one v_bfi / v_bitop3)

// bits needed for a count in [0, N]
template <int N>
struct BitsFor {
    static constexpr int value = (N < 2) ? 1 : (N < 4) ? 2 : (N < 8) ? 3 : (N < 16) ? 4 : (N < 32) ? 5 : 6;
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
                col[b][tail[b]++] = xor3(x, y, z);
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

// OR over the 64 lanes of a wave, result wave-uniform: four DPP steps inside each row of 16 lanes, then the four rows by v_readlane
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v |= __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, true);  // row_half_mirror
    v |= __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, true);  // row_mirror
    return __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
}

// system row, dword indices
constexpr int BSYS_ZERO = 0;     // {0,0}: summary read by a missing edge of a variable -> incoming message 0
constexpr int BSYS_KNOWN0 = 2;   // {~0,0}: message read by a missing edge of a short check row: a known 0, neutral for erasure count and parity
constexpr int BSYS_TICKET = 4;
constexpr int BSYS_VERDICT = 8;  // two words per wave: (changed, erased) frame masks; NW <= 16
constexpr int BSYS_WRONG = 48;   // one word per wave: frames with a wrong decision (SIM)

template <int DC, int DV, int CRW, int VRW, int NW, bool SIM, int VRX, int DVX>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? (NW == 4 ? 4 : 2) : 2) void k_fused_becs(const FusedArgs A) {
    constexpr int CR = fused_check_rows(8, DC, DV, CRW, VRW, NW, VRX);  // rows of 64 check slots; check row r belongs to wave r % NW
    constexpr int VR = VRW * NW;
    constexpr int VNK = VRX * DVX + (VRW - VRX) * DV;  // gathers of one wave's variable phase == its v2c rows
    constexpr int VN0 = VRX * DVX;
    constexpr int CNE = CRW * DC;
    constexpr int CNW = (CNE + 1) / 2, VNW = (VNK + 1) / 2;
    constexpr uint32_t SUM_BASE = (uint32_t)NW * VNK * 512u, SYS_BASE = SUM_BASE + (uint32_t)CR * 512u;
    constexpr bool OWN_REGS = VNK <= 16;  // the variable's own last outgoing messages stay in registers (else re-read from its rows)
    static_assert(CR <= CRW * NW, "check rows fit the waves");
    static_assert(VR <= NW * VNK, "staging area fits the v2c rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int n = A.n, max_iter = A.max_iter;
    const bool early = !(A.flags & FLAG_NO_EARLY_EXIT);
    const int32_t* vslot = A.var_of_slot + w * VRW * 64;

    // gather addresses: 16-bit ELEMENT indices (byte offset / 8), two per word, resident for the whole launch
    uint32_t cn_idx[CNW], vn_idx[VNW];
#pragma unroll
    for (int i = 0; i < CNW; ++i) cn_idx[i] = A.cn_tab[(w * CNW + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < VNW; ++i) vn_idx[i] = A.vn_tab[(w * VNW + i) * 64 + lane];
    auto cn_addr = [&](int k) -> uint32_t { return half_of<CNE>(cn_idx, k) << 3; };
    auto vn_addr = [&](int k) -> uint32_t { return half_of<VNK>(vn_idx, k) << 3; };

    unsigned valid = 0;  // bit q: slot (w*VRW + q, lane) holds a real variable
#pragma unroll
    for (int q = 0; q < VRW; ++q) valid |= (vslot[q * 64 + lane] >= 0) ? (1u << q) : 0u;
    asm volatile("" : "+v"(valid));
    auto vmask = [&](int q) -> uint32_t { return (uint32_t)__builtin_amdgcn_sbfe((int)valid, q, 1); };  // all ones for a real variable

    auto sysw = [&](int i) { return lds_word(smem + SYS_BASE) + i; };
    if (threadIdx.x == 0) {
        *sysw(BSYS_ZERO) = 0u;
        *sysw(BSYS_ZERO + 1) = 0u;
        *sysw(BSYS_KNOWN0) = ~0u;
        *sysw(BSYS_KNOWN0 + 1) = 0u;
    }
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t own_vaddr = (uint32_t)(uintptr_t)smem + (uint32_t)w * VNK * 512u + lane8;        // this wave's v2c rows (DS address)
    const uint32_t sum_vaddr = (uint32_t)(uintptr_t)smem + SUM_BASE + (uint32_t)w * 512u + lane8;   // summary rows w, w + NW, ...
    const uint32_t stage_vaddr = (uint32_t)(uintptr_t)smem + (uint32_t)w * VRW * 512u + lane8;     // staging rows of this wave's variables
    const uint32_t own_off = (uint32_t)w * VNK * 512u + lane8, stage_off = (uint32_t)w * VRW * 512u + lane8;

    // Monte-Carlo counters of the workgroup (wave 0): histogram of executed sweeps one bin per lane, the rest wave-uniform
    unsigned hacc = 0;
    u64 c_tot = 0, c_wec = 0, c_iter = 0, c_bec = 0;  // c_bec: per lane (bit errors of the variables it owns)

    constexpr int NSHARD = 8;
    const long long nslab = (A.B + BEC_SLAB - 1) / BEC_SLAB;
    const long long shard_len = (nslab + NSHARD - 1) / NSHARD;
    int shard = (int)(blockIdx.x % NSHARD), shards_left = NSHARD;
    auto next_slab = [&]() -> long long {  // wave-uniform; -1 when every shard is drained (same scheme as k_fused_bp, unit = slab)
        long long got = -1;
        while (shards_left > 0) {
            const long long base = shard * shard_len;
            const long long len = (base + shard_len <= nslab ? shard_len : nslab - base);
            u64 t = 0;
            if (lane == 0) t = atomicAdd(A.next_frame + shard * 8, 1ull);
            const long long k = (long long)(((u64)__builtin_amdgcn_readfirstlane((unsigned)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)t));
            if (k < len) { got = base + k; break; }
            shard = (shard + 1) % NSHARD;
            --shards_left;
        }
        return got;
    };

    for (;;) {
        long long slab_s = 0;
        if constexpr (NW == 1) {
            slab_s = next_slab();
        } else {
            wg_barrier();  // everybody is done with the previous slab (staging, verdict words)
            if (w == 0) {
                const long long s0 = next_slab();
                if (lane == 0) *sysw(BSYS_TICKET) = (uint32_t)(int32_t)s0;
            }
            wg_barrier();
            slab_s = (long long)(int32_t)__builtin_amdgcn_readfirstlane(*sysw(BSYS_TICKET));
        }
        if (slab_s < 0) break;
        const u64 f0 = (u64)slab_s * BEC_SLAB;                     // first frame of the slab
        const long long left = A.B - (long long)f0;
        const int nfr = left < BEC_SLAB ? (int)left : BEC_SLAB;    // frames of this slab (the last one may be short)
        const uint32_t fmask = nfr >= 32 ? ~0u : ((1u << nfr) - 1u);

        // ---- received word -> bit planes in the staging area: element of slot s = {k: symbol known, v: symbol is 1}
        if constexpr (SIM) {
            // Channel in the kernel, the SAME stream as ldpc_channel (src/bec.py:17: erased where the uniform draw is below p): Philox block b of
            // frame f holds the words of variables 4b..4b+3.  Lanes 0-31 take block 2p for the 32 frames, lanes 32-63 block 2p+1; the
            // comparison's lane mask IS the plane word of a variable (low half: block 2p, high half: 2p+1).
            const int nblk = (n + 3) >> 2, npair = (nblk + 1) >> 1;
            const uint32_t thr32 = A.bsc_thr > 0xffffffffull ? 0xffffffffu : (uint32_t)A.bsc_thr;
            const bool all_erased = A.bsc_thr > 0xffffffffull;
            const int half = lane >> 5, t4 = lane & 3;
            for (int p = w; p < npair; p += NW) {
                const int blk = 2 * p + half;
                const int var = 4 * blk + t4;
                int slot = 0;
                const bool writer = (lane & 31) < 4 && var < n;
                if (writer) slot = A.slot_of_var[var];  // issued ahead of the Philox rounds that hide it
                const Philox4 ph = philox_word_block(A.seed, A.stream, A.frame0 + f0 + (u64)(lane & 31), (uint32_t)blk);
                uint32_t word = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u64 m = __ballot(all_erased || ph.w[t] < thr32);
                    word = (lane == t) ? (uint32_t)m : word;
                    word = (lane == 32 + t) ? (uint32_t)(m >> 32) : word;
                }
                if (writer) {
                    const uint32_t known = ~word;
                    *reinterpret_cast<uint2*>(smem + (uint32_t)slot * 8u) = make_uint2(known, A.codeword ? known : 0u);
                }
            }
        } else {
            // received symbols {0,1,2} from HBM (src/bec.py:76,85): lanes take consecutive variables, so each of the 32 frame rows is read in
            // coalesced 64-byte pieces
            for (int v = (int)threadIdx.x; v < n; v += 64 * NW) {
                const uint8_t* yp = A.y0 + f0 * (u64)n + (u64)v;
                uint32_t kk = 0, vv = 0;
                for (int f = 0; f < nfr; ++f) {
                    const uint32_t y = yp[(size_t)f * n];
                    kk |= (y != 2u ? 1u : 0u) << f;
                    vv |= (y == 1u ? 1u : 0u) << f;
                }
                *reinterpret_cast<uint2*>(smem + (uint32_t)A.slot_of_var[v] * 8u) = make_uint2(kk, vv);
            }
        }
        if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
        uint32_t pk[VRW], pv[VRW];  // priors; frames beyond the batch and padded slots are "known 0" / "erased, masked out"
        uint32_t xe[VRW], xv[VRW];  // x_hat: erased plane, value plane (src/bec.py:89: x_hat starts as the received word)
        uint32_t era = 0;
#pragma unroll
        for (int q = 0; q < VRW; ++q) {
            const P2 e = lds_ld2(smem, stage_off + q * 512);
            const uint32_t vm = vmask(q);
            pk[q] = (e.k & vm & fmask) | (vm & ~fmask);  // padded slot: erased (k = 0) and silent; dead frame of a short slab: known 0
            pv[q] = e.v & vm & fmask;
            xe[q] = ~pk[q];
            xv[q] = pv[q];
            era |= xe[q] & vm;
        }
        if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();  // staging is read: its rows become message rows
        // v2c = prior on every edge (src/bec.py:86); check messages start at 0, which the first variable phase never reads before a
        // check phase has written the summaries
        uint32_t ok[OWN_REGS ? VNK : 1], ov[OWN_REGS ? VNK : 1];
        static_for<0, VRW>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int wd = q < VRX ? DVX : DV, g0 = q < VRX ? q * DVX : VN0 + (q - VRX) * DV;
            static_for<0, wd>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                lds_st2<(g0 + j) * 512>(own_vaddr, pk[q], pv[q]);
                if constexpr (OWN_REGS) { ok[g0 + j] = pk[q]; ov[g0 + j] = pv[q]; }
            });
        });
        // frame masks are wave-uniform and identical in every wave
        auto exchange = [&](uint32_t chg_lane, uint32_t era_lane, uint32_t& chg_all, uint32_t& era_all) {  // contains one barrier for NW > 1
            const uint32_t c = wave_or(chg_lane), e = wave_or(era_lane);
            if constexpr (NW == 1) {
                chg_all = c;
                era_all = e;
                __builtin_amdgcn_wave_barrier();
            } else {
                if (lane == 0) {
                    *sysw(BSYS_VERDICT + 2 * w) = c;
                    *sysw(BSYS_VERDICT + 2 * w + 1) = e;
                }
                wg_barrier();
                uint32_t ca = 0, ea = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    ca |= *sysw(BSYS_VERDICT + 2 * i);
                    ea |= *sysw(BSYS_VERDICT + 2 * i + 1);
                }
                chg_all = __builtin_amdgcn_readfirstlane(ca);
                era_all = __builtin_amdgcn_readfirstlane(ea);
            }
        };
        int it = 0;              // sweeps executed so far
        int itf = 0;             // decode: sweeps of frame `lane` (wave 0, lanes 0..31)
        uint32_t L = fmask;      // frames still in the loop
        auto retire = [&](uint32_t X) {  // frames X leave now, having executed `it` sweeps (the sweep that found no change included)
            if (X == 0u) return;
            if constexpr (SIM) {
                if (w == 0) {
                    const int cnt = __popc(X);
                    c_iter += (u64)cnt * (u64)it;
                    const int bin = it < A.hist_bins ? it : A.hist_bins - 1;
                    hacc += (lane == bin) ? (unsigned)cnt : 0u;
                }
            } else {
                if (w == 0) itf = ((X >> (lane & 31)) & 1u) ? it : itf;
            }
        };
        uint32_t chg_all = 0, era_all = 0;
        exchange(0u, era, chg_all, era_all);  // the barrier also publishes the initial v2c rows
        for (;;) {
            if (max_iter > 0 && it >= max_iter) break;                 // src/bec.py:96
            if (early) {
                retire(L & ~era_all);                                   // src/bec.py:97: no erasure left
                L &= era_all;
                if (L == 0u) break;
            }
            // ---------------- check phase: one summary element per check
            {
                P2 mg[2][DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) mg[0][j] = lds_ld2(smem, cn_addr(j));
                static_for<0, CRW>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    if constexpr (i + 1 < CRW) {
#pragma unroll
                        for (int j = 0; j < DC; ++j) mg[(i + 1) & 1][j] = lds_ld2(smem, cn_addr((i + 1) * DC + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    uint32_t one = 0, two = 0, par = 0;  // >= 1 / >= 2 erased incoming messages; parity of the incoming +1s
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        const uint32_t nk = ~mg[i & 1][j].k;
                        two |= one & nk;
                        one |= nk;
                        par ^= mg[i & 1][j].v;
                    }
                    const uint32_t sa = one & ~two;           // exactly one erased
                    const uint32_t sb = ~one | (sa & par);    // none erased, or the parity the erased edge learns
                    if (i * NW + NW - 1 < CR || w + i * NW < CR) lds_st2<i * NW * 512>(sum_vaddr, sa, sb);  // (the last local row exists for the first waves only)
                });
            }
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
            // ---------------- variable phase
            uint32_t chg = 0;
            era = 0;
            auto var_round = [&](auto Q_, auto WD_, const P2 (&sm)[decltype(WD_)::value]) {
                constexpr int q = decltype(Q_)::value, wd = decltype(WD_)::value;
                constexpr int g0 = q < VRX ? q * DVX : VN0 + (q - VRX) * DV;
                uint32_t ck[wd], cv[wd];
                uint32_t in[2 * (wd + 1)];
                in[0] = pv[q];
                in[1] = ~pk[q] | pv[q];  // [prior >= 0]
                static_for<0, wd>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    uint32_t k0, v0;
                    if constexpr (OWN_REGS) { k0 = ok[g0 + j]; v0 = ov[g0 + j]; }
                    else { const P2 o = lds_ld2(smem, own_off + (g0 + j) * 512); k0 = o.k; v0 = o.v; }
                    const uint32_t a = sm[j].k, b = sm[j].v;
                    ck[j] = (b & ~a) | (a & ~k0);            // echo of a known message, or the one erased edge of its check
                    cv[j] = b & mux(a, ~k0, v0);
                    in[2 + 2 * j] = cv[j];                   // [c > 0]
                    in[3 + 2 * j] = ~ck[j] | cv[j];          // [c >= 0]
                });
                uint32_t S[BitsFor<2 * (wd + 1)>::value];
                plane_count<2 * (wd + 1)>(in, S);            // S = marginal + wd + 1
                const uint32_t ge0 = plane_ge(S, wd + 1), ge1 = plane_ge(S, wd + 2), ge2 = plane_ge(S, wd + 3), gem1 = plane_ge(S, wd);
                // decision: sign(marginal) -> 1 / 0 / erased (src/bec.py:119); frames that have left keep theirs
                const uint32_t ne = ge0 & ~ge1, nv = ge1;
                chg |= (ne ^ xe[q]) | (nv ^ xv[q]);
                xe[q] = mux(L, ne, xe[q]);
                xv[q] = mux(L, nv, xv[q]);
                era |= xe[q] & vmask(q);
                // v2c_j = sign(marginal - c_j) (src/bec.py:116)
                static_for<0, wd>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    const uint32_t pos = mux(ck[j], mux(cv[j], ge2, ge0), ge1);
                    const uint32_t neg = mux(ck[j], mux(cv[j], ~ge1, ~gem1), ~ge0);
                    lds_st2<(g0 + j) * 512>(own_vaddr, pos | neg, pos);
                    if constexpr (OWN_REGS) { ok[g0 + j] = pos | neg; ov[g0 + j] = pos; }
                });
            };
            if constexpr (VRX > 0) {
                P2 sw[2][DVX];
#pragma unroll
                for (int j = 0; j < DVX; ++j) sw[0][j] = lds_ld2(smem, vn_addr(j));
                static_for<0, VRX>([&](auto Q_) {
                    constexpr int q = decltype(Q_)::value;
                    if constexpr (q + 1 < VRX) {
#pragma unroll
                        for (int j = 0; j < DVX; ++j) sw[(q + 1) & 1][j] = lds_ld2(smem, vn_addr((q + 1) * DVX + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    var_round(Q_, std::integral_constant<int, DVX>{}, sw[q & 1]);
                });
            }
            constexpr int VRN = VRW - VRX;
            if constexpr (VRN > 0) {
                P2 sn[2][DV];
#pragma unroll
                for (int j = 0; j < DV; ++j) sn[0][j] = lds_ld2(smem, vn_addr(VN0 + j));
                static_for<0, VRN>([&](auto U_) {
                    constexpr int u = decltype(U_)::value;
                    if constexpr (u + 1 < VRN) {
#pragma unroll
                        for (int j = 0; j < DV; ++j) sn[(u + 1) & 1][j] = lds_ld2(smem, vn_addr(VN0 + (u + 1) * DV + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    var_round(std::integral_constant<int, VRX + u>{}, std::integral_constant<int, DV>{}, sn[u & 1]);
                });
            }
            exchange(chg & L, era, chg_all, era_all);  // barrier: the new v2c rows are visible, the summaries may be overwritten
            ++it;
            if (early) {
                retire(L & ~chg_all);  // src/bec.py:120: x_hat did not change -- stopping set
                L &= chg_all;
            }
        }
        retire(L);  // sweep cap

        if constexpr (SIM) {
            // errors against the all-`codeword` word (src/main.py:41-45); an unresolved erasure counts as a bit error
            uint32_t wrong = 0;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const uint32_t wr = ((A.codeword ? ~xv[q] : xv[q]) | xe[q]) & vmask(q) & fmask;
                c_bec += (u64)__popc(wr);
                wrong |= wr;
            }
            uint32_t wrong_all = wave_or(wrong);
            if constexpr (NW > 1) {
                if (lane == 0) *sysw(BSYS_WRONG + w) = wrong_all;
                wg_barrier();
                if (w == 0) {
                    uint32_t t = 0;
#pragma unroll
                    for (int i = 0; i < NW; ++i) t |= *sysw(BSYS_WRONG + i);
                    wrong_all = __builtin_amdgcn_readfirstlane(t);
                }
            }
            if (w == 0) {
                c_tot += (u64)nfr;
                c_wec += (u64)__popc(wrong_all);
            }
        } else {
            // decisions back to [frame][variable] bytes through the staging rows; {0,1,2 = still erased}
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();  // every wave is out of the sweep loop
#pragma unroll
            for (int q = 0; q < VRW; ++q) lds_st2_dyn(stage_vaddr + q * 512, xe[q], xv[q]);
            if constexpr (NW > 1) wg_barrier(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
            for (int v = (int)threadIdx.x; v < n; v += 64 * NW) {
                const P2 e = lds_ld2(smem, (uint32_t)A.slot_of_var[v] * 8u);
                uint8_t* xp = A.xhat + f0 * (u64)n + (u64)v;
                for (int f = 0; f < nfr; ++f) xp[(size_t)f * n] = ((e.k >> f) & 1u) ? (uint8_t)2 : (uint8_t)((e.v >> f) & 1u);
            }
            if (w == 0 && lane < nfr) A.iters[f0 + lane] = itf;
        }
    }
    if constexpr (SIM) {
        // bit errors: per-lane sums of every wave; the rest lives in wave 0
        u64 b = c_bec;
#pragma unroll
        for (int o = 32; o; o >>= 1) b += __shfl_xor(b, o);
        if (lane == 0 && b) atomicAdd(&A.counters[2], b);
        if (w == 0) {
            if (lane == 0) {
                if (c_tot) atomicAdd(&A.counters[0], c_tot);
                if (c_wec) atomicAdd(&A.counters[1], c_wec);
                if (c_iter) atomicAdd(&A.counters[3], c_iter);
            }
            if (lane < A.hist_bins && hacc) atomicAdd(&A.counters[4 + lane], (u64)hacc);
        }
    }
}

template <int DC, int DV, int CRW, int VRW, int NW, int VRX = 0, int DVX = DV>
constexpr ShapeEntry shape_entry_becs() {
    return ShapeEntry{ALG_BEC, DC, DV, CRW, VRW, NW, VRX, DVX, (const void*)k_fused_becs<DC, DV, CRW, VRW, NW, false, VRX, DVX>,
                      (const void*)k_fused_becs<DC, DV, CRW, VRW, NW, true, VRX, DVX>, 8};
}

}  // namespace
}  // namespace ldpc
