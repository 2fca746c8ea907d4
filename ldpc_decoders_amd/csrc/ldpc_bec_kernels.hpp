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
#include "ldpc_bec_planes.hpp"
#include "ldpc_fused_kernels.hpp"

namespace ldpc {
namespace {

// LDS byte address of the kernel's dynamic shared array, taken through an address-space-3 pointer: spelled via the generic pointer
// ((uint32_t)(uintptr_t)smem) every constant offset becomes "truncate(addrspacecast(@smem) + c)", whose lowering failed to compile for
// the 84 KB shape ("V_CMP_NE_U32_e32 0, $src_shared_base")
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ uint32_t lds_base_of(unsigned char* smem) { return (uint32_t)(uintptr_t)(lds_u8*)smem; }
__device__ __forceinline__ lds_vu32* lds_words_at(uint32_t lds_addr) { return (lds_vu32*)(uintptr_t)lds_addr; }
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
constexpr int BSYS_VERDICT = 8;  // four words per wave: frame masks (changed, erased, wrong); NW <= 8
constexpr int BSYS_HIST = 64;    // Monte-Carlo kernel: histogram of executed sweeps, up to 64 bins; the second accumulator slot's histogram
                                 // follows in the row behind the system row (words 128..191)

template <int DC_, int DV_, int CRW_, int VRW_, int NW_, int VRX_, int DVX_, bool MC_ = false>
struct BecShape {
    static constexpr int DC = DC_, DV = DV_, CRW = CRW_, VRW = VRW_, NW = NW_, VRX = VRX_, DVX = DVX_;
    static constexpr int CR = fused_check_rows(8, DC, DV, CRW, VRW, NW, VRX);  // rows of 64 check slots; check row r belongs to wave r % NW
    static constexpr int VR = VRW * NW;
    static constexpr int VNK = VRX * DVX + (VRW - VRX) * DV;  // gathers of one wave's variable phase == its v2c rows
    static constexpr int VN0 = VRX * DVX;
    static constexpr int CNE = CRW * DC;
    static constexpr int CNW = (CNE + 1) / 2, VNW = (VNK + 1) / 2;
    static constexpr uint32_t SUM_BASE = (uint32_t)NW * VNK * 512u, SYS_BASE = SUM_BASE + (uint32_t)CR * 512u;
    // gather tables: 16-bit BYTE offsets while the slab fits 64 KB (one instruction per gather address: mask or shift of the packed word),
    // element indices (x 8: one more shift) beyond -- becs_build_plan writes them by the same rule
    static constexpr bool BYTE_TAB = SYS_BASE + 1024u <= 65536u;
    // the variable's own last outgoing messages stay in registers where the budget allows, else they are re-read from its rows
    // (lane-contiguous, 2 LDS cycles a row): the Monte-Carlo kernel of the four-wave shape would spill 19 registers with them
    static constexpr bool OWN_REGS = VNK <= 16 && !MC_;
    static constexpr int NOWN = OWN_REGS ? VNK : 1;
    static_assert(CR <= CRW * NW, "check rows fit the waves");
    static_assert(VR <= NW * VNK, "staging area fits the v2c rows");
    static_assert(NW <= 8, "verdict words of the system row");
};

// ---------------- check phase: wave w writes the summary element of its check rows w, w + NW, ...
// `keep`: frame positions whose summaries are written as computed; the others read as "no message" (a position that was just refilled:
// its first variable phase then sees marginal = prior, i.e. v2c = prior and x_hat = the received word -- src/bec.py:86,89)
template <class SH>
__device__ __forceinline__ void becs_check_phase(const unsigned char* smem, const uint32_t (&cn_idx)[SH::CNW], uint32_t sum_vaddr, int w, uint32_t keep) {
    constexpr int DC = SH::DC, CRW = SH::CRW, NW = SH::NW, CR = SH::CR;
    auto cn_addr = [&](int k) -> uint32_t { return SH::BYTE_TAB ? half_of<SH::CNE>(cn_idx, k) : half_of<SH::CNE>(cn_idx, k) << 3; };
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
        // all = every incoming message known so far, two = at least two erased, par = parity of the incoming +1s
        uint32_t all = mg[i & 1][0].k, two = 0, par = mg[i & 1][0].v;
#pragma unroll
        for (int j = 1; j < DC; ++j) {
            const uint32_t kj = mg[i & 1][j].k;
            two = B3(two, all, kj, X0 | (~X1 & ~X2));
            all &= kj;
            par ^= mg[i & 1][j].v;
        }
        const uint32_t sa = B3(all, two, keep, ~X0 & ~X1 & X2);            // exactly one erased
        const uint32_t sb = B3(all, sa, par, X0 | (X1 & X2)) & keep;        // none erased, or the parity the erased edge learns
        if (i * NW + NW - 1 < CR || w + i * NW < CR) lds_st2<i * NW * 512>(sum_vaddr, sa, sb);  // (the last local row exists for the first waves only)
    });
}

// ---------------- variable phase of one wave: marginals, decisions (latched for the frame positions in U), new v2c rows.
// chg: positions where some decision differs from x_hat; era: positions with an erased decision; wrong: positions with a decision that
// is not the sent bit (cwm = all ones for the all-one word) -- all three per lane, over this lane's variables.
template <class SH, bool WRONG>
__device__ __forceinline__ void becs_var_phase(const unsigned char* smem, const uint32_t (&vn_idx)[SH::VNW], uint32_t own_vaddr, uint32_t own_off,
                                               const uint32_t (&pk)[SH::VRW], const uint32_t (&pv)[SH::VRW], uint32_t (&xe)[SH::VRW],
                                               uint32_t (&xv)[SH::VRW], uint32_t (&ok)[SH::NOWN], uint32_t (&ov)[SH::NOWN], uint32_t U, uint32_t cwm,
                                               uint32_t& chg, uint32_t& era, uint32_t& wrong) {
    constexpr int DV = SH::DV, VRW = SH::VRW, VRX = SH::VRX, DVX = SH::DVX, VN0 = SH::VN0;
    constexpr bool OWN_REGS = SH::OWN_REGS;
    auto vn_addr = [&](int k) -> uint32_t { return SH::BYTE_TAB ? half_of<SH::VNK>(vn_idx, k) : half_of<SH::VNK>(vn_idx, k) << 3; };
    // own_at(q): first own-message row of variable round q.  Where the wave re-reads its own last messages from its rows (Monte-Carlo kernel),
    // those reads travel with the summary gathers of the round -- one round ahead -- instead of in front of their first use
    auto own_row0 = [&](int q) -> int { return q < VRX ? q * DVX : VN0 + (q - VRX) * DV; };
    auto var_round = [&](auto Q_, auto WD_, const P2 (&sm)[decltype(WD_)::value], const P2 (&own)[decltype(WD_)::value]) {
        constexpr int q = decltype(Q_)::value, wd = decltype(WD_)::value;
        constexpr int g0 = q < VRX ? q * DVX : VN0 + (q - VRX) * DV;
        uint32_t ck[wd], cv[wd];
        uint32_t in[2 * (wd + 1)];
        in[0] = pv[q];
        in[1] = B3(pk[q], pv[q], pv[q], ~X0 | X1);  // [prior >= 0]
        static_for<0, wd>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            uint32_t k0, v0;
            if constexpr (OWN_REGS) { k0 = ok[g0 + j]; v0 = ov[g0 + j]; }
            else { k0 = own[j].k; v0 = own[j].v; }
            const uint32_t a = sm[j].k, b = sm[j].v;
            ck[j] = B3(a, b, k0, (X1 & ~X0) | (X0 & ~X2));            // echo of a known message, or the one erased edge of its check
            cv[j] = b & B3(a, k0, v0, (X0 & ~X1) | (~X0 & X2));       // b & (a ? ~k0 : v0)
            in[2 + 2 * j] = cv[j];                                   // [c > 0]
            in[3 + 2 * j] = B3(ck[j], cv[j], cv[j], ~X0 | X1);        // [c >= 0]
        });
        uint32_t S[BitsFor<2 * (wd + 1)>::value];
        plane_count<2 * (wd + 1)>(in, S);            // S = marginal + wd + 1
        const uint32_t ge0 = plane_ge(S, wd + 1), ge1 = plane_ge(S, wd + 2), ge2 = plane_ge(S, wd + 3), gem1 = plane_ge(S, wd);
        // decision: sign(marginal) -> 1 / 0 / erased (src/bec.py:119); positions outside U keep theirs
        const uint32_t ne = B3(ge0, ge1, ge1, X0 & ~X1), nv = ge1;
        chg |= B3(ne, xe[q], nv ^ xv[q], (X0 ^ X1) | X2);
        xe[q] = mux(U, ne, xe[q]);
        xv[q] = mux(U, nv, xv[q]);
        era |= xe[q];
        if constexpr (WRONG) wrong |= B3(xv[q], cwm, xe[q], (X0 ^ X1) | X2);
        // v2c_j = sign(marginal - c_j) (src/bec.py:116)
        static_for<0, wd>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            const uint32_t pos = mux(ck[j], mux(cv[j], ge2, ge0), ge1);
            const uint32_t neg = B3(ck[j], B3(cv[j], ge1, gem1, (X0 & ~X1) | (~X0 & ~X2)), ge0, (X0 & X1) | (~X0 & ~X2));
            lds_st2<(g0 + j) * 512>(own_vaddr, pos | neg, pos);
            if constexpr (OWN_REGS) { ok[g0 + j] = pos | neg; ov[g0 + j] = pos; }
        });
    };
    if constexpr (VRX > 0) {
        P2 sw[2][DVX], ow[2][DVX];
#pragma unroll
        for (int j = 0; j < DVX; ++j) {
            sw[0][j] = lds_ld2(smem, vn_addr(j));
            if constexpr (!OWN_REGS) ow[0][j] = lds_ld2(smem, own_off + (own_row0(0) + j) * 512);
        }
        static_for<0, VRX>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            if constexpr (q + 1 < VRX) {
#pragma unroll
                for (int j = 0; j < DVX; ++j) {
                    sw[(q + 1) & 1][j] = lds_ld2(smem, vn_addr((q + 1) * DVX + j));
                    if constexpr (!OWN_REGS) ow[(q + 1) & 1][j] = lds_ld2(smem, own_off + (own_row0(q + 1) + j) * 512);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            var_round(Q_, std::integral_constant<int, DVX>{}, sw[q & 1], ow[q & 1]);
        });
    }
    constexpr int VRN = VRW - VRX;
    if constexpr (VRN > 0) {
        P2 sn[2][DV], on[2][DV];
#pragma unroll
        for (int j = 0; j < DV; ++j) {
            sn[0][j] = lds_ld2(smem, vn_addr(VN0 + j));
            if constexpr (!OWN_REGS) on[0][j] = lds_ld2(smem, own_off + (own_row0(VRX) + j) * 512);
        }
        static_for<0, VRN>([&](auto U_) {
            constexpr int u = decltype(U_)::value;
            if constexpr (u + 1 < VRN) {
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    sn[(u + 1) & 1][j] = lds_ld2(smem, vn_addr(VN0 + (u + 1) * DV + j));
                    if constexpr (!OWN_REGS) on[(u + 1) & 1][j] = lds_ld2(smem, own_off + (own_row0(VRX + u + 1) + j) * 512);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            var_round(std::integral_constant<int, VRX + u>{}, std::integral_constant<int, DV>{}, sn[u & 1], on[u & 1]);
        });
    }
}

// frame masks of all lanes and waves: OR over the workgroup, wave-uniform and identical in every wave (one barrier for NW > 1)
template <class SH, int NWORDS>
__device__ __forceinline__ void becs_exchange(unsigned char* smem, int w, int lane, const uint32_t (&mine)[NWORDS], uint32_t (&all)[NWORDS]) {
    uint32_t red[NWORDS];
#pragma unroll
    for (int k = 0; k < NWORDS; ++k) red[k] = wave_or(mine[k]);
    if constexpr (SH::NW == 1) {
#pragma unroll
        for (int k = 0; k < NWORDS; ++k) all[k] = red[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        lds_vu32* sys = lds_words_at(lds_base_of(smem) + SH::SYS_BASE);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NWORDS; ++k) sys[BSYS_VERDICT + 4 * w + k] = red[k];
        }
        wg_barrier();
#pragma unroll
        for (int k = 0; k < NWORDS; ++k) {
            uint32_t t = 0;
#pragma unroll
            for (int i = 0; i < SH::NW; ++i) t |= sys[BSYS_VERDICT + 4 * i + k];
            all[k] = __builtin_amdgcn_readfirstlane(t);
        }
    }
}

// 64-bit add to a counter in HBM, through a global-address-space pointer: a generic pointer makes the compiler test at run time whether the
// address lies in the LDS aperture (and that expansion fails to compile for one of the shapes below: "V_CMP_NE_U32_e32 0, $src_shared_base")
typedef __attribute__((address_space(1))) u64 global_u64;
__device__ __forceinline__ void global_add(u64* p, u64 v) {
    __hip_atomic_fetch_add((global_u64*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 global_fetch_add(u64* p, u64 v) {
    return __hip_atomic_fetch_add((global_u64*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// slab hand-out: NSHARD counters, a workgroup drains its home shard (blockIdx % 8 -- the XCD it runs on) and then helps the others;
// same scheme as k_fused_bp, unit = a slab of 32 frames
struct SlabTickets {
    static constexpr int NSHARD = 8;
    long long nslab, shard_len;
    int shard, shards_left;
    __device__ __forceinline__ void init(long long B) {
        nslab = (B + BEC_SLAB - 1) / BEC_SLAB;
        shard_len = (nslab + NSHARD - 1) / NSHARD;
        shard = (int)(blockIdx.x % NSHARD);
        shards_left = NSHARD;
    }
    __device__ __forceinline__ long long next(u64* counters, int lane) {  // wave-uniform; -1 when every shard is drained
        long long got = -1;
        while (shards_left > 0) {
            const long long base = shard * shard_len;
            const long long len = (base + shard_len <= nslab ? shard_len : nslab - base);
            u64 t = 0;
            if (lane == 0) t = global_fetch_add(counters + shard * 8, 1ull);
            const long long k = (long long)(((u64)__builtin_amdgcn_readfirstlane((unsigned)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)t));
            if (k < len) { got = base + k; break; }
            shard = (shard + 1) % NSHARD;
            --shards_left;
        }
        return got;
    }
};

// =====================================================================================================================
// Decode kernel: received symbols in HBM, decisions and sweep counts out; a workgroup takes one slab of 32 frames at a time
// (batched bec.SPA.decode, src/bec.py:83-122).
template <int DC, int DV, int CRW, int VRW, int NW, int VRX, int DVX>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? (NW == 4 ? 4 : 2) : 2) void k_fused_becs(const FusedArgs A) {
    using SH = BecShape<DC, DV, CRW, VRW, NW, VRX, DVX>;
    constexpr int VNK = SH::VNK, VN0 = SH::VN0, CNW = SH::CNW, VNW = SH::VNW;
    constexpr uint32_t SUM_BASE = SH::SUM_BASE, SYS_BASE = SH::SYS_BASE;
    constexpr bool OWN_REGS = SH::OWN_REGS;
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

    unsigned valid = 0;  // bit q: slot (w*VRW + q, lane) holds a real variable
#pragma unroll
    for (int q = 0; q < VRW; ++q) valid |= (vslot[q * 64 + lane] >= 0) ? (1u << q) : 0u;
    asm volatile("" : "+v"(valid));
    auto vmask = [&](int q) -> uint32_t { return (uint32_t)__builtin_amdgcn_sbfe((int)valid, q, 1); };  // all ones for a real variable

    const uint32_t lds0 = lds_base_of(smem);
    auto sysw = [&](int i) { return lds_words_at(lds0 + SYS_BASE) + i; };
    if (threadIdx.x == 0) {
        *sysw(BSYS_ZERO) = 0u;
        *sysw(BSYS_ZERO + 1) = 0u;
        *sysw(BSYS_KNOWN0) = ~0u;
        *sysw(BSYS_KNOWN0 + 1) = 0u;
    }
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t own_vaddr = lds0 + (uint32_t)w * VNK * 512u + lane8;        // this wave's v2c rows (DS address)
    const uint32_t sum_vaddr = lds0 + SUM_BASE + (uint32_t)w * 512u + lane8;   // summary rows w, w + NW, ...
    const uint32_t stage_vaddr = lds0 + (uint32_t)w * VRW * 512u + lane8;     // staging rows of this wave's variables
    const uint32_t own_off = (uint32_t)w * VNK * 512u + lane8, stage_off = (uint32_t)w * VRW * 512u + lane8;

    SlabTickets tickets;
    tickets.init(A.B);
    for (;;) {
        long long slab_s = 0;
        if constexpr (NW == 1) {
            slab_s = tickets.next(A.next_frame, lane);
        } else {
            wg_barrier();  // everybody is done with the previous slab (staging, verdict words)
            if (w == 0) {
                const long long s0 = tickets.next(A.next_frame, lane);
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

        // ---- received symbols {0,1,2} from HBM (src/bec.py:76,85) -> bit planes in the staging area: element of slot s = {k: symbol known,
        // v: symbol is 1}.  Lanes take consecutive variables, so each of the 32 frame rows is read in coalesced 64-byte pieces.
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
        if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
        uint32_t pk[VRW], pv[VRW];  // priors; a padded slot and a dead frame of a short slab are a known 0: never erased, never changing
        uint32_t xe[VRW], xv[VRW];  // x_hat: erased plane, value plane (src/bec.py:89: x_hat starts as the received word)
        uint32_t era = 0;
#pragma unroll
        for (int q = 0; q < VRW; ++q) {
            const P2 e = lds_ld2(smem, stage_off + q * 512);
            const uint32_t real = vmask(q) & fmask;
            pk[q] = e.k | ~real;
            pv[q] = e.v & real;
            xe[q] = ~pk[q];
            xv[q] = pv[q];
            era |= xe[q];
        }
        if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();  // staging is read: its rows become message rows
        // v2c = prior on every edge (src/bec.py:86); check messages start at 0, which the first variable phase never reads before a
        // check phase has written the summaries
        uint32_t ok[SH::NOWN], ov[SH::NOWN];
        static_for<0, VRW>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            constexpr int wd = q < VRX ? DVX : DV, g0 = q < VRX ? q * DVX : VN0 + (q - VRX) * DV;
            static_for<0, wd>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                lds_st2<(g0 + j) * 512>(own_vaddr, pk[q], pv[q]);
                if constexpr (OWN_REGS) { ok[g0 + j] = pk[q]; ov[g0 + j] = pv[q]; }
            });
        });
        int it = 0;              // sweeps executed so far
        int itf = 0;             // sweeps of frame `lane` (wave 0, lanes 0..31)
        uint32_t L = fmask;      // frames still in the loop
        auto retire = [&](uint32_t X) {  // frames X leave now, having executed `it` sweeps (the sweep that found no change included)
            if (w == 0) itf = ((X >> (lane & 31)) & 1u) ? it : itf;
        };
        uint32_t mine[2] = {0u, era}, all[2];
        becs_exchange<SH, 2>(smem, w, lane, mine, all);  // the barrier also publishes the initial v2c rows
        for (;;) {
            if (max_iter > 0 && it >= max_iter) break;                 // src/bec.py:96
            if (early) {
                retire(L & ~all[1]);                                    // src/bec.py:97: no erasure left
                L &= all[1];
                if (L == 0u) break;
            }
            becs_check_phase<SH>(smem, cn_idx, sum_vaddr, w, ~0u);
            if constexpr (NW > 1) wg_barrier(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
            uint32_t chg = 0, wrong = 0;
            era = 0;
            becs_var_phase<SH, false>(smem, vn_idx, own_vaddr, own_off, pk, pv, xe, xv, ok, ov, L, 0u, chg, era, wrong);
            mine[0] = chg & L;
            mine[1] = era;
            becs_exchange<SH, 2>(smem, w, lane, mine, all);  // barrier: the new v2c rows are visible, the summaries may be overwritten
            ++it;
            if (early) {
                retire(L & ~all[0]);  // src/bec.py:120: x_hat did not change -- stopping set
                L &= all[0];
            }
        }
        retire(L);  // sweep cap

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

// =====================================================================================================================
// Monte-Carlo kernel (behind ldpc_simulate): channel in the kernel, counters out, and CONTINUOUS REFILL -- the 32 bit positions of a
// workgroup's planes are 32 independent decoder instances.  A frame leaves per upstream's rules (src/bec.py:96-97,120) at its own sweep;
// its position is given the next fresh frame at once instead of idling until the slowest frame of a slab is done (n = 1200, eps = 0.40:
// frames need 16.8 sweeps on average, the slowest of 32 needs 36).  A refilled position costs nothing extra in the sweep: the check
// phase writes "no message" summaries for it, so its first variable phase computes marginal = prior -- v2c = prior, x_hat = received
// word, which is exactly the decoder's initial state (src/bec.py:86,89) -- and it joins the live set one sweep later with age 0.
// Fresh frames come in slabs of 32 (one ticket): their noise is drawn with the ballot transposition into the summary rows (dead between
// a variable phase and the next check phase), from where each wave keeps the known-planes of its variables in registers (`resk`).
template <int DC, int DV, int CRW, int VRW, int NW, int VRX, int DVX>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? (NW == 4 ? 4 : 2) : 2) void k_fused_becs_mc(const FusedArgs A) {
    using SH = BecShape<DC, DV, CRW, VRW, NW, VRX, DVX, true>;
    constexpr int VNK = SH::VNK, CNW = SH::CNW, VNW = SH::VNW;
    constexpr uint32_t SUM_BASE = SH::SUM_BASE, SYS_BASE = SH::SYS_BASE;
    static_assert(SH::VR * 256 <= SH::CR * 512, "the known-plane words of 32 fresh frames fit the summary rows");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int n = A.n, max_iter = A.max_iter;
    const bool early = !(A.flags & FLAG_NO_EARLY_EXIT);
    const int32_t* vslot = A.var_of_slot + w * VRW * 64;
    const uint32_t cwm = A.codeword ? ~0u : 0u;

    uint32_t cn_idx[CNW], vn_idx[VNW];
#pragma unroll
    for (int i = 0; i < CNW; ++i) cn_idx[i] = A.cn_tab[(w * CNW + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < VNW; ++i) vn_idx[i] = A.vn_tab[(w * VNW + i) * 64 + lane];
    unsigned valid = 0;  // bit q: slot (w*VRW + q, lane) holds a real variable
#pragma unroll
    for (int q = 0; q < VRW; ++q) valid |= (vslot[q * 64 + lane] >= 0) ? (1u << q) : 0u;
    asm volatile("" : "+v"(valid));

    const uint32_t lds0 = lds_base_of(smem);
    auto sysw = [&](int i) { return lds_words_at(lds0 + SYS_BASE) + i; };
    if (threadIdx.x == 0) {
        *sysw(BSYS_ZERO) = 0u;
        *sysw(BSYS_ZERO + 1) = 0u;
        *sysw(BSYS_KNOWN0) = ~0u;
        *sysw(BSYS_KNOWN0 + 1) = 0u;
    }
    for (int i = (int)threadIdx.x; i < 128; i += 64 * NW) *sysw(BSYS_HIST + i) = 0u;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t own_vaddr = lds0 + (uint32_t)w * VNK * 512u + lane8;
    const uint32_t sum_vaddr = lds0 + SUM_BASE + (uint32_t)w * 512u + lane8;
    const uint32_t own_off = (uint32_t)w * VNK * 512u + lane8;

    // every position starts empty: a known codeword bit everywhere (never erased, never wrong, never changing)
    uint32_t pk[VRW], pv[VRW], xe[VRW], xv[VRW], resk[VRW];
    uint32_t ok[SH::NOWN], ov[SH::NOWN];
#pragma unroll
    for (int q = 0; q < VRW; ++q) { pk[q] = ~0u; pv[q] = cwm; xe[q] = 0u; xv[q] = cwm; resk[q] = ~0u; }
#pragma unroll
    for (int g = 0; g < SH::NOWN; ++g) { ok[g] = ~0u; ov[g] = cwm; }
    uint32_t L = 0;          // positions whose frame is in the sweep loop
    uint32_t R = 0;          // positions refilled since the last variable phase (initialised by the coming one)
    uint32_t res_left = 0;   // columns of the reservoir (resk) not handed out yet
    bool drained = false;    // no slab ticket left
    uint32_t age = 0;        // lanes 0..31 (of every wave, identically): sweeps executed by the frame at position `lane`
    // Counters (src/main.py:41-45) per ROUND: one launch may carry several rounds of A.B frames (ldpc_simulate_rounds), each with its own
    // counter row, while the 32 positions are refilled across round boundaries -- a workgroup never drains between rounds.  Frames of at
    // most TWO rounds are in flight in a workgroup at a time (slots 0 / 1, tagged row_of[s]); a slab of a third round waits (`pending`)
    // until every frame of one of the two has left and that slot's sums have been added to its row.
    uint32_t S1 = 0;         // positions whose frame belongs to accumulator slot 1 (the others: slot 0)
    int row_of[2] = {-1, -1};
    // 32-bit partial sums, added to the 64-bit rows when a slot is released and at the latest every A.flush_every frames (fused_launch:
    // 2^31 / max(n, max_iter), so that neither the sweep sum nor the bit-error sum of a lane can wrap)
    // Packed (this kernel sits at its register limits): c_isum -- lanes 0..31 the sweep sums of slot 0's positions, lanes 32..63 those of slot 1
    // (`age` is kept in both halves); c_bec -- bit errors of this wave's variables per lane, slot 0 in the low, slot 1 in the high 16 bits;
    // c_tot / c_wec -- frames / word errors, one 16-bit field per slot, wave-uniform (flush_every <= 4096 frames keeps every field in range).
    // A lane's c_bec field of one slot grows by at most VRW per frame and is flushed once the slot has counted flush_every (<= BECS_FLUSH_MAX)
    // frames, checked after up to 32 more have left in the same sweep: (BECS_FLUSH_MAX + 32) * VRW must stay below 2^16 or the sum would
    // carry into the other slot's field silently.
    static_assert((BECS_FLUSH_MAX + 32) * VRW < 65536, "packed 16-bit bit-error counters: lower BECS_FLUSH_MAX for this VRW");
    uint32_t c_isum = 0, c_bec = 0;
    uint32_t c_tot = 0, c_wec = 0;
    uint32_t all[3] = {0u, 0u, 0u};  // last exchange: changed, erased, wrong
#define BECS_SPR ((uint32_t)((A.B + BEC_SLAB - 1) / BEC_SLAB))  /* slabs per round (re-read from the kernel arguments where needed: no live register) */
    int pending = -1;        // a slab ticket taken but not drawn yet
    int res_row = 0;         // round of the frames in the reservoir

    auto flush_slot = [&](auto S_, bool release) {  // wave-uniform: sums of slot s -> its counter row; `release`: the slot is free afterwards
        constexpr int s = decltype(S_)::value;
        u64* row = (u64*)A.counters + (size_t)row_of[s] * (size_t)A.counter_stride;
        u64 b = (u64)((c_bec >> (16 * s)) & 0xffffu);
#pragma unroll
        for (int o = 32; o; o >>= 1) b += __shfl_xor(b, o);
        if (lane == 0 && b) global_add(&row[2], b);
        c_bec &= 0xffff0000u >> (16 * s);
        if (w == 0) {
            u64 t = (lane >> 5) == s ? (u64)c_isum : 0ull;
#pragma unroll
            for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) {
                const uint32_t ft = (c_tot >> (16 * s)) & 0xffffu, fw = (c_wec >> (16 * s)) & 0xffffu;
                if (t) global_add(&row[3], t);
                if (ft) global_add(&row[0], (u64)ft);
                if (fw) global_add(&row[1], (u64)fw);
            }
            if (A.hist_bins > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's ds_add_u32 of the slot have landed (only wave 0 touches the histogram)
                auto hp = sysw(BSYS_HIST + 64 * s + lane);
                const unsigned h = *hp;
                *hp = 0u;
                if (lane < A.hist_bins && h) global_add(&row[4 + lane], (u64)h);
            }
        }
        c_isum = (lane >> 5) == s ? 0u : c_isum;
        c_tot &= 0xffff0000u >> (16 * s);
        c_wec &= 0xffff0000u >> (16 * s);
        if (release) row_of[s] = -1;
    };

    SlabTickets tickets;
    tickets.init((long long)BECS_SPR * BEC_SLAB * (long long)(A.rounds > 0 ? A.rounds : 1));
    const int nblk = (n + 3) >> 2, npair = (nblk + 1) >> 1;
    const uint32_t thr32 = A.bsc_thr > 0xffffffffull ? 0xffffffffu : (uint32_t)A.bsc_thr;
    const bool all_erased = A.bsc_thr > 0xffffffffull;
    if constexpr (NW > 1) wg_barrier();

    for (;;) {
        // ---------------- frames that leave now (src/bec.py:96-97,120), with `age` sweeps executed
        const uint32_t capm = max_iter > 0 ? (uint32_t)__ballot((int)age >= max_iter) : 0u;
        const uint32_t X = L & (early ? (~all[0] | ~all[1] | capm) : capm);
        if (X != 0u) {
            // the leaving frames by accumulator slot: those of round row_of[0], the rest belong to row_of[1]
            const uint32_t in0 = ~S1;
            static_for<0, 2>([&](auto S_) {
                constexpr int sl = decltype(S_)::value;
                const uint32_t XS = sl == 0 ? (X & in0) : (X & ~in0);
                if (XS != 0u) {  // wave-uniform
#pragma unroll
                    for (int q = 0; q < VRW; ++q) c_bec += (uint32_t)__popc(B3(xv[q], cwm, xe[q], (X0 ^ X1) | X2) & XS) << (16 * sl);  // an unresolved erasure counts as a bit error (src/main.py:41)
                    c_isum += ((lane >> 5) == sl && ((XS >> (lane & 31)) & 1u)) ? age : 0u;
                    c_tot += (uint32_t)__popc(XS) << (16 * sl);
                    c_wec += (uint32_t)__popc(all[2] & XS) << (16 * sl);
                    if (w == 0) {
                        if (lane < 32 && ((XS >> lane) & 1u)) {
                            if (A.hist_bins > 0) {
                                typedef __attribute__((address_space(3))) unsigned lds_u32;
                                lds_u32* bin = (lds_u32*)(uintptr_t)(lds0 + SYS_BASE) + BSYS_HIST + 64 * sl + ((int)age < A.hist_bins ? (int)age : A.hist_bins - 1);
                                __hip_atomic_fetch_add(bin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u32
                            }
                        }
                    }
                    if (((c_tot >> (16 * sl)) & 0xffffu) >= (uint32_t)A.flush_every) flush_slot(S_, false);
                }
            });
            L &= ~X;
        }
        // ---------------- refill the free positions from the reservoir; an empty reservoir is restocked with the next slab of 32 frames
        uint32_t F = ~L;
        while (F != 0u) {
            if (res_left == 0u) {
                if (pending < 0) {
                    if (drained) break;
                    long long slab_s = 0;
                    if constexpr (NW == 1) {
                        slab_s = tickets.next(A.next_frame, lane);
                    } else {
                        if (w == 0) {
                            const long long s0 = tickets.next(A.next_frame, lane);
                            if (lane == 0) *sysw(BSYS_TICKET) = (uint32_t)(int32_t)s0;
                        }
                        wg_barrier();
                        slab_s = (long long)(int32_t)__builtin_amdgcn_readfirstlane(*sysw(BSYS_TICKET));
                    }
                    if (slab_s < 0) { drained = true; break; }
                    pending = (int)slab_s;
                }
                // round of the slab and its accumulator slot
                const uint32_t spr = BECS_SPR;
                const uint32_t rnd = (uint32_t)pending / spr, tslab = (uint32_t)pending - rnd * spr;
                if ((int)rnd != row_of[0] && (int)rnd != row_of[1]) {
                    const uint32_t busy = L | R;
                    const uint32_t in0 = ~S1;
                    if (row_of[0] < 0 || (busy & in0) == 0u) {
                        if (row_of[0] >= 0) flush_slot(std::integral_constant<int, 0>{}, true);
                        row_of[0] = (int)rnd;
                    } else if (row_of[1] < 0 || (busy & ~in0) == 0u) {
                        if (row_of[1] >= 0) flush_slot(std::integral_constant<int, 1>{}, true);
                        row_of[1] = (int)rnd;
                    } else {
                        break;  // frames of two other rounds are still in flight: the slab waits for one of them to empty
                    }
                }
                pending = -1;
                res_row = (int)rnd;
                const u64 f0 = (u64)rnd * A.round_stride + (u64)tslab * BEC_SLAB;   // first frame of the slab, relative to A.frame0
                const long long left = A.B - (long long)tslab * BEC_SLAB;            // frames of the round from this slab on
                // Channel, the SAME stream as ldpc_channel (src/bec.py:17: erased where the uniform draw is below p): Philox block b of frame f
                // holds the words of variables 4b..4b+3.  Lanes 0-31 take block 2p for the 32 frames, lanes 32-63 block 2p+1; the comparison's
                // lane mask IS the plane word of a variable (low half: block 2p, high half: 2p+1).
                const int half = lane >> 5, t4 = lane & 3;
                const u64 allm = all_erased ? ~0ull : 0ull;
                for (int p = w; p < npair; p += NW) {
                    const int blk = 2 * p + half;
                    const int var = 4 * blk + t4;
                    // slot of the variable whose plane word this lane will store (lanes 0..3 and 32..35): loaded by every lane (eight distinct
                    // addresses), not under a branch, so that the wait for it sits behind the Philox rounds
                    const int slot = A.slot_of_var[var < n ? var : 0];
                    const Philox4 ph = philox_word_block(A.seed, A.stream, A.frame0 + f0 + (u64)(lane & 31), (uint32_t)blk);
                    uint32_t word = 0;
                    static_for<0, 4>([&](auto T_) {
                        constexpr int t = decltype(T_)::value;
                        const u64 m = __ballot(ph.w[t] < thr32) | allm;  // lanes 0-31: the 32 frames of variable 4 blk + t; lanes 32-63: of the next block
                        write_lane<t>(word, (uint32_t)m);
                        write_lane<32 + t>(word, (uint32_t)(m >> 32));
                    });
                    if ((lane & 31) < 4 && var < n) *lds_words_at(lds0 + SUM_BASE + (uint32_t)slot * 4u) = ~word;  // known plane of the variable
                }
                if constexpr (NW > 1) wg_barrier(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
                for (int q = 0; q < VRW; ++q) {
                    const uint32_t t = *lds_words_at(lds0 + SUM_BASE + (uint32_t)((w * VRW + q) * 64 + lane) * 4u);
                    resk[q] = t | ~(uint32_t)__builtin_amdgcn_sbfe((int)valid, q, 1);  // padded slot: known
                }
                if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();  // the summary rows may be written again
                res_left = left >= BEC_SLAB ? ~0u : ((1u << (int)left) - 1u);
            }
            const int d = __builtin_ctz(F), sc = __builtin_ctz(res_left);
            F &= F - 1u;
            res_left &= res_left - 1u;
            const uint32_t dm = 1u << d;
            R |= dm;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const uint32_t kb = ((resk[q] >> sc) & 1u) << d;
                pk[q] = (pk[q] & ~dm) | kb;
                pv[q] = (pv[q] & ~dm) | (kb & cwm);
            }
            age = (lane & 31) == d ? 0u : age;
            S1 = res_row == row_of[1] ? (S1 | dm) : (S1 & ~dm);
        }
        if ((L | R) == 0u) {
            if (pending < 0) break;  // nothing in flight and nothing left to start
            continue;                // (a waiting slab: both slots are free now, the refill above takes it on the next trip)
        }
        // ---------------- one sweep of all 32 positions
        becs_check_phase<SH>(smem, cn_idx, sum_vaddr, w, ~R);
        if constexpr (NW > 1) wg_barrier(); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        uint32_t chg = 0, era = 0, wrong = 0;
        const uint32_t U = L | R;
        becs_var_phase<SH, true>(smem, vn_idx, own_vaddr, own_off, pk, pv, xe, xv, ok, ov, U, cwm, chg, era, wrong);
        const uint32_t mine[3] = {chg & L, era & U, wrong & U};
        becs_exchange<SH, 3>(smem, w, lane, mine, all);
        age += (L >> (lane & 31)) & 1u;
        all[0] |= R;  // a position initialised by this sweep has not been compared with anything yet
        L = U;
        R = 0u;
    }
    // ---------------- counters of the workgroup -> their rows (src/main.py:41-45)
    if constexpr (NW > 1) wg_barrier();
    if (row_of[0] >= 0) flush_slot(std::integral_constant<int, 0>{}, true);
    if (row_of[1] >= 0) flush_slot(std::integral_constant<int, 1>{}, true);
}

template <int DC, int DV, int CRW, int VRW, int NW, int VRX = 0, int DVX = DV>
constexpr ShapeEntry shape_entry_becs() {
    return ShapeEntry{ALG_BEC, DC, DV, CRW, VRW, NW, VRX, DVX, (const void*)k_fused_becs<DC, DV, CRW, VRW, NW, VRX, DVX>,
                      (const void*)k_fused_becs_mc<DC, DV, CRW, VRW, NW, VRX, DVX>, 8};
}

}  // namespace
}  // namespace ldpc
