// Fused on-chip backend -- device code (kernels + shape descriptors), included by the ldpc_fused_shapes_*.hip translation units.
//
// Fused on-chip backend: one workgroup of NW wavefronts owns one frame for ALL of its sweeps; messages never leave the CU.
//
// Regime: (dv,dc)-regular codes whose per-frame state fits the LDS (n = 1200 (3,6): 20 KB -> 8 frames per CU).
// The HBM traffic of a frame shrinks from sizeof(T)(4E+n) PER SWEEP (streaming backend) to its priors in and its
// decisions out, once; the kernel is bound by LDS gathers and VALU instead.
//
// Mapping (all tables are built once on the host, FusedPlan):
//   * wave w, lane L owns check slots (w*CRW + r, L), r < CRW, and variable slots (w*VRW + q, L), q < VRW.
//   * registers: the check->variable messages of the owned checks (c2v_old[CRW][DC]), the priors of the owned
//     variables, and every gather address (packed 16-bit LDS byte offsets) -- loaded once per launch.
//   * LDS (per frame): marg[VR*64]  marginal of variable slot s at dword s            (VR = NW*VRW)
//                      c2v [CR*DC*64 (+64)]  message of (check slot (R,L), edge position j) at dword (R*DC+j)*64+L;
//                                            an optional last row stays 0 (target of the gathers of missing edges)
//   * check phase : v2c_j = marg[var] - c2v_old_j  (dc LDS gathers), leave-one-out min / join + sign parity, write c2v
//                   (lane-contiguous, conflict-free); the syndrome of the previous decisions falls out of the same
//                   gathers (sign of marg) -- that is the reference's early-exit test (src/bpa.py:29).
//   * variable phase: marginal = prior + ((0 + c_a) + c_b) + c_c in ascending edge order (dv LDS gathers), write marg.
//   NW = 1: LDS is private to the wave and DS operations of one wave execute in order, so the phases need no barrier.
//   NW = 2: twice the resident waves per CU (the kernel is latency-bound at 2 waves per SIMD); one s_barrier after each
//           phase, the syndrome verdicts of the waves are exchanged through padded c2v slots each wave owns.
//   NW = 4: codes up to m = 1536 / n = 2816 (48 KB of LDS per frame, 3 frames per CU), same hand-off scheme.
//           (NW = 4 with 3 check rounds per wave was measured for n = 1200 in fp32: slower than NW = 2, 4.15 vs 3.74 ms.  The fp64
//           kernel below does run n = 1200 with four waves -- on 10 check rows instead of 12, see fused_check_rows.)
//
// Arithmetic identical to the streaming backend / reference (src/bpa.py:17-63, 86-102): the leave-one-out minimum
// equals "second minimum at the first arg-min, first minimum elsewhere"; min/compare/negate/add/sub only.
#pragma once
#include <type_traits>

#include "ldpc_cn.hpp"
#include "ldpc_common.hpp"
#include "ldpc_rng.hpp"

namespace ldpc {

// One instantiated kernel shape: what the host-side plan builder (ldpc_fused.hip) chooses from.
struct ShapeEntry {
    int alg, DC, DV, CRW, VRW, NW, VRX, DVX;  // VRX: variable rounds of other widths than DV (irregular codes), 0 for regular -- low four bits: the
                                              // wave's FIRST rounds gather DVX messages ("wide"), the bits above: its LAST rounds gather two ("pair" rounds)
    const void* kernel;      // decode: priors in, decisions out
    const void* kernel_sim;  // simulate: noise in the kernel, counters out (null: decode only)
    int esz = 4;             // bytes per LDS element: 4 (fp32 kernels), 8 (fp64 kernels)
    // exact-in-fp32 mode (fp32 min-sum on priors quantised to a 2^-k grid, LDPC_FLAG_PRIOR_GRID): the same kernels with the exactness
    // guard compiled in; null where no such variant is built
    const void* kernel_grid = nullptr;
    const void* kernel_sim_grid = nullptr;
};
// Rows of 64 check slots a frame holds in the LDS.  Normally CRW per wave.  The fp64 shapes of four and more waves for regular codes
// keep only the rows a code can fill -- E = m DC <= n DV, hence m <= VR 64 DV / DC -- and the last wave(s) run fewer rows: the (3,6)
// n = 1200 frame is 10 check rows + 20 marginal rows = 40 KB, four frames per CU, with FOUR waves each (rows 3 + 3 + 3 + 1) instead of two.
// A wave's variable rounds: the first wide_rounds(VRX) gather DVX messages per variable, the last pair_rounds(VRX) two, those between DV.  (The
// two counts share one template argument so that the kernels of all other shapes keep their names: PMC counters are filed under them.)
constexpr int wide_rounds(int vrx_arg) { return vrx_arg & 15; }
constexpr int pair_rounds(int vrx_arg) { return vrx_arg >> 4; }
constexpr int vrx_arg(int wide, int pairs) { return wide + 16 * pairs; }
constexpr int fused_check_rows(int esz, int DC, int DV, int CRW, int VRW, int NW, int VRX) {
    const int all = CRW * NW, fill = (VRW * NW * DV + DC - 1) / DC;
    return (esz == 8 && VRX == 0 && NW >= 4 && fill < all) ? fill : all;
}
inline int fused_check_rows(const ShapeEntry& s) { return fused_check_rows(s.esz, s.DC, s.DV, s.CRW, s.VRW, s.NW, s.VRX); }

// shape tables, one per translation unit (built in parallel); preference order = table order
const ShapeEntry* fused_shapes_f32_dc6(int* count);
const ShapeEntry* fused_shapes_f32_dcx(int* count);
const ShapeEntry* fused_shapes_f64_dc6(int* count);
const ShapeEntry* fused_shapes_f64_dcx(int* count);
const ShapeEntry* fused_shapes_bec(int* count);  // bit-sliced erasure decoder (ldpc_bec_kernels.hpp)

namespace {

using u64 = unsigned long long;

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) -- needed where the index feeds an asm immediate
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int K>
__device__ __forceinline__ uint32_t half_of(const uint32_t (&tab)[(K + 1) / 2], int k) {
    const uint32_t w = tab[k >> 1];
    return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

// Lane-contiguous LDS store without an address register: LDS[M0 + OFF + 4*lane] = v (ds_write_addtid_b32 moves one
// source dword instead of two -> half the store-path cycles of ds_write_b32; MI355X_MICROARCH.md, LDS table).
// Inline asm: the compiler does not count it in lgkmcnt; its own waits then only become more conservative
// (LDS operations of a wave retire in order), never too early.
template <int OFF>
__device__ __forceinline__ void lds_st_tid(float v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v), "n"(OFF) : "memory");
}
// (No "m0" in the clobber list: hipcc treats M0 as a RESERVED register -- "inline asm clobber list contains reserved registers: m0 ...
// may not be preserved across the asm statement" [-Winline-asm] -- i.e. the clobber is not honoured, it only adds a warning per
// instantiation.  What makes this sound is that the compiler has no M0 use of its own in these kernels, which
// tests/test_host_cpu.py::test_compiler_never_touches_m0_in_the_fused_kernels checks on the disassembly of the built library.)
__device__ __forceinline__ void lds_set_m0(uint32_t base) { asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory"); }
// the same in the middle of a run of add-TID stores (the 16-wave shape switches its base once per check phase): an SALU write of M0 needs
// one wait state before an LDS add-TID instruction reads it, and inline asm is invisible to the compiler's hazard recogniser
__device__ __forceinline__ void lds_switch_m0(uint32_t base) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base) : "memory"); }

// The few words the waves of a frame hand each other (syndrome verdicts, frame numbers, error counts) are accessed as LDS words:
// through a generic `volatile uint32_t*` the compiler emits flat_store / flat_load ... sc0 sc1 -- the aperture check of the vector
// memory pipeline, vmcnt AND lgkmcnt to wait for -- on the serial store -> barrier -> load chain that ends every sweep.
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
__device__ __forceinline__ lds_vu32* lds_word(const void* p) { return (lds_vu32*)(uintptr_t)(uint32_t)(uintptr_t)p; }  // low half of a generic LDS address = the LDS byte address

// Workgroup barrier that also drains this wave's LDS queue: the ds_write_addtid stores above are inline asm, invisible to
// the compiler's s_waitcnt insertion, so a plain __syncthreads() may reach s_barrier with such stores still in flight.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

__device__ __forceinline__ float lds_ld(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(base + byte_off);
}
// gather through a table entry: a 16-bit LDS byte offset, or (BIG: frames beyond 64 KB of LDS) a 16-bit dword index
template <bool BIG>
__device__ __forceinline__ float lds_gat(const unsigned char* base, uint32_t entry) {
    return *reinterpret_cast<const float*>(base + (BIG ? (entry << 2) : entry));
}
// gather through entry k of a packed table.  BIG: address = 4 * (16-bit half of the word) + base in ONE instruction (v_mad_u32_u16 with op_sel
// picking the half) instead of and/shift + shift-add -- the 16-wave shape keeps all its tables packed (128 VGPRs) and unpacks 54 entries per sweep
typedef __attribute__((address_space(3))) const float lds_cf32;
template <bool BIG, int K, bool MAD = BIG>
__device__ __forceinline__ float gat_tab(const unsigned char* base, const uint32_t (&tab)[(K + 1) / 2], int k) {
    if constexpr (BIG && MAD) {
        uint32_t addr;
        const uint32_t b = (uint32_t)(uintptr_t)base;
        if (k & 1) asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(tab[k >> 1]), "s"(b));
        else asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(addr) : "v"(tab[k >> 1]), "s"(b));
        return *(lds_cf32*)(uintptr_t)addr;
    } else {
        return lds_gat<BIG>(base, half_of<K>(tab, k));
    }
}
// c2v store of the BIG shape: ds_write_addtid reaches M0[15:0] + 16-bit offset only, so rows beyond that use an address
// register (lane-contiguous all the same; 4 instead of 2 store-path cycles)
// Two rows per instruction: ds_write2st64_b32 stores a at vaddr + R0*256 and b at vaddr + R1*256 (offsets in units of 64 dwords
// == one lane-contiguous row); three source dwords -> 6 store-path cycles for two rows.
template <int R0, int R1>
__device__ __forceinline__ void lds_st2_rows(uint32_t vaddr, float a, float b) {
    static_assert(R0 >= 0 && R0 < 256 && R1 >= 0 && R1 < 256, "8-bit row offsets");
    // (the same two rows as two ds_write_b32 -- 4 + 4 store-path cycles instead of 6 -- measured 12 % slower: profiles/r04_big_store_pairing.txt)
    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(vaddr), "v"(a), "v"(b), "n"(R0), "n"(R1) : "memory");
}

template <int OFF>
__device__ __forceinline__ void lds_st_row(uint32_t vaddr, float a) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(vaddr), "v"(a), "n"(OFF) : "memory");
}

// one row of 8-byte elements, lane-contiguous: ds_write_b64 with an immediate row offset.  Written as inline asm so that the
// compiler's load/store optimiser does not pair two of them into ds_write2st64_b64 (five source dwords: 13 store-path cycles for
// two rows against 2 x 6, MI355X_MICROARCH.md LDS table); like lds_st_tid the store is invisible to the compiler's s_waitcnt
// accounting, which only makes its waits more conservative -- barriers drain the queue explicitly.
template <int OFF>
__device__ __forceinline__ void lds_st64(uint32_t vaddr, double v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(vaddr), "v"(v), "n"(OFF) : "memory");
}

struct FusedArgs {
    const void* priors;             // [B,n] float (fp32 kernels) or double (fp64 min-sum kernel)
    const uint8_t* y0;
    long long B;
    int n, max_iter;
    unsigned flags;
    const uint32_t* cn_tab;
    const uint32_t* vn_tab;
    const int32_t* var_of_slot;
    const u64* cn_active;
    uint8_t* xhat;
    int32_t* iters;
    void* soft;                     // optional [B,n] (float / double as the kernel): marginals of each frame's last executed sweep (src/bpa.py:35), 0 if none
    u64* next_frame;
    int zero_row;
    int sync_off[4];                 // 1 < NW <= 4: byte offset of a padded c2v slot owned by wave w (verdict / frame hand-off)
    int msync_off[4];                // 1 < NW <= 4: byte offset of a padded MARGINAL slot owned by wave w (end-of-sweep hand-off)
    int sys_off;                     // NW = 16: byte offset of the system row (last marginal row, never written by the sweeps)
    // fused simulate (SIM kernels): BI-AWGN noise generated in the kernel, errors counted in the kernel
    const int32_t* slot_of_var;     // [n rounded up to 4] LDS dword index (marg area) of each variable
    float sim_mean, sim_sigma, sim_k;  // y = mean + sigma z ; prior = -(k y), k = 2/sigma^2   (src/biawgn.py:17-28)
    unsigned long long seed, frame0;
    unsigned stream;
    int codeword, hist_bins;
    int sim_channel;                 // CH_BIAWGN or CH_BSC
    unsigned long long bsc_thr;     // BSC: flip <=> Philox word < thr   (src/bsc.py:16)
    float bsc_llr;                  // BSC: prior = llr * (1 - 2y)        (src/bsc.py:21,25), llr > 0 here
    double sim_mean_d, sim_sigma_d, sim_k_d, bsc_llr_d;  // the same constants for the fp64 kernels (k_biawgn<double> / k_discrete<double>)
    uint32_t certain_entry;         // gather-table entry of one of the "certain" variable slots that pad short check rows (0xffffffff: no short rows)
    unsigned long long* counters;   // [4 + hist_bins] tot, wec, bec, iter_sum, histogram of sweeps
    int flush_every;                // SIM: frames a workgroup counts in 32-bit lanes before adding them to `counters`
    // several rounds per launch (erasure Monte-Carlo kernel, ldpc_simulate_rounds): `rounds` rounds of B frames each; round r covers
    // frames frame0 + r * round_stride + [0, B) and accumulates into counters + r * counter_stride
    int rounds, counter_stride;
    unsigned long long round_stride;
    // exact-in-fp32 mode: priors are rounded to multiples of 1 / grid_scale (SIM: in the kernel; decode: by the channel kernel); a frame
    // in which some |v2c|, |marginal| reaches grid_limit -- beyond it an fp32 sum of grid multiples may round -- is counted in grid_viol
    float grid_scale, grid_inv, grid_limit;
    unsigned long long* grid_viol;  // [0] frames beyond the guard since the last reset, [1 .. GRID_REDO_CAP] their global indices (redo list)
};
constexpr int GRID_REDO_CAP = 4095;

// Monte-Carlo counters of a workgroup (SIM kernels) in ONE register: lanes [0, hist_bins) hold the histogram of executed sweeps,
// lanes 60..63 tot / wec / bec / iter_sum (src/main.py:41-45) as 32-bit partial sums, added to the global 64-bit counters every
// FusedArgs::flush_every frames and at the end.  (Five 64-bit accumulators + a histogram register were spilled around the sweep
// loop and written back to scratch once per frame.)
// SIM kernels: how many words of the packed gather tables are made opaque to the optimiser at the top of every sweep.  Left alone,
// the optimiser unpacks the WHOLE 16-bit table into one address register per gather ahead of the frame loop (the decode kernels keep
// the variable map in those registers; the SIM kernels do not) and then spills what no longer fits -- launch invariants around the
// sweep loop, or table entries that are reloaded from scratch in every sweep.  An opaque word keeps its two unpacking instructions
// in the sweep instead.  Counts found by compiling each shape over a grid of values (tools/kernel_resources.py; the CPU test
// tests/test_host_cpu.py::test_simulate_kernels_do_not_spill pins the result): the smallest counts with no spilled register.
constexpr int sim_opaque_cn(int alg, int nw, int vrx_arg_) {
    const int vrx = wide_rounds(vrx_arg_);
    // two-wave irregular min-sum on the shape with pair rounds (34 instead of 40 variable-phase gathers): 4 + 4 packed words, two spilled
    // registers -- 2.84 ms per 65 536 frames at 1.0 dB against 3.03 (15 + 8, the two-width shape's setting), 2.85 (0 + 0, 2 + 2), 2.91 (6 + 6),
    // 2.95 (8 + 8); at 2.0 dB 1.317 against 1.393 (round 6, two codes, six interleaved runs)
    if (nw == 2 && pair_rounds(vrx_arg_) > 0) return alg == ALG_MSA ? 4 : 8;  // (sum-product: 8 + 8 -> 4.95 ms against 5.08 with 15 + 15, 5.00 with 12 + 12, 5.08 with 4 + 4)
    if (nw > 4) return alg == ALG_MSA ? 0 : 15;           // one frame per CU (16 waves): min-sum by the compiler's own allocation (see MAD in the kernel)
    if (vrx == 0) return alg == ALG_MSA ? 0 : 6;          // regular shapes: min-sum 0 + 2; sum-product 6 + 8 (round 5: the pair-tree rule needs
                                                          // fewer registers than prefix / suffix did -- 15 + 15 before; tools/ab_spa.sh: +5.7 %)
    return 15;                                            // irregular shapes (wide variable rounds)
}
constexpr int sim_opaque_vn(int alg, int nw, int vrx_arg_) {
    const int vrx = wide_rounds(vrx_arg_);
    if (nw == 2 && pair_rounds(vrx_arg_) > 0) return alg == ALG_MSA ? 4 : 8;
    if (nw > 4) return alg == ALG_SPA ? 15 : 0;
    if (vrx == 0) return alg == ALG_SPA ? 8 : 2;
    return alg == ALG_MSA ? 8 : 15;
}
// the guarded (exact-in-fp32) variants carry one more live register (the guard's running maximum): more table words stay packed
// (measured, round 5, tools/ab_grid2.sh / ab_grid.sh: the regular two-wave shape of n = 1200 -- whose guard no longer watches marginals --
// is fastest with the plain kernel's own 0 + 2 packed words: 3.00-3.04 ms per 65 536 frames against 3.17 (8 + 8) and 3.33 (15 + 15); the
// irregular shapes are fastest with all of them packed -- re-measured in round 6 on the shapes with pair rounds: two waves 3.33-3.43 ms for
// 0 / 4 / 8 / 12 packed words against 3.35 with all; the 16-wave shape, now with the one-instruction unpack (MAD in the kernel): 18.27-18.30
// ms per 32 768 frames with 12 + 12 packed words, 18.34-18.57 with 4 ... 10, 18.65 with all, 20.13 with all and the shift + add unpack)
constexpr int grid_opaque_cn(int nw, int vrx) { return nw > 4 ? 12 : ((nw == 2 && vrx == 0) ? 0 : 15); }
constexpr int grid_opaque_vn(int nw, int vrx) { return nw > 4 ? 12 : ((nw == 2 && vrx == 0) ? 2 : 15); }
constexpr int SIM_ACC_LANE0 = 60;  // hist_bins <= 60 (fused_simulate_supported)
__device__ __forceinline__ void sim_count(unsigned& accv, int lane, int err, int it, int hist_bins) {
    const int bin = it < hist_bins ? it : hist_bins - 1;  // no histogram: -1, no lane
    unsigned add = (lane == bin) ? 1u : 0u;
    add = lane == SIM_ACC_LANE0 ? 1u : add;
    add = lane == SIM_ACC_LANE0 + 1 ? (err > 0 ? 1u : 0u) : add;
    add = lane == SIM_ACC_LANE0 + 2 ? (unsigned)err : add;
    add = lane == SIM_ACC_LANE0 + 3 ? (unsigned)it : add;
    accv += add;
}
__device__ __forceinline__ void sim_flush(unsigned& accv, int lane, int hist_bins, unsigned long long* counters) {
    // lane id taken afresh (v_mbcnt): the counter addresses are then formed here, not at kernel entry and carried through every sweep
    unsigned zero = 0;
    asm volatile("" : "+v"(zero));  // opaque: keeps the two instructions below from being hoisted to kernel entry
    lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
    if (accv) {
        if (lane >= SIM_ACC_LANE0) atomicAdd(&counters[lane - SIM_ACC_LANE0], (unsigned long long)accv);
        else if (lane < hist_bins) atomicAdd(&counters[4 + lane], (unsigned long long)accv);
    }
    accv = 0;
}

// (the body of k_fused_bp / k_fused_bp_grid: the kernels themselves follow it)
template <int ALG, int DC, int DV, int CRW, int VRW, int NW, bool SIM, int VRXA, int DVX, bool GRID>
__device__ __forceinline__ void fused_bp_body(const FusedArgs& A) {
    static_assert(!GRID || ALG == ALG_MSA, "the exactness guard belongs to min-sum: only add / subtract / compare");
    constexpr int VRX = wide_rounds(VRXA), VR2 = pair_rounds(VRXA);  // the wave's first VRX rounds gather DVX messages, its last VR2 rounds two
    static_assert(VR2 == 0 || (VRX > 0 && DV > 2 && VRX + VR2 <= VRW), "pair rounds belong to the irregular shapes");
    // BIG: a frame takes the whole LDS of a CU (160 KB) and a 16-wave workgroup.  Table entries are dword indices, c2v stores
    // use an address register, and the LAST marginal row is a system row that no sweep writes: dwords [0,16) hand-off
    // channel A (one word per wave), [16,32) channel B, [32] frame hand-out, [33] always zero (target of missing edges).
    constexpr bool BIG = NW > 4;
    // SYS: the system row (hand-off words + zero word in the last marginal row) is also what lets SEVERAL waves share an
    // irregular frame of the small shapes (they have no always-zero row and may have no padded slot per wave)
    constexpr bool SYS = BIG || (NW > 1 && VRX > 0);
    constexpr int VNK = VRX * DVX + (VRW - VRX - VR2) * DV + VR2 * 2;  // gathers of a variable phase: wide rounds first, then narrow ones, pair rounds last
    constexpr int VN0 = VRX * DVX;                      // first gather index of the narrow rounds
    constexpr int CR = CRW * NW, VR = VRW * NW;
    constexpr int NPAD = VR * 64;
    constexpr int CNW = (CRW * DC + 1) / 2, VNW = (VNK + 1) / 2;
    constexpr int VRN = VRW - VRX;  // narrow variable rounds (DV gathers; the last VR2 of them two); they follow the VRX wide rounds
    // narrow round u: its width and the index of its first gather
    auto nar_w = [](int u) constexpr { return u < VRN - VR2 ? DV : 2; };
    auto nar_0 = [](int u) constexpr { return VN0 + (u < VRN - VR2 ? u * DV : (VRN - VR2) * DV + (u - (VRN - VR2)) * 2); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    float* lds_marg = reinterpret_cast<float*>(smem) + w * VRW * 64;  // this wave's marginal rows
    const int32_t* vslot = A.var_of_slot + w * VRW * 64;
    const u64* cn_active = A.cn_active + w * CRW;
    const int n = A.n, max_iter = A.max_iter;

    // gather addresses, resident in registers for the whole launch
    uint32_t cn_idx[CNW], vn_idx[VNW];
#pragma unroll
    for (int i = 0; i < CNW; ++i) cn_idx[i] = A.cn_tab[(w * CNW + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < VNW; ++i) vn_idx[i] = A.vn_tab[(w * VNW + i) * 64 + lane];
    // variable index of each owned slot (-1: padding): resident in registers where the budget allows, else re-read per frame
    constexpr bool VMAP_RESIDENT = !SIM && !BIG && ((NW == 1) || (ALG == ALG_MSA));
    int vmap_reg[VMAP_RESIDENT ? VRW : 1];
    if constexpr (VMAP_RESIDENT) {
#pragma unroll
        for (int q = 0; q < VRW; ++q) vmap_reg[q] = vslot[q * 64 + lane];
    }
    // (BIG: 128 VGPRs per wave do not hold both gather tables next to the messages and priors; the compiler spills part of the
    // variable-phase table and reloads it after the barrier that ends the check phase.  Re-reading the whole table from L2 every sweep,
    // issued BEFORE that barrier, removes every spill but measured slower for min-sum -- 11.3 vs 10.0 ms per 16 384 frames x 48.7 sweeps
    // at n = 10 000 -- and no different for sum-product.)
    auto opaque_tables = [&]() {  // see sim_opaque_cn
        if constexpr (SIM) {
#pragma unroll
            for (int i = 0; i < (GRID ? grid_opaque_cn(NW, VRX) : sim_opaque_cn(ALG, NW, VRXA)) && i < CNW; ++i) asm volatile("" : "+v"(cn_idx[i]));
#pragma unroll
            for (int i = 0; i < (GRID ? grid_opaque_vn(NW, VRX) : sim_opaque_vn(ALG, NW, VRXA)) && i < VNW; ++i) asm volatile("" : "+v"(vn_idx[i]));
        }
    };
    auto vmap_of = [&](int q) -> int {
        if constexpr (VMAP_RESIDENT) return vmap_reg[q]; else return vslot[q * 64 + lane];
    };
    if (A.zero_row && w == 0) reinterpret_cast<float*>(smem)[NPAD + CR * DC * 64 + lane] = 0.0f;  // the always-zero row
    // the 16-wave shape's gathers: one-instruction unpack (gat_tab) -- except in the min-sum Monte-Carlo kernel, where the compiler's own
    // choice is measurably better still: it unpacks the variable-phase addresses once, spills 19 of them and reloads 13 per sweep from its
    // private segment (13 scratch_load_dword on an otherwise idle vector-memory pipe instead of 47 unpack instructions): 17.86 ms against
    // 18.70 (one-instruction unpack, no spill) and 19.10 (shift + add unpack, no spill) per 32 768 frames, profiles/r03C_spill_or_unpack.txt.
    // The GUARDED min-sum kernel keeps most table words packed (grid_opaque_cn) and does take the one-instruction unpack: +10 % (round 6).
    constexpr bool MAD = BIG && (GRID || !(ALG == ALG_MSA && SIM));
    const bool own_last = !(SYS && w == NW - 1);  // with a system row, wave NW-1 never writes its last marginal row
    // the sign of an outgoing message is merged with ONE v_and_or_b32 (inline asm; the compiler emits v_and_b32 + v_or_b32 for the same
    // expression in this kernel): the IEEE sign bit lives in a scalar register
    const uint32_t sign_mask = 0x80000000u;
    auto sysw = [&](int i) { return lds_word(smem + A.sys_off) + i; };
    if constexpr (SYS) {
        if (threadIdx.x < 31) *sysw(33 + threadIdx.x) = 0u;  // the zero words missing edges gather (one per bank 1..31: the half-wave's free one,
                                                             // ldpc_fused.hip); published by the barrier at the top of the frame loop
    }

    // BIG: the c2v rows of the 16 waves are INTERLEAVED -- local row k = r*DC + j of wave w is row k*NW + w of the c2v area -- so that the
    // wave-dependent part of a row address is small (w * 256 B) and the large part (k * NW * 256 B) is a compile-time immediate:
    // ds_write_addtid_b32 reaches M0[15:0] + a 16-bit immediate, i.e. the first 128 KB of the 160 KB frame, with TWO bases per wave
    // (rows k < BIG_KA from M0_a = c2v area + w*256, rows BIG_KA <= k < BIG_KB from M0_b = 65 535 - (NW-1-w)*256).  Only the rows beyond
    // (k >= BIG_KB: 8 of a wave's 30, four paired stores) keep an address register.  An add-TID store costs 2.2 issue-path cycles per row
    // against 3 for the paired form (+1.2 % measured); the LDS-array counters (SQ_LDS_IDX_ACTIVE 4 562, SQ_LDS_BANK_CONFLICT 787 per
    // frame-sweep) did not move -- row stores are conflict-free either way: those 787 cycles were the GATHERS of padding positions (one
    // certain slot, one zero word for every half-wave), found with store-less / linear-gather probe builds and fixed in the table
    // builder (ldpc_fused.hip; HISTORY.md, round 6).
    constexpr int BIG_C2V = NPAD * 4;                                                      // byte offset of the c2v area
    constexpr int BIG_ROWB = NW * 256;                                                     // bytes between two local rows of a wave
    constexpr int BIG_KA = BIG ? (65535 / BIG_ROWB + 1 < CRW * DC ? 65535 / BIG_ROWB + 1 : CRW * DC) : 0;
    constexpr int BIG_MB = 65535 - (NW - 1) * 256;                                         // M0_b without its w * 256: the last wave's is 65 535 (the sum M0 + immediate + 4 * lane is what must be aligned)
    constexpr int BIG_KB = BIG ? ((65535 + BIG_MB - BIG_C2V) / BIG_ROWB + 1 < CRW * DC ? (65535 + BIG_MB - BIG_C2V) / BIG_ROWB + 1 : CRW * DC) : 0;
    static_assert(!BIG || (BIG_C2V + (NW - 1) * 256 < 65536 && BIG_C2V + BIG_KA * BIG_ROWB >= BIG_MB), "add-TID bases of the 16-wave shape");
    static_assert(!BIG || ((CRW * DC - BIG_KB) * NW < 256), "8-bit row offsets of the paired address-register stores");
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
    const uint32_t m0_c2v = BIG ? lds_base + (uint32_t)(BIG_C2V + w * 256) : lds_base + (uint32_t)(NPAD + w * CRW * DC * 64) * 4;  // this wave's c2v rows
    const uint32_t m0_c2v_b = lds_base + (uint32_t)(BIG_MB + w * 256);               // BIG: second add-TID base
    const uint32_t m0_marg = lds_base + (uint32_t)(w * VRW * 64) * 4;            // this wave's marginal rows
    const uint32_t my_sync = (uint32_t)A.sync_off[SYS ? 0 : w];
    const uint32_t my_msync = (uint32_t)A.msync_off[SYS ? 0 : w];
    const uint32_t c2v_vaddr = m0_c2v + (uint32_t)(BIG_KB * BIG_ROWB) + (uint32_t)lane * 4u;  // BIG: address register of the c2v rows beyond add-TID reach (k >= BIG_KB)
    const bool early = !(A.flags & FLAG_NO_EARLY_EXIT);
    // SIM: per-workgroup counters live in wave 0 (scalars + one histogram bin per lane), flushed once at the end
    unsigned valid = 0;  // bit q: slot (q, lane) holds a real variable
    unsigned accv = 0;  // sim_count / sim_flush
    int acc_frames = 0;
    unsigned dummy = 0;  // bit q: slot (q, lane) is the "certain" slot that pads short check rows (var_of_slot == -2)
    if constexpr (SIM || VRX > 0) {
#pragma unroll
        for (int q = 0; q < VRW; ++q) {
            valid |= (vslot[q * 64 + lane] >= 0) ? (1u << q) : 0u;
            dummy |= (vslot[q * 64 + lane] == -2) ? (1u << q) : 0u;
        }
        // one register each, opaque to the optimiser: otherwise it keeps the ten per-round masks of `valid` as ten separate
        // launch-invariant values and spills them around the sweep loop
        asm volatile("" : "+v"(valid), "+v"(dummy));
    }
    // bit r: check slot (w*CRW + r, lane) holds a real check.  A PADDED check lane gathers whatever is free for its half-wave (an
    // address a real lane reads anyway: LDS broadcast, no extra cycle) and is masked out of the syndrome here.  (Rounds 1-2 let it read
    // ONE marginal dc times instead -- even sign parity, no mask needed for even dc -- but that one address sat on a busy bank in
    // most of the dc gathers: 49 of the 67 measured bank-conflict cycles per frame-sweep of the n = 1200 plan.)
    unsigned cn_valid = 0;
    if constexpr (!BIG) {
#pragma unroll
        for (int r = 0; r < CRW; ++r) cn_valid |= ((cn_active[r] >> lane) & 1ull) ? (1u << r) : 0u;
        asm volatile("" : "+v"(cn_valid));
    }

    // verdict exchange between the NW waves of a frame: each wave publishes "my checks see an unsatisfied syndrome" in a
    // padded c2v slot it owns (nobody else ever writes it; its own garbage write precedes in program order) -> barrier ->
    // everybody reads all verdicts.  The next write to those slots happens after the following barrier.
    auto any_unsat = [&](bool mine) -> bool {
        if constexpr (NW == 1) {
            return mine;
        } else if constexpr (SYS) {
            if (lane == 0) *sysw(w) = mine ? 1u : 0u;
            wg_barrier();
            return __ballot(*sysw(lane & (NW - 1)) != 0u) != 0;
        } else {
            if (lane == 0) *lds_word(smem + my_sync) = mine ? 1u : 0u;
            wg_barrier();
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) v |= *lds_word(smem + A.sync_off[i]);
            return v != 0u;
        }
    };

    // frame error counts after the last sweep: through the pair of hand-off words the sweep loop did NOT just use for its exit
    // verdict (the other wave may still be reading that one): the marginal slots
    auto exchange_add = [&](int mine) -> int {
        if constexpr (NW == 1) {
            return mine;
        } else if constexpr (SYS) {
            constexpr int CH = 16;  // the channel the sweep loop did not just use
            if (lane == 0) *sysw(CH + w) = (uint32_t)mine;
            wg_barrier();
            int sum = lane < NW ? (int)*sysw(CH + (lane & (NW - 1))) : 0;
#pragma unroll
            for (int o = NW / 2; o; o >>= 1) sum += __shfl_xor(sum, o);
            return __builtin_amdgcn_readfirstlane(sum);
        } else {
            const uint32_t mine_off = my_msync;
            if (lane == 0) *lds_word(smem + mine_off) = (uint32_t)mine;
            wg_barrier();
            int sum = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) sum += (int)*lds_word(smem + A.msync_off[i]);
            return sum;
        }
    };

    // Frame hand-out.  One device-wide counter sustains only ~88 dequeues/us (MI355X_MICROARCH.md, row "dequeue"): 65 536
    // single-frame dequeues alone would take 0.75 ms.  The frame range is therefore cut into NSHARD contiguous shards with
    // one counter each (64 B apart); a workgroup drains its home shard (blockIdx % NSHARD -- the XCD it runs on, as
    // observed) one frame at a time and then helps with the next shards.  Frame granularity keeps the load balanced when
    // frames need different numbers of sweeps.
    constexpr int NSHARD = 8;
    const long long shard_len = (A.B + NSHARD - 1) / NSHARD;
    int shard = (int)(blockIdx.x % NSHARD), shards_left = NSHARD;
    auto next_frame = [&]() -> long long {  // wave-uniform; -1 when every shard is drained
        long long got = -1;
        while (shards_left > 0) {
            const long long base = shard * shard_len;
            const long long len = (base + shard_len <= A.B ? shard_len : A.B - base);
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
        long long fr_s = 0;
        if constexpr (NW == 1) {
            fr_s = next_frame();
        } else {
            wg_barrier();  // the verdict slots of the previous frame have been read by everybody
            if (w == 0) {
                const long long f0 = next_frame();
                if (lane == 0) *(SYS ? sysw(32) : lds_word(smem + A.sync_off[0])) = (uint32_t)(int32_t)f0;
            }
            wg_barrier();
            fr_s = (long long)(int32_t)__builtin_amdgcn_readfirstlane(*(SYS ? sysw(32) : lds_word(smem + A.sync_off[0])));
        }
        if (fr_s < 0) break;
        const u64 fr = (u64)fr_s;
        float prior[VRW];
        float c2v_old[CRW][DC];
        unsigned xb = 0;  // bit q = hard decision of variable slot (w*VRW + q, lane)
        if constexpr (SIM) {
            // channel + LLR in the kernel: the workgroup draws the frame's noise block by block (one Philox block = 4
            // consecutive variables, exactly as k_biawgn does) and drops every prior into the LDS slot of its variable
            // (the thread index enters through an opaque copy: everything derived from it -- table addresses, the first Philox
            // round -- would otherwise be hoisted out of the FRAME loop and spilled around the sweeps)
            int tid0 = (int)threadIdx.x;
            asm volatile("" : "+v"(tid0));
            for (int blk = tid0; blk * 4 < n; blk += 64 * NW) {
                const Philox4 ph = philox_word_block(A.seed, A.stream, A.frame0 + fr, (uint32_t)blk);
                const int4 sl = *reinterpret_cast<const int4*>(A.slot_of_var + blk * 4);
                const int slots[4] = {sl.x, sl.y, sl.z, sl.w};
                float pri4[4];
                if (A.sim_channel == CH_BIAWGN) {
                    float z[4];
                    box_muller<float>(ph.w[0], ph.w[1], z[0], z[1]);
                    box_muller<float>(ph.w[2], ph.w[3], z[2], z[3]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        pri4[t] = -(A.sim_k * (A.sim_mean + A.sim_sigma * z[t]));
                        if constexpr (GRID) pri4[t] = __builtin_rintf(pri4[t] * A.grid_scale) * A.grid_inv;  // both scalings are exact (powers of two)
                        if constexpr (ALG == ALG_SPA) pri4[t] *= SPA2_LOG2E;  // sum-product runs in the base-2 LLR domain (ldpc_cn.hpp)
                    }
                } else {  // CH_BSC: same integer threshold and the same LLR expression as k_discrete
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int y = A.codeword ^ ((u64)ph.w[t] < A.bsc_thr ? 1 : 0);
                        pri4[t] = (GRID ? __builtin_rintf(A.bsc_llr * A.grid_scale) * A.grid_inv : A.bsc_llr) * (float)(1 - 2 * y);
                        if constexpr (ALG == ALG_SPA) pri4[t] *= SPA2_LOG2E;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (blk * 4 + t < n) reinterpret_cast<float*>(smem)[slots[t]] = pri4[t];
            }
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                prior[q] = lds_marg[q * 64 + lane];  // padded slots: stale words, never used
                if (A.sim_channel == CH_BSC) xb |= (__float_as_uint(prior[q]) >> 31) << q;  // x_hat starts as the received word
            }
        } else {
            const float* pf = reinterpret_cast<const float*>(A.priors) + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const int v = vmap_of(q);
                prior[q] = v >= 0 ? pf[v] : 0.0f;
                if constexpr (ALG == ALG_SPA) prior[q] *= SPA2_LOG2E;  // base-2 LLR domain (ldpc_cn.hpp); the soft output is scaled back
            }
        }
        if constexpr (VRX > 0) {
            // the padding slot of short check rows is a variable known with certainty: +inf LLR (bit 0) -- it never wins a minimum, adds
            // nothing to a join and has sign 0
#pragma unroll
            for (int q = 0; q < VRW; ++q)
                if ((dummy >> q) & 1u) prior[q] = __builtin_huge_valf();
        }
#pragma unroll
        for (int r = 0; r < CRW; ++r)
#pragma unroll
            for (int j = 0; j < DC; ++j) c2v_old[r][j] = 0.0f;

        int it = 0;
        bool left_at_0 = false;
        // GRID -- the exactness guard.  Every value of the iteration is a multiple of 2^-k; fp32 adds / subtracts of such values are exact
        // while the result stays below 2^(24-k).  With P = max |prior| and C = max |c2v| of a frame: a partial sum of the variable update is
        // at most dv C, the marginal P + dv C, v2c = marg - c2v_old at most P + (dv + 1) C -- so P < L and C < L with
        // L = 2^(24-k) / (dv_max + 2) (rounded down to a power of two by the host: A.grid_limit) keep EVERY sum of every sweep exact.  The
        // priors are watched once per frame (here), the outgoing magnitudes in every check phase; marginals need no watching of their
        // own (rounds 3-4 watched them in every variable phase: 4 instructions per variable row and sweep).  That is the scheme of the
        // REGULAR shapes (GRID_PRIORS); the irregular ones keep the older one -- outgoing magnitudes and marginals both below 2^(21-k),
        // watched in every sweep: a partial sum of up to 8 messages stays below 2^(24-k) -- because a pass over the priors costs them
        // registers they do not have (measured both as a per-slot pass here and inside the noise loop: 5-9 % slower; tools/ab_grid.sh).
        constexpr bool GRID_PRIORS = GRID && VRX == 0;
        float gmax = 0.0f;  // largest |prior| / |c2v| (/ |marginal|) this lane has seen in this frame
        if constexpr (GRID_PRIORS) {
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                // (not watched: SIM's padded slots -- stale words)
                const bool watched = !SIM || ((valid >> q) & 1u);
                gmax = fmaxf(gmax, watched ? __builtin_fabsf(prior[q]) : 0.0f);
            }
        }
        if (!SIM && A.y0 != nullptr) {
            // iteration-0 test of the received hard word (src/bpa.py:20,29): park it in the marg area as -+1
            const uint8_t* yf = A.y0 + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const int v = vmap_of(q);
                const bool one = v >= 0 && yf[v] != 0;
                if (q < VRW - 1 || own_last) lds_marg[q * 64 + lane] = one ? -1.0f : 1.0f;
                xb |= one ? (1u << q) : 0u;
            }
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
            u64 unsat = 0;
#pragma unroll
            for (int r = 0; r < CRW; ++r) {
                u64 par = 0;
#pragma unroll
                for (int j = 0; j < DC; ++j) par ^= __ballot(gat_tab<BIG, CRW * DC, MAD>(smem, cn_idx, r * DC + j) < 0.0f);
                unsat |= par & cn_active[r];  // padded check lanes read arbitrary marginals: masked out
            }
            left_at_0 = early && !any_unsat(unsat != 0);
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
        }
        if (!left_at_0) {
#pragma unroll
            for (int q = 0; q < VRW; ++q) if (q < VRW - 1 || own_last) lds_marg[q * 64 + lane] = prior[q];
            if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
            // The sweep is one software-pipelined stream of LDS traffic: the gathers of check round r+1 (variable
            // group g+1) are issued before round r (group g) is computed, so a wave always has a full round of
            // ds_reads in flight while it does arithmetic.
            constexpr int VG = BIG ? 1 : ((ALG == ALG_MSA && NW == 1) ? 4 : 2);  // variable rounds per pipeline stage (register budget)
            constexpr int NVG = (VRN + VG - 1) / VG;
            for (;;) {
                if (max_iter > 0 && it >= max_iter) break;
                lds_set_m0(m0_c2v);
                opaque_tables();
                // ---------------- check phase (+ syndrome of the decisions of the previous sweep)
                uint32_t synd = 0;  // bit 31: some owned check is unsatisfied by the previous decisions
                float mg[2][DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) mg[0][j] = gat_tab<BIG, CRW * DC, MAD>(smem, cn_idx, j);
                static_for<0, CRW>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    if constexpr (r + 1 < CRW) {
#pragma unroll
                        for (int j = 0; j < DC; ++j) mg[(r + 1) & 1][j] = gat_tab<BIG, CRW * DC, MAD>(smem, cn_idx, (r + 1) * DC + j);
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this round's arithmetic
                    // Sign handling on the raw IEEE bits (bit 31), all in the vector ALU: row parity = XOR of the sign
                    // bits, extrinsic sign = parity ^ own sign.  Equivalent to the reference's comparisons
                    // ((v < 0) for the parity, (v >= 0) for the own sign, src/math_utils.py:10,38-43) because v is never
                    // -0.0 here: marginals are built as prior + ((0.0 + c_a) + c_b + c_c) (see the variable phase).
                    float v[DC], a[DC];
                    uint32_t vx = 0, mx = 0;
                    // (six v_sub_f32: forcing the pairs back into v_pk_add_f32 with an explicit 2-vector subtraction -- the compiler stopped pairing
                    // them when the messages became inline-asm outputs -- measured 1 % SLOWER on every fp32 shape, profiles/r03A_history.txt)
#pragma unroll
                    for (int j = 0; j < DC; ++j) v[j] = mg[r & 1][j] - c2v_old[r][j];
#pragma unroll
                    for (int j = 0; j < DC; ++j) a[j] = __builtin_fabsf(v[j]);
                    // GRID, rows without padding positions (regular shapes): every outgoing magnitude of a row is its first or second minimum,
                    // and the second minimum of the row is at most the larger of ANY two of its entries -- one v_max3_f32 per row bounds all six
                    // (conservative: it may set a frame aside that the per-edge watch would have kept; LLRs of thousands only occur in the
                    // diverging frames the guard exists for)
                    constexpr bool GRID_ROW_BOUND = GRID && VRX == 0 && DC >= 2;
                    if constexpr (GRID_ROW_BOUND) gmax = fmaxf(fmaxf(gmax, a[0]), a[1]);

                    // XOR of the raw words, three inputs per instruction (v_bitop3_b32, truth table 0x96)
#pragma unroll
                    for (int j = 0; j + 2 < DC; j += 3) {
                        mx ^= xor3(__float_as_uint(mg[r & 1][j]), __float_as_uint(mg[r & 1][j + 1]), __float_as_uint(mg[r & 1][j + 2]));
                        vx ^= xor3(__float_as_uint(v[j]), __float_as_uint(v[j + 1]), __float_as_uint(v[j + 2]));
                    }
#pragma unroll
                    for (int j = DC - DC % 3; j < DC; ++j) {
                        mx ^= __float_as_uint(mg[r & 1][j]);
                        vx ^= __float_as_uint(v[j]);
                    }
                    // bit 31 only is tested: the sign parity of a REAL check of this round.  (BIG: 128 registers hold no extra mask; there a
                    // padded lane reads zero words in every position -- sign bit 0 -- ldpc_fused.hip)
                    if constexpr (BIG) {
                        synd |= mx;
                    } else {
                        synd |= mx & (cn_valid << (31 - r));
                    }
                    // leave-one-out reduction of |v|: minimum (min-sum) or join of 1 - tanh(|v|/2) (sum-product, ldpc_cn.hpp)
                    float pre[DC], suf[DC];
                    float preo[ALG == ALG_MSA ? 1 : DC], sufo[ALG == ALG_MSA ? 1 : DC];  // odd parts (sum-product only)
                    if constexpr (ALG == ALG_MSA && DC == 6) {
                        // 11 minimum instructions for the six leave-one-out minima (v_min3_f32 where three inputs meet)
                        const float s3 = fminf(a[4], a[5]), s2 = fminf(fminf(a[3], a[4]), a[5]), s1 = fminf(a[2], s2);
                        const float p2 = fminf(a[0], a[1]), p3 = fminf(fminf(a[0], a[1]), a[2]);
                        pre[0] = fminf(fminf(a[1], a[2]), s2); suf[0] = pre[0];
                        pre[1] = fminf(a[0], s1);              suf[1] = pre[1];
                        pre[2] = fminf(p2, s2);                suf[2] = pre[2];
                        pre[3] = fminf(p3, s3);                suf[3] = pre[3];
                        pre[4] = fminf(fminf(p3, a[3]), a[5]); suf[4] = pre[4];
                        pre[5] = fminf(fminf(p3, a[3]), a[4]); suf[5] = pre[5];
                    } else if constexpr (ALG == ALG_MSA) {  // prefix / suffix minima, 3 DC - 6 instructions (no +inf seeds: fminf(inf, x) is not foldable)
                        static_assert(DC >= 3, "prefix / suffix network");
                        pre[1] = a[0];
#pragma unroll
                        for (int j = 2; j < DC; ++j) pre[j] = fminf(pre[j - 1], a[j - 1]);
                        suf[DC - 2] = a[DC - 1];
#pragma unroll
                        for (int j = DC - 3; j >= 0; --j) suf[j] = fminf(suf[j + 1], a[j + 1]);
                        pre[0] = suf[0];            // fminf(pre[j], suf[j]) below: the two ends are one-sided
                        suf[DC - 1] = pre[DC - 1];
                    } else if constexpr (DC == 6) {
                        // sum-product, base-2 LLR domain: the six leave-one-out magnitudes from the pair tree (ldpc_cn.hpp spa2_loo6) -> pre[]
#pragma unroll
                        for (int j = 0; j < DC; ++j) a[j] = spa2_u_of_llr(v[j]);
                        spa2_loo6(a, pre);
                    } else {
                        // sum-product: (E, O) pairs of prod (1 + u_i), prefix in (pre, preo), suffix in (suf, sufo) -- ldpc_cn.hpp
#pragma unroll
                        for (int j = 0; j < DC; ++j) a[j] = spa2_u_of_llr(v[j]);
                        pre[0] = 1.0f; preo[0] = 0.0f;
#pragma unroll
                        for (int j = 1; j < DC; ++j) {
                            pre[j] = pre[j - 1]; preo[j] = preo[j - 1];
                            spa_eo_push(pre[j], preo[j], a[j - 1]);
                        }
                        suf[DC - 1] = 1.0f; sufo[DC - 1] = 0.0f;
#pragma unroll
                        for (int j = DC - 2; j >= 0; --j) {
                            suf[j] = suf[j + 1]; sufo[j] = sufo[j + 1];
                            spa_eo_push(suf[j], sufo[j], a[j + 1]);
                        }
                    }
                    static_for<0, DC>([&](auto J_) {
                        constexpr int j = decltype(J_)::value;
                        float mag;
                        if constexpr (ALG == ALG_MSA) mag = fminf(pre[j], suf[j]);
                        else if constexpr (DC == 6) mag = pre[j];
                        else mag = spa2_llr_of_eo(spa2_join(spa_f2{pre[j], preo[j]}, spa_f2{suf[j], sufo[j]}));
                        // GRID, rows that may hold padding positions: the outgoing magnitudes themselves are watched (the incoming |v2c|
                        // cannot be: a short row's padding position is +inf by construction)
                        if constexpr (GRID && !GRID_ROW_BOUND) gmax = fmaxf(gmax, mag);
                        float c;  // mag | ((vx ^ v[j]) & sign bit)
                        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(c) : "v"(vx ^ __float_as_uint(v[j])), "s"(sign_mask), "v"(mag));
                        c2v_old[r][j] = c;
                        constexpr int k = r * DC + j;
                        if constexpr (!BIG) {
                            lds_st_tid<k * 256>(c);
                        } else if constexpr (k < BIG_KA) {
                            lds_st_tid<(k < BIG_KA ? k : 0) * BIG_ROWB>(c);
                        } else if constexpr (k < BIG_KB) {
                            if constexpr (k == BIG_KA) lds_switch_m0(m0_c2v_b);
                            lds_st_tid<(k >= BIG_KA && k < BIG_KB ? BIG_C2V + k * BIG_ROWB - BIG_MB : 0)>(c);
                        }
                    });
                    if constexpr (BIG) {
                        // rows of this check row beyond add-TID reach: address register, two rows per instruction where two are left
                        constexpr int k0 = r * DC < BIG_KB ? BIG_KB : r * DC, k1 = (r + 1) * DC;  // [k0, k1)
                        if constexpr (k0 < k1) {
                            constexpr int np = (k1 - k0) / 2;
                            static_for<0, np>([&](auto P_) {
                                constexpr int ka = k0 + 2 * decltype(P_)::value;
                                lds_st2_rows<(ka - BIG_KB) * NW, (ka + 1 - BIG_KB) * NW>(c2v_vaddr, c2v_old[r][ka - r * DC], c2v_old[r][ka + 1 - r * DC]);
                            });
                            if constexpr ((k1 - k0) % 2 == 1) lds_st_row<(k1 - 1 - BIG_KB) * BIG_ROWB>(c2v_vaddr, c2v_old[r][DC - 1]);
                        }
                    }
                });
                const bool unsat = any_unsat(__ballot((synd & 0x80000000u) != 0u) != 0);  // NW > 1: contains the barrier
                // it == 0: only the BSC checks the received word itself (src/bpa.py:20,29); in SIM mode marg holds +-llr there
                if (early && (it > 0 || (SIM && A.sim_channel == CH_BSC)) && !unsat) break;
                if constexpr (NW == 1) __builtin_amdgcn_wave_barrier();
                // ---------------- variable phase
                lds_set_m0(m0_marg);
                xb = 0;
                unsigned xr = 0;  // min-sum: the decision bits shifted in row after row (one v_alignbit_b32 each), bit-reversed once per sweep
                // one variable: ordered sum from +0.0 (as scipy; keeps -0.0 out of the marginals), prior last, decision bit
                auto finish_var = [&](auto Q_, float s) {
                    constexpr int q = decltype(Q_)::value;
                    const float m1 = prior[q] + s;
                    if constexpr (GRID && !GRID_PRIORS) {
                        // (not watched: SIM's padded slots -- stale words -- and the "certain" slot that pads short check rows: +inf by design)
                        const bool watched = (!SIM || ((valid >> q) & 1u)) && !(VRX > 0 && ((dummy >> q) & 1u));
                        gmax = fmaxf(gmax, watched ? __builtin_fabsf(m1) : 0.0f);
                    }
                    if (q < VRW - 1 || own_last) lds_st_tid<q * 256>(m1);
                    // decision: (m1 < 0); for min-sum the sign bit itself (m1 is never -0.0 and never NaN for finite priors)
                    if constexpr (ALG == ALG_MSA) xr = __builtin_amdgcn_alignbit(xr, __float_as_uint(m1), 31);  // (xr << 1) | sign: rows come in ascending q
                    else xb |= (m1 < 0.0f) ? (1u << q) : 0u;
                };
                if constexpr (VRX > 0) {  // wide rounds of irregular codes: DVX gathers per variable, one round per stage
                    float cw[2][DVX];
#pragma unroll
                    for (int j = 0; j < DVX; ++j) cw[0][j] = gat_tab<BIG, VNK, MAD>(smem, vn_idx, j);
                    static_for<0, VRX>([&](auto Q_) {
                        constexpr int q = decltype(Q_)::value;
                        if constexpr (q + 1 < VRX) {
#pragma unroll
                            for (int j = 0; j < DVX; ++j) cw[(q + 1) & 1][j] = gat_tab<BIG, VNK, MAD>(smem, vn_idx, (q + 1) * DVX + j);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float sw = 0.0f + cw[q & 1][0];
#pragma unroll
                        for (int j = 1; j < DVX; ++j) sw += cw[q & 1][j];
                        finish_var(Q_, sw);
                    });
                }
                float cv[2][VG][DV];
#pragma unroll
                for (int u = 0; u < VG; ++u)
#pragma unroll
                    for (int j = 0; j < DV; ++j)
                        if (u < VRN && j < nar_w(u)) cv[0][u][j] = gat_tab<BIG, VNK, MAD>(smem, vn_idx, nar_0(u) + j);
                static_for<0, NVG>([&](auto G_) {
                    constexpr int g = decltype(G_)::value;
                    if constexpr (g + 1 < NVG) {
#pragma unroll
                        for (int u = 0; u < VG; ++u)
#pragma unroll
                            for (int j = 0; j < DV; ++j)
                                if ((g + 1) * VG + u < VRN && j < nar_w((g + 1) * VG + u))
                                    cv[(g + 1) & 1][u][j] = gat_tab<BIG, VNK, MAD>(smem, vn_idx, nar_0((g + 1) * VG + u) + j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, VG>([&](auto U_) {
                        constexpr int u = decltype(U_)::value;
                        if constexpr (g * VG + u < VRN) {
                            float sn = 0.0f + cv[g & 1][u][0];
#pragma unroll
                            for (int j = 1; j < DV; ++j)
                                if (j < nar_w(g * VG + u)) sn += cv[g & 1][u][j];
                            finish_var(std::integral_constant<int, VRX + g * VG + u>{}, sn);
                        }
                    });
                });
                if constexpr (ALG == ALG_MSA) xb = __brev(xr) >> (32 - VRW);  // bit q = row q again
                if constexpr (NW > 1) wg_barrier(); else __builtin_amdgcn_wave_barrier();
                ++it;
            }
        }
        // GRID: beyond grid_limit an fp32 sum of grid multiples may round -- the frame is no longer KNOWN to equal the fp64 computation.
        // Such a frame (min-sum messages of a frame caught in a trapping set grow geometrically: about one frame in 10^4 at 2 dB) is
        // not counted / is marked, and its index goes onto the decoder's redo list: the host decodes it again in fp64.
        int wave_viol = 0;
        if constexpr (GRID) wave_viol = __ballot(gmax >= A.grid_limit) != 0ull ? 1 : 0;
        auto grid_redo = [&]() {
            if (w == 0 && lane == 0) {
                const u64 idx = atomicAdd(A.grid_viol, 1ull);
                if (idx < (u64)GRID_REDO_CAP) A.grid_viol[1 + idx] = A.frame0 + fr;
            }
        };
        if constexpr (SIM) {
            // errors against the all-`codeword` word (src/main.py:41-45), counted from the decision bits
            const unsigned wrong = (A.codeword ? ~xb : xb) & valid;
            int err = 0;
#pragma unroll
            for (int q = 0; q < VRW; ++q) err += __popcll(__ballot((wrong >> q) & 1u));
            err = exchange_add(err + (wave_viol << 20));  // (err <= n < 2^20: the waves' guard verdicts ride on the same hand-off)
            err = __builtin_amdgcn_readfirstlane(err);  // wave-uniform: keep the accumulators in scalar registers
            if (GRID && (err >> 20) != 0) grid_redo();
            else sim_count(accv, lane, err, it, A.hist_bins);
            if (++acc_frames >= A.flush_every) {
                if (w == 0) sim_flush(accv, lane, A.hist_bins, A.counters);
                accv = 0;
                acc_frames = 0;
            }
        } else {
            int it_out = it;
            if constexpr (GRID) {
                if (__builtin_amdgcn_readfirstlane(exchange_add(wave_viol)) != 0) {
                    grid_redo();
                    it_out = -1 - it;  // marked: decisions of this frame are not known to be the fp64 reference's
                }
            }
            if (w == 0 && lane == 0) A.iters[fr] = it_out;
            uint8_t* xf = A.xhat + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const int v = vmap_of(q);
                if (v >= 0) xf[v] = (uint8_t)((xb >> q) & 1u);
            }
            // soft output: the marginal rows this wave wrote in its last variable phase are still in the LDS (the check phase
            // that found the syndrome satisfied, or the sweep cap, does not touch them)
            if (A.soft != nullptr) {
                float* sf = reinterpret_cast<float*>(A.soft) + fr * n;
#pragma unroll
                for (int q = 0; q < VRW; ++q) {
                    const int v = vmap_of(q);
                    if (v >= 0) sf[v] = it > 0 ? lds_marg[q * 64 + lane] * (ALG == ALG_SPA ? SPA2_LN2 : 1.0f) : 0.0f;
                }
            }
        }
    }
    if constexpr (SIM) {
        if (w == 0) sim_flush(accv, lane, A.hist_bins, A.counters);
    }
}

#define LDPC_FUSED_BP_BOUNDS __launch_bounds__(64 * NW, NW == 1 ? (CRW <= 4 ? 4 : 2) : ((NW == 4 || DVX > 8 || (NW == 2 && DC >= 7)) ? 3 : 4))
template <int ALG, int DC, int DV, int CRW, int VRW, int NW, bool SIM, int VRX, int DVX>  // (VRX: wide rounds + 16 * pair rounds, see ShapeEntry)
__global__ LDPC_FUSED_BP_BOUNDS void k_fused_bp(const FusedArgs A) {
    fused_bp_body<ALG, DC, DV, CRW, VRW, NW, SIM, VRX, DVX, false>(A);
}
// exact-in-fp32 variant (LDPC_FLAG_PRIOR_GRID): fp32 min-sum with priors on a 2^-k grid and the exactness guard compiled in
template <int DC, int DV, int CRW, int VRW, int NW, bool SIM, int VRX, int DVX>
__global__ LDPC_FUSED_BP_BOUNDS void k_fused_bp_grid(const FusedArgs A) {
    fused_bp_body<ALG_MSA, DC, DV, CRW, VRW, NW, SIM, VRX, DVX, true>(A);
}


// ---------------------------------------------------------------------------------------------------------------------
// fp64 belief propagation on the LDS: the reference's own arithmetic (src/bpa.py:17-63 computes in float64).
//   ALG_MSA  min-sum (src/bpa.py:86-102): only add/sub/compare -> hard decisions and iteration counts bit-identical to the
//            reference on identical priors -- at LDS speed instead of HBM speed.
//   ALG_SPA  sum-product, the reference formula verbatim (src/bpa.py:66-75, src/math_utils.py:47-60: tanh, exp-sum-log product,
//            divide, atanh, +-1 -> +-inf; inf - inf -> NaN -> decision 0) through cn_spa<double> of ldpc_cn.hpp, i.e. the very
//            same device code as the streaming kernel: bit-identical to it.  The row sum of log|tanh| is order dependent, so
//            the plan of an fp64 sum-product decoder keeps every check's edges in their canonical (ascending variable) order.
// Check degrees DC in 4..8, NW waves per frame with the system-row hand-off protocol of the big fp32 shapes (upper half of the last marginal row
// reserved: dwords [0,16) verdict channel A, [16,32) channel B, [32] frame hand-out, [34,36) an always-zero double).  Same tables
// and layout plan as the fp32 kernels with 8-byte elements; gathers are ds_read_b64 (2 LDS cycles, as b32), stores ds_write_b64.
// NW = 4 (min-sum, n = 1200): the frame keeps only the check rows a code can fill (fused_check_rows), the last wave runs fewer rows, the
// register budget is 128 (tables stay packed, see the opaque words at the top of the sweep).
// SIM: channel + LLR in the kernel (Philox noise, the inline functions of the stand-alone channel kernels: bit-identical
// priors) and error counting in the kernel -- priors and decisions never exist in HBM.
template <int ALG, int DC, int DV, int CRW, int VRW, int NW, bool SIM, int VRXA, int DVX>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 4 : 2) void k_fused_f64(const FusedArgs A) {
    static_assert(ALG == ALG_MSA || ALG == ALG_SPA, "LLR decoders");
    constexpr int VRX = wide_rounds(VRXA), VR2 = pair_rounds(VRXA);  // as in fused_bp_body
    static_assert(VR2 == 0 || (VRX > 0 && DV > 2 && VRX + VR2 <= VRW), "pair rounds belong to the irregular shapes");
    constexpr int VR = VRW * NW, NPAD = VR * 64;
    constexpr int CRT = fused_check_rows(8, DC, DV, CRW, VRW, NW, VRX);  // check rows of the frame; < CRW * NW: the last waves run fewer rows
    constexpr bool RAGGED = CRT != CRW * NW;
    constexpr int VNK = VRX * DVX + (VRW - VRX - VR2) * DV + VR2 * 2;  // gathers of a wave's variable phase: VRX wide rounds (irregular codes) first, pair rounds last
    constexpr int VN0 = VRX * DVX, VRN = VRW - VRX;
    auto nar_w = [](int u) constexpr { return u < VRN - VR2 ? DV : 2; };
    auto nar_0 = [](int u) constexpr { return VN0 + (u < VRN - VR2 ? u * DV : (VRN - VR2) * DV + (u - (VRN - VR2)) * 2); };
    constexpr int CNW = (CRW * DC + 1) / 2, VNW = (VNK + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    double* const lds = reinterpret_cast<double*>(smem);
    double* const my_marg = lds + w * VRW * 64 + lane;                  // row q of this wave at my_marg[q * 64]
    double* const my_c2v = lds + NPAD + w * CRW * DC * 64 + lane;       // message (r, j) of this wave at my_c2v[(r * DC + j) * 64]
    const uint32_t c2v_vaddr = (uint32_t)(uintptr_t)my_c2v, marg_vaddr = (uint32_t)(uintptr_t)my_marg;  // the same as LDS byte addresses
    const int32_t* vslot = A.var_of_slot + w * VRW * 64;
    const u64* cn_active = A.cn_active + w * CRW;
    const int n = A.n, max_iter = A.max_iter;
    const bool early = !(A.flags & FLAG_NO_EARLY_EXIT);
    const bool own_last = !(NW > 1 && w == NW - 1) || lane < 32;  // the system words live in the upper half of the last marginal row
    auto sysw = [&](int i) { return lds_word(smem + A.sys_off) + i; };
    // table entries: 16-bit byte offsets, or -- frames beyond 64 KB of LDS (WIDE) -- 16-bit indices of 8-byte elements
    constexpr bool WIDE = (size_t)(VR * 64 + CRT * DC * 64) * 8 > 65536;
    const int my_rows = RAGGED ? (CRT - w * CRW < 0 ? 0 : (CRT - w * CRW < CRW ? CRT - w * CRW : CRW)) : CRW;  // wave-uniform
    auto gat = [&](uint32_t entry) { return *reinterpret_cast<const double*>(smem + (WIDE ? (entry << 3) : entry)); };
    // system row (32 doubles): words 0..NW-1 sweep verdicts, SYS_ERR.. error counts, SYS_TICKET the frame hand-off, then always-zero doubles for
    // missing edges to gather.  Regular shapes: one (double 17, bytes 136..143).  Irregular shapes (VRX > 0) pack the words and keep doubles
    // 9..31 zero: one word per bank, every half-wave gathers the one on the bank its real lanes leave free (ldpc_fused.hip)
    constexpr int SYS_ERR = VRX > 0 ? 8 : 16, SYS_TICKET = VRX > 0 ? 16 : 32;
    static_assert(NW <= 8, "system-row words of k_fused_f64");
    if constexpr (VRX > 0) {
        if (threadIdx.x < 46) *sysw(18 + threadIdx.x) = 0u;
    } else if (threadIdx.x == 0) {
        *sysw(34) = 0u;
        *sysw(35) = 0u;
    }

    uint32_t cn_idx[CNW], vn_idx[VNW];
#pragma unroll
    for (int i = 0; i < CNW; ++i) cn_idx[i] = A.cn_tab[(w * CNW + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < VNW; ++i) vn_idx[i] = A.vn_tab[(w * VNW + i) * 64 + lane];
    int vmap[VRW];  // 128-VGPR shape: not kept, re-read from the (L1-resident) table where a frame starts and ends
    unsigned valid = 0;  // bit q: slot (q, lane) holds a real variable
    unsigned dummy = 0;  // bit q: the "certain" slot that pads short check rows (var_of_slot == -2): prior +inf
#pragma unroll
    for (int q = 0; q < VRW; ++q) {
        vmap[q] = vslot[q * 64 + lane];
        valid |= vmap[q] >= 0 ? (1u << q) : 0u;
        dummy |= vmap[q] == -2 ? (1u << q) : 0u;
    }
    auto vmap_at = [&](int q) -> int { if constexpr (NW == 4) return vslot[q * 64 + lane]; else return vmap[q]; };
    if constexpr (NW == 4) asm volatile("" : "+v"(valid), "+v"(dummy));  // 128 VGPRs: one word each, not one register per tested bit
    unsigned cn_valid = 0;  // bit r: check slot (w*CRW + r, lane) holds a real check (padded lanes gather anything, see k_fused_bp)
#pragma unroll
    for (int r = 0; r < CRW; ++r) cn_valid |= ((cn_active[r] >> lane) & 1ull) ? (1u << r) : 0u;
    asm volatile("" : "+v"(cn_valid));
    // sum-product: positions of short check rows that read the certain slot (bit r*DC+j).  Their message must stay 0: the certain
    // marginal is +inf, and the verbatim rule would feed inf - (+-inf) = NaN back into the row once the other edges saturate
    // (upstream has no such edge: tanh(inf/2) = 1 contributes log 1 = 0 to the row sum and leaves the parity alone).
    u64 padpos = 0;
    if constexpr (ALG == ALG_SPA) {
        if (A.certain_entry != 0xffffffffu) {  // (there is a certain slot on every bank the frame has room for: var_of_slot == -2 marks them)
#pragma unroll
            for (int k = 0; k < CRW * DC; ++k) {
                const uint32_t e = half_of<CRW * DC>(cn_idx, k);
                padpos |= A.var_of_slot[WIDE ? e : (e >> 3)] == -2 ? (1ull << k) : 0ull;
            }
        }
    }
    // counting mode (SIM, or A.counters != null): the Monte-Carlo counters of main.test (src/main.py:41-45) are accumulated here instead
    // of writing decisions and iteration counts out -- per-workgroup sums in wave 0, one histogram bin per lane, flushed once
    const bool counting = SIM || A.counters != nullptr;
    unsigned accv = 0;  // sim_count / sim_flush
    int acc_frames = 0;

    auto any_unsat = [&](bool mine) -> bool {  // contains the barrier that separates the phases
        if constexpr (NW == 1) {
            __builtin_amdgcn_wave_barrier();
            return mine;
        } else {
            if (lane == 0) *sysw(w) = mine ? 1u : 0u;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            return __ballot(*sysw(lane & (NW - 1)) != 0u) != 0;
        }
    };
    // the same exchange in two halves, so that the first gathers of the variable phase can be issued BETWEEN the barrier and the use of the
    // verdict (they read messages every wave has written by then; LDS returns in order, so waiting for the verdict word does not wait for them;
    // on an exit they are simply dropped): one LDS round trip off the serial path of every sweep
    auto post_verdict = [&](bool mine) {
        if (lane == 0) *sysw(w) = mine ? 1u : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto load_verdict = [&]() -> uint32_t { return *sysw(lane & (NW - 1)); };
    auto phase_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the inline-asm row stores are invisible to the compiler's own waits
        if constexpr (NW > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    };

    constexpr int NSHARD = 8;
    const long long shard_len = (A.B + NSHARD - 1) / NSHARD;
    int shard = (int)(blockIdx.x % NSHARD), shards_left = NSHARD;
    auto next_frame = [&]() -> long long {
        long long got = -1;
        while (shards_left > 0) {
            const long long base = shard * shard_len;
            const long long len = (base + shard_len <= A.B ? shard_len : A.B - base);
            u64 t = 0;
            if (lane == 0) t = atomicAdd(A.next_frame + shard * 8, 1ull);
            const long long k = (long long)(((u64)__builtin_amdgcn_readfirstlane((unsigned)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)t));
            if (k < len) { got = base + k; break; }
            shard = (shard + 1) % NSHARD;
            --shards_left;
        }
        return got;
    };
    const double* priors = reinterpret_cast<const double*>(A.priors);
    for (;;) {
        long long fr_s;
        if constexpr (NW == 1) {
            fr_s = next_frame();
        } else {
            __syncthreads();
            if (w == 0) {
                const long long f0 = next_frame();
                if (lane == 0) *sysw(SYS_TICKET) = (uint32_t)(int32_t)f0;
            }
            __syncthreads();
            fr_s = (long long)(int32_t)__builtin_amdgcn_readfirstlane(*sysw(SYS_TICKET));
        }
        if (fr_s < 0) break;
        const u64 fr = (u64)fr_s;
        double prior[VRW], c2v_old[CRW][DC];
        unsigned xb = 0;
        if constexpr (SIM) {
            // channel + LLR in the kernel: one Philox block = 4 consecutive variables, exactly as k_biawgn<double> / k_discrete<double>
            // do (same inline functions, same fp64 expressions -> bit-identical priors); every prior goes to the LDS slot of its variable
            int blk0 = threadIdx.x;
            if constexpr (NW == 4) asm volatile("" : "+v"(blk0));  // 128 VGPRs: the per-lane addresses of this loop are recomputed per frame, not kept (spilled) across the sweeps
            for (int blk = blk0; blk * 4 < n; blk += 64 * NW) {
                const Philox4 ph = philox_word_block(A.seed, A.stream, A.frame0 + fr, (uint32_t)blk);
                const int4 sl = *reinterpret_cast<const int4*>(A.slot_of_var + blk * 4);
                const int slots[4] = {sl.x, sl.y, sl.z, sl.w};
                double pri4[4];
                if (A.sim_channel == CH_BIAWGN) {
                    double z[4];
                    box_muller<double>(ph.w[0], ph.w[1], z[0], z[1]);
                    box_muller<double>(ph.w[2], ph.w[3], z[2], z[3]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) pri4[t] = -(A.sim_k_d * (A.sim_mean_d + A.sim_sigma_d * z[t]));
                } else {  // BSC: same integer threshold and the same LLR expression as k_discrete
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int y = A.codeword ^ ((u64)ph.w[t] < A.bsc_thr ? 1 : 0);
                        pri4[t] = A.bsc_llr_d * (double)(1 - 2 * y);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (blk * 4 + t < n) lds[slots[t]] = pri4[t];
            }
            phase_barrier();
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                prior[q] = ((dummy >> q) & 1u) ? __builtin_huge_val() : ((q < VRW - 1 || own_last) ? my_marg[q * 64] : 0.0);  // padded slots: stale words, never used
                if (A.sim_channel == CH_BSC) xb |= ((uint32_t)__double2hiint(prior[q]) >> 31) << q;  // x_hat starts as the received word
            }
        } else {
            const double* pf = priors + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q) prior[q] = vmap_at(q) >= 0 ? pf[vmap_at(q)] : (((dummy >> q) & 1u) ? __builtin_huge_val() : 0.0);
        }
#pragma unroll
        for (int r = 0; r < CRW; ++r)
#pragma unroll
            for (int j = 0; j < DC; ++j) c2v_old[r][j] = 0.0;
        int it = 0;
        bool left_at_0 = false;
        if (!SIM && A.y0 != nullptr) {  // iteration-0 test of the received hard word (src/bpa.py:20,29)
            const uint8_t* yf = A.y0 + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q) {
                const bool one = vmap_at(q) >= 0 && yf[vmap_at(q)] != 0;
                if (q < VRW - 1 || own_last) my_marg[q * 64] = one ? -1.0 : 1.0;
                xb |= one ? (1u << q) : 0u;
            }
            phase_barrier();
            u64 unsat = 0;
#pragma unroll
            for (int r = 0; r < CRW; ++r) {
                u64 par = 0;
#pragma unroll
                for (int j = 0; j < DC; ++j) par ^= __ballot(gat(half_of<CRW * DC>(cn_idx, r * DC + j)) < 0.0);
                unsat |= par & cn_active[r];  // padded check lanes read arbitrary marginals: masked out
            }
            left_at_0 = early && !any_unsat(unsat != 0);
            phase_barrier();
        }
        if (!left_at_0) {
            if constexpr (!SIM) {
#pragma unroll
                for (int q = 0; q < VRW; ++q)
                    if (q < VRW - 1 || own_last) my_marg[q * 64] = prior[q];
            } else if constexpr (VRX > 0) {
                if (dummy) {
#pragma unroll
                    for (int q = 0; q < VRW; ++q)
                        if ((dummy >> q) & 1u) my_marg[q * 64] = prior[q];  // the certain slot: +inf
                }
            }
            phase_barrier();
            for (;;) {
                if (max_iter > 0 && it >= max_iter) break;
                if constexpr (NW == 4) {  // 128 VGPRs: the packed tables stay packed (unpacked once per use); hoisted out of the loop, the
                                          // unpacked addresses would be spilled and re-read from scratch in every sweep
                    // how many table words stay packed: every check-phase word, six of the eight variable-phase words -- the most the 128 registers
                    // take unpacked without a spill (round 5, tools/ab_sim.sh c2_f64: 5.48-5.52 ms per 65 536 frames against 5.53-5.56 with all of
                    // them packed; nothing packed: 5.45-5.48 with 17 spilled registers)
#pragma unroll
                    for (int i = 0; i < CNW; ++i) asm volatile("" : "+v"(cn_idx[i]));
#pragma unroll
                    for (int i = 0; i < VNW && i < 6; ++i) asm volatile("" : "+v"(vn_idx[i]));
                }
                // ---- check phase (+ syndrome of the previous decisions: sign of the gathered marginals)
                uint32_t synd = 0;   // min-sum: bit 31 = some owned check is unsatisfied (XOR of IEEE sign bits)
                u64 synd_mask = 0;   // sum-product: lanes whose check is unsatisfied ((marginal < 0) compares: NaN counts as bit 0)
                auto check_rows = [&](auto NR_) {
                constexpr int NR = decltype(NR_)::value;
                double mg[2][DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) mg[0][j] = gat(half_of<CRW * DC>(cn_idx, j));
                static_for<0, NR>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    constexpr int cur = r & 1, nxt = (r + 1) & 1;
                    if constexpr (r + 1 < NR) {  // also for a row this wave does not have: its table entries are 0, one broadcast read (skipping them
                                                 // behind a wave-uniform branch -- 12 of 72 gathers per frame-sweep -- measured 3 % SLOWER, round 4)
#pragma unroll
                        for (int j = 0; j < DC; ++j) mg[nxt][j] = gat(half_of<CRW * DC>(cn_idx, (r + 1) * DC + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (RAGGED && r >= my_rows) return;  // (scalar branch) no branch between a row's gathers and the waits that count them
                    double v[DC];
#pragma unroll
                    for (int j = 0; j < DC; ++j) v[j] = mg[cur][j] - c2v_old[r][j];
                    if constexpr (ALG == ALG_MSA) {
                        double a[DC];
                        uint32_t vx = 0, mx = 0;  // XOR of the sign words (high dwords); v is never -0.0 (marginals start from +0.0 sums)
#pragma unroll
                        for (int j = 0; j < DC; ++j) a[j] = __builtin_fabs(v[j]);
#pragma unroll
                        for (int j = 0; j + 2 < DC; j += 3) {  // three inputs per instruction (v_bitop3_b32, truth table 0x96)
                            vx ^= xor3((uint32_t)__double2hiint(v[j]), (uint32_t)__double2hiint(v[j + 1]), (uint32_t)__double2hiint(v[j + 2]));
                            mx ^= xor3((uint32_t)__double2hiint(mg[cur][j]), (uint32_t)__double2hiint(mg[cur][j + 1]), (uint32_t)__double2hiint(mg[cur][j + 2]));
                        }
#pragma unroll
                        for (int j = DC - DC % 3; j < DC; ++j) {
                            vx ^= (uint32_t)__double2hiint(v[j]);
                            mx ^= (uint32_t)__double2hiint(mg[cur][j]);
                        }
                        synd |= mx & (cn_valid << (31 - r));  // bit 31 only is tested: the sign parity of a REAL check of this round
                        double mag[DC];
                        if constexpr (DC == 6) {  // 12 two-input minima for the six leave-one-out minima (prefix p, suffix s; no v_min3_f64 exists)
                            const double p2 = fmin(a[0], a[1]), p3 = fmin(p2, a[2]), p4 = fmin(p3, a[3]);
                            const double s2 = fmin(a[4], a[5]), s3 = fmin(a[3], s2), s4 = fmin(a[2], s3);
                            mag[0] = fmin(a[1], s4);
                            mag[1] = fmin(a[0], s4);
                            mag[2] = fmin(p2, s3);
                            mag[3] = fmin(p3, s2);
                            mag[4] = fmin(p4, a[5]);
                            mag[5] = fmin(p4, a[4]);
                        } else {  // prefix / suffix minima: the leave-one-out minimum == "second minimum at the first arg-min, first elsewhere"
                            static_assert(DC >= 3, "prefix / suffix network");
                            double pre[DC], suf[DC];  // 3 DC - 6 instructions (no +inf seeds: fmin(inf, x) is not foldable)
                            pre[1] = a[0];
#pragma unroll
                            for (int j = 2; j < DC; ++j) pre[j] = fmin(pre[j - 1], a[j - 1]);
                            suf[DC - 2] = a[DC - 1];
#pragma unroll
                            for (int j = DC - 3; j >= 0; --j) suf[j] = fmin(suf[j + 1], a[j + 1]);
                            mag[0] = suf[0];
                            mag[DC - 1] = pre[DC - 1];
#pragma unroll
                            for (int j = 1; j < DC - 1; ++j) mag[j] = fmin(pre[j], suf[j]);
                        }
#pragma unroll
                        for (int j = 0; j < DC; ++j) {
                            const uint32_t sgn = (vx ^ (uint32_t)__double2hiint(v[j])) & 0x80000000u;
                            const double c = __hiloint2double((int)((uint32_t)__double2hiint(mag[j]) | sgn), __double2loint(mag[j]));
                            c2v_old[r][j] = c;
                        }
                        static_for<0, DC>([&](auto J_) {
                            constexpr int j = decltype(J_)::value;
                            lds_st64<(r * DC + j) * 512>(c2v_vaddr, c2v_old[r][j]);
                        });
                    } else {
                        u64 par = 0;
#pragma unroll
                        for (int j = 0; j < DC; ++j) par ^= __ballot(mg[cur][j] < 0.0);
                        synd_mask |= par & cn_active[r];
                        cn_spa<DC>(v, DC);  // the streaming kernel's rule, edges in canonical order (see the plan)
#pragma unroll
                        for (int j = 0; j < DC; ++j) c2v_old[r][j] = ((padpos >> (r * DC + j)) & 1ull) ? 0.0 : v[j];
                        static_for<0, DC>([&](auto J_) {
                            constexpr int j = decltype(J_)::value;
                            lds_st64<(r * DC + j) * 512>(c2v_vaddr, c2v_old[r][j]);
                        });
                    }
                });
                };
                check_rows(std::integral_constant<int, CRW>{});
                // narrow variable rounds, VG of them per pipeline stage (see the variable phase below)
                constexpr int VG = ALG == ALG_MSA ? 3 : 1;
                constexpr int NVG = (VRN + VG - 1) / VG;
                constexpr bool EARLY_GATHERS = NW > 1 && VRX == 0 && ALG == ALG_MSA;
                double cv[2][VG][DV];
                bool unsat;
                const bool mine_unsat = ALG == ALG_MSA ? (__ballot((synd & 0x80000000u) != 0u) != 0) : (synd_mask != 0);
                if constexpr (EARLY_GATHERS) {
                    post_verdict(mine_unsat);
                    const uint32_t vw = load_verdict();
#pragma unroll
                    for (int u = 0; u < VG; ++u)
#pragma unroll
                        for (int j = 0; j < DV; ++j)
                            if (u < VRN && j < nar_w(u)) cv[0][u][j] = gat(half_of<VNK>(vn_idx, nar_0(u) + j));
                    __builtin_amdgcn_sched_barrier(0);
                    unsat = __ballot(vw != 0u) != 0;
                } else {
                    unsat = any_unsat(mine_unsat);
                }
                // it == 0: only the BSC checks the received word itself (src/bpa.py:20,29); in SIM mode marg holds +-llr there
                if (early && (it > 0 || (SIM && A.sim_channel == CH_BSC)) && !unsat) break;
                // ---- variable phase: ordered sum from +0.0 (scipy COO), prior last, decision bit
                xb = 0;
                unsigned xr = 0;  // min-sum: decision bits shifted in row after row, bit-reversed once per sweep (as in k_fused_bp)
                auto finish_var = [&](auto Q_, double sn) {
                    constexpr int q = decltype(Q_)::value;
                    const double m1 = prior[q] + sn;
                    if (q < VRW - 1 || own_last) lds_st64<q * 512>(marg_vaddr, m1);
                    // (m1 < 0).  Min-sum: the sign bit itself -- m1 is never -0.0 (sums start from +0.0) nor NaN for finite priors
                    if constexpr (ALG == ALG_MSA) xr = __builtin_amdgcn_alignbit(xr, (uint32_t)__double2hiint(m1), 31);
                    else xb |= (m1 < 0.0) ? (1u << q) : 0u;
                };
                if constexpr (VRX > 0) {  // wide rounds: DVX gathers per variable (missing edges read the zero double)
                    double cw[2][DVX];
#pragma unroll
                    for (int j = 0; j < DVX; ++j) cw[0][j] = gat(half_of<VNK>(vn_idx, j));
                    static_for<0, VRX>([&](auto Q_) {
                        constexpr int q = decltype(Q_)::value;
                        if constexpr (q + 1 < VRX) {
#pragma unroll
                            for (int j = 0; j < DVX; ++j) cw[(q + 1) & 1][j] = gat(half_of<VNK>(vn_idx, (q + 1) * DVX + j));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        double sw = 0.0 + cw[q & 1][0];
#pragma unroll
                        for (int j = 1; j < DVX; ++j) sw += cw[q & 1][j];
                        finish_var(Q_, sw);
                    });
                }
                // narrow rounds, VG of them per pipeline stage: the gathers of stage g+1 are in flight while stage g is summed.  Measured on
                // the headline workload (min-sum, 65 536 frames x 49.4 sweeps): VG = 1 / 2 / 3 -> 6.37 / 6.28 / 6.21 ms; sum-product has no
                // registers to spare (VG = 3 adds spills there)
                if constexpr (!EARLY_GATHERS) {
#pragma unroll
                    for (int u = 0; u < VG; ++u)
#pragma unroll
                        for (int j = 0; j < DV; ++j)
                            if (u < VRN && j < nar_w(u)) cv[0][u][j] = gat(half_of<VNK>(vn_idx, nar_0(u) + j));
                }
                static_for<0, NVG>([&](auto G_) {
                    constexpr int g = decltype(G_)::value;
                    if constexpr (g + 1 < NVG) {
#pragma unroll
                        for (int u = 0; u < VG; ++u)
#pragma unroll
                            for (int j = 0; j < DV; ++j)
                                if ((g + 1) * VG + u < VRN && j < nar_w((g + 1) * VG + u)) cv[(g + 1) & 1][u][j] = gat(half_of<VNK>(vn_idx, nar_0((g + 1) * VG + u) + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, VG>([&](auto U_) {
                        constexpr int u = decltype(U_)::value;
                        if constexpr (g * VG + u < VRN) {
                            double sn = 0.0 + cv[g & 1][u][0];
#pragma unroll
                            for (int j = 1; j < DV; ++j)
                                if (j < nar_w(g * VG + u)) sn += cv[g & 1][u][j];
                            finish_var(std::integral_constant<int, VRX + g * VG + u>{}, sn);
                        }
                    });
                });
                if constexpr (ALG == ALG_MSA) xb = __brev(xr) >> (32 - VRW);
                phase_barrier();
                ++it;
            }
        }
        if (counting) {
            const unsigned wrong = (A.codeword ? ~xb : xb) & valid;
            int err = 0;
#pragma unroll
            for (int q = 0; q < VRW; ++q) err += __popcll(__ballot((wrong >> q) & 1u));
            if constexpr (NW > 1) {  // channel B of the system row (the sweep loop's last exchange used channel A)
                if (lane == 0) *sysw(SYS_ERR + w) = (uint32_t)err;
                __syncthreads();
                int sum = lane < NW ? (int)*sysw(SYS_ERR + (lane & (NW - 1))) : 0;
#pragma unroll
                for (int o = NW / 2; o; o >>= 1) sum += __shfl_xor(sum, o);
                err = sum;
            }
            err = __builtin_amdgcn_readfirstlane(err);
            sim_count(accv, lane, err, it, A.hist_bins);
            if (++acc_frames >= A.flush_every) {
                if (w == 0) sim_flush(accv, lane, A.hist_bins, A.counters);
                accv = 0;
                acc_frames = 0;
            }
        } else {
            if (w == 0 && lane == 0) A.iters[fr] = it;
            uint8_t* xf = A.xhat + fr * n;
#pragma unroll
            for (int q = 0; q < VRW; ++q)
                if (vmap_at(q) >= 0) xf[vmap_at(q)] = (uint8_t)((xb >> q) & 1u);
            if (A.soft != nullptr) {  // marginals of the last executed sweep: still in this wave's LDS rows
                double* sf = reinterpret_cast<double*>(A.soft) + fr * n;
#pragma unroll
                for (int q = 0; q < VRW; ++q)
                    if (vmap_at(q) >= 0) sf[vmap_at(q)] = it > 0 ? my_marg[q * 64] : 0.0;
            }
        }
    }
    if (counting && w == 0) sim_flush(accv, lane, A.hist_bins, A.counters);
}


template <int ALG, int DC, int DV, int CRW, int VRW, int NW, int VRX = 0, int DVX = DV>
constexpr ShapeEntry shape_entry64() {
    static_assert(CRW * DC <= 64, "pad-position mask is one 64-bit word");
    return ShapeEntry{ALG, DC, DV, CRW, VRW, NW, VRX, DVX, (const void*)k_fused_f64<ALG, DC, DV, CRW, VRW, NW, false, VRX, DVX>,
                      (const void*)k_fused_f64<ALG, DC, DV, CRW, VRW, NW, true, VRX, DVX>, 8};
}

template <int ALG, int DC, int DV, int CRW, int VRW, int NW, int VRX = 0, int DVX = DV>
constexpr ShapeEntry shape_entry() {
    return ShapeEntry{ALG, DC, DV, CRW, VRW, NW, VRX, DVX, (const void*)k_fused_bp<ALG, DC, DV, CRW, VRW, NW, false, VRX, DVX>,
                      (const void*)k_fused_bp<ALG, DC, DV, CRW, VRW, NW, true, VRX, DVX>};
}
// the same shape with the exact-in-fp32 variants (min-sum only)
template <int DC, int DV, int CRW, int VRW, int NW, int VRX = 0, int DVX = DV>
constexpr ShapeEntry shape_entry_grid() {
    ShapeEntry e = shape_entry<ALG_MSA, DC, DV, CRW, VRW, NW, VRX, DVX>();
    e.kernel_grid = (const void*)k_fused_bp_grid<DC, DV, CRW, VRW, NW, false, VRX, DVX>;
    e.kernel_sim_grid = (const void*)k_fused_bp_grid<DC, DV, CRW, VRW, NW, true, VRX, DVX>;
    return e;
}

}  // namespace
}  // namespace ldpc
