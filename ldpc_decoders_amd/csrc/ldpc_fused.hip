// Fused on-chip backend: one wavefront owns one frame for ALL of its sweeps; messages never leave the CU.
//
// Regime: (dv,dc)-regular codes whose per-frame state fits the LDS (n = 1200 (3,6): 20 KB -> 8 frames per CU).
// The HBM traffic of a frame shrinks from sizeof(T)(4E+n) PER SWEEP (streaming backend) to its priors in and its
// decisions out, once; the kernel is bound by LDS gathers and VALU instead.
//
// Mapping (all tables are built once on the host, FusedPlan):
//   * lane L of the wave owns check slots (r, L), r = 0..CR-1, and variable slots (q, L), q = 0..VR-1.
//   * registers: the check->variable messages of the owned checks (c2v_old[CR][DC]), the priors of the owned
//     variables, and every gather address (packed 16-bit LDS byte offsets) -- loaded once per launch.
//   * LDS (per wave): marg[VR*64]  marginal of variable slot s at dword s
//                     c2v [(CR*DC+1)*64]  message of (check slot (r,L), edge position j) at dword (r*DC+j)*64+L;
//                                         the last row stays 0 (target of padded gathers)
//   * check phase : v2c_j = marg[var] - c2v_old_j  (6 LDS gathers), leave-one-out min + sign parity, write c2v
//                   (lane-contiguous, conflict-free); the syndrome of the previous decisions falls out of the same
//                   gathers (sign of marg) -- that is the reference's early-exit test (src/bpa.py:29).
//   * variable phase: marginal = prior + ((c_a + c_b) + c_c) in ascending edge order (3 LDS gathers), write marg.
//   Single-wave workgroups: LDS is private to the wave and DS operations of one wave execute in order, so the two
//   phases need no s_barrier.
//
// Arithmetic identical to the streaming backend / reference (src/bpa.py:17-63, 86-102): the leave-one-out minimum
// equals "second minimum at the first arg-min, first minimum elsewhere"; min/compare/negate/add/sub only.
#include <algorithm>
#include <numeric>

#include <cstdlib>
#include <type_traits>

#include "ldpc_cn.hpp"
#include "ldpc_common.hpp"
#include "ldpc_layout.hpp"

namespace ldpc {

struct FusedPlan {
    bool ok = false;
    int DC = 0, DV = 0, CR = 0, VR = 0;
    uint32_t* d_cn_tab = nullptr;     // [(CR*DC+1)/2][64] two 16-bit marg byte offsets per word
    uint32_t* d_vn_tab = nullptr;     // [(VR*DV+1)/2][64] two 16-bit c2v byte offsets per word
    int32_t* d_var_of_slot = nullptr;  // [VR*64] variable index of a slot, -1 for padding
    unsigned long long* d_cn_active = nullptr;  // [CR] lanes holding a real check in round r
    unsigned long long* d_next = nullptr;       // frame dispenser
    size_t lds_bytes = 0;
    int waves_per_cu = 0, num_cu = 0;
    double extra_identity = 0, extra_planned = 0, base_cycles = 0;
};

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
__device__ __forceinline__ void lds_set_m0(uint32_t base) { asm volatile("s_mov_b32 m0, %0" ::"s"(base) : "memory"); }

__device__ __forceinline__ float lds_ld(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(base + byte_off);
}

template <int ALG, int DC, int DV, int CR, int VR>
__global__ __launch_bounds__(64, 2) void k_fused_bp(const float* __restrict__ priors, const uint8_t* __restrict__ y0,
                                                     long long B, int n, int max_iter, unsigned flags,
                                                     const uint32_t* __restrict__ cn_tab, const uint32_t* __restrict__ vn_tab,
                                                     const int32_t* __restrict__ var_of_slot,
                                                     const u64* __restrict__ cn_active, uint8_t* __restrict__ xhat,
                                                     int32_t* __restrict__ iters, u64* __restrict__ next_frame) {
    constexpr int NPAD = VR * 64;
    constexpr int CNW = (CR * DC + 1) / 2, VNW = (VR * DV + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    float* lds_marg = reinterpret_cast<float*>(smem);
    float* lds_c2v = reinterpret_cast<float*>(smem) + NPAD;

    // gather addresses, resident in registers for the whole launch
    uint32_t cn_idx[CNW], vn_idx[VNW];
#pragma unroll
    for (int i = 0; i < CNW; ++i) cn_idx[i] = cn_tab[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < VNW; ++i) vn_idx[i] = vn_tab[i * 64 + lane];
    lds_c2v[CR * DC * 64 + lane] = 0.0f;  // the always-zero row

    const bool early = !(flags & FLAG_NO_EARLY_EXIT);
    for (;;) {
        u64 fr = 0;
        if (lane == 0) fr = atomicAdd(next_frame, 1ull);
        fr = ((u64)__builtin_amdgcn_readfirstlane((unsigned)(fr >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)fr);
        if ((long long)fr >= B) break;
        const float* pf = priors + fr * n;

        float prior[VR];
        float c2v_old[CR][DC];
        unsigned xb = 0;  // bit q = hard decision of variable slot (q, lane)
#pragma unroll
        for (int q = 0; q < VR; ++q) {
            const int v = var_of_slot[q * 64 + lane];
            prior[q] = v >= 0 ? pf[v] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < CR; ++r)
#pragma unroll
            for (int j = 0; j < DC; ++j) c2v_old[r][j] = 0.0f;

        int it = 0;
        bool left_at_0 = false;
        if (y0 != nullptr) {
            // iteration-0 test of the received hard word (src/bpa.py:20,29): park it in the marg area as -+1
            const uint8_t* yf = y0 + fr * n;
#pragma unroll
            for (int q = 0; q < VR; ++q) {
                const int v = var_of_slot[q * 64 + lane];
                const bool one = v >= 0 && yf[v] != 0;
                lds_marg[q * 64 + lane] = one ? -1.0f : 1.0f;
                xb |= one ? (1u << q) : 0u;
            }
            __builtin_amdgcn_wave_barrier();
            u64 unsat = 0;
#pragma unroll
            for (int r = 0; r < CR; ++r) {
                u64 par = 0;
#pragma unroll
                for (int j = 0; j < DC; ++j) par ^= __ballot(lds_ld(smem, half_of<CR * DC>(cn_idx, r * DC + j)) < 0.0f);
                if constexpr (DC % 2 == 0) unsat |= par; else unsat |= par & cn_active[r];
            }
            left_at_0 = early && unsat == 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (!left_at_0) {
#pragma unroll
            for (int q = 0; q < VR; ++q) lds_marg[q * 64 + lane] = prior[q];
            __builtin_amdgcn_wave_barrier();
            // The sweep is one software-pipelined stream of LDS traffic: the gathers of check round r+1 (variable
            // group g+1) are issued before round r (group g) is computed, so a wave always has a full round of
            // ds_reads in flight while it does arithmetic.  (The wave is latency-bound otherwise: 2 waves per SIMD.)
            constexpr int VG = ALG == ALG_MSA ? 4 : 2;  // variable rounds per pipeline stage (sum-product needs the registers)
            constexpr int NVG = (VR + VG - 1) / VG;
            const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem);
            for (;;) {
                if (max_iter > 0 && it >= max_iter) break;
                lds_set_m0(lds_base);
                // ---------------- check phase (+ syndrome of the decisions of the previous sweep)
                uint32_t synd = 0;  // bit 31: some owned check is unsatisfied by the previous decisions
                float mg[2][DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) mg[0][j] = lds_ld(smem, half_of<CR * DC>(cn_idx, j));
                static_for<0, CR>([&](auto R_) {
                    constexpr int r = decltype(R_)::value;
                    if constexpr (r + 1 < CR) {
#pragma unroll
                        for (int j = 0; j < DC; ++j) mg[(r + 1) & 1][j] = lds_ld(smem, half_of<CR * DC>(cn_idx, (r + 1) * DC + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this round's arithmetic
                    // Sign handling on the raw IEEE bits (bit 31), all in the vector ALU: row parity = XOR of the sign
                    // bits, extrinsic sign = parity ^ own sign.  Equivalent to the reference's comparisons
                    // ((v < 0) for the parity, (v >= 0) for the own sign, src/math_utils.py:10,38-43) because v is never
                    // -0.0 here: marginals are built as prior + ((0.0 + c_a) + c_b + c_c) (see the variable phase).
                    float v[DC], a[DC];
                    uint32_t vx = 0, mx = 0;
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        const float m0 = mg[r & 1][j];
                        mx ^= __float_as_uint(m0);
                        v[j] = m0 - c2v_old[r][j];
                        a[j] = __builtin_fabsf(v[j]);
                        vx ^= __float_as_uint(v[j]);
                    }
                    if constexpr (DC % 2 == 0) synd |= mx; else synd |= ((cn_active[r] >> lane) & 1ull) ? mx : 0u;
                    // leave-one-out reduction of |v|: minimum (min-sum) or join of 1 - tanh(|v|/2) (sum-product, ldpc_cn.hpp)
                    float pre[DC], suf[DC];
                    if constexpr (ALG == ALG_MSA) {
                        pre[0] = __builtin_huge_valf();
#pragma unroll
                        for (int j = 1; j < DC; ++j) pre[j] = fminf(pre[j - 1], a[j - 1]);
                        suf[DC - 1] = __builtin_huge_valf();
#pragma unroll
                        for (int j = DC - 2; j >= 0; --j) suf[j] = fminf(suf[j + 1], a[j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < DC; ++j) a[j] = spa_d_of_llr(a[j]);
                        pre[0] = 0.0f;
#pragma unroll
                        for (int j = 1; j < DC; ++j) pre[j] = spa_join(pre[j - 1], a[j - 1]);
                        suf[DC - 1] = 0.0f;
#pragma unroll
                        for (int j = DC - 2; j >= 0; --j) suf[j] = spa_join(suf[j + 1], a[j + 1]);
                    }
                    static_for<0, DC>([&](auto J_) {
                        constexpr int j = decltype(J_)::value;
                        const float mag = ALG == ALG_MSA ? fminf(pre[j], suf[j]) : spa_llr_of_d(spa_join(pre[j], suf[j]));
                        const float c = __uint_as_float(__float_as_uint(mag) | ((vx ^ __float_as_uint(v[j])) & 0x80000000u));
                        c2v_old[r][j] = c;
                        lds_st_tid<(NPAD + (r * DC + j) * 64) * 4>(c);
                    });
                });
                const u64 unsat = __ballot((synd & 0x80000000u) != 0u);
                if (early && it > 0 && unsat == 0) break;
                __builtin_amdgcn_wave_barrier();
                // ---------------- variable phase
                xb = 0;
                float cv[2][VG][DV];
#pragma unroll
                for (int u = 0; u < VG; ++u)
#pragma unroll
                    for (int j = 0; j < DV; ++j)
                        if (u < VR) cv[0][u][j] = lds_ld(smem, half_of<VR * DV>(vn_idx, u * DV + j));
                static_for<0, NVG>([&](auto G_) {
                    constexpr int g = decltype(G_)::value;
                    if constexpr (g + 1 < NVG) {
#pragma unroll
                        for (int u = 0; u < VG; ++u)
#pragma unroll
                            for (int j = 0; j < DV; ++j)
                                if ((g + 1) * VG + u < VR)
                                    cv[(g + 1) & 1][u][j] = lds_ld(smem, half_of<VR * DV>(vn_idx, ((g + 1) * VG + u) * DV + j));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, VG>([&](auto U_) {
                        constexpr int u = decltype(U_)::value;
                        constexpr int q = g * VG + u;
                        if constexpr (q < VR) {
                            float s = 0.0f + cv[g & 1][u][0];  // as scipy: accumulate from +0.0 (keeps -0.0 out of the marginals)
#pragma unroll
                            for (int j = 1; j < DV; ++j) s += cv[g & 1][u][j];
                            const float m1 = prior[q] + s;
                            lds_st_tid<q * 256>(m1);
                            xb |= (m1 < 0.0f) ? (1u << q) : 0u;
                        }
                    });
                });
                __builtin_amdgcn_wave_barrier();
                ++it;
            }
        }
        if (lane == 0) iters[fr] = it;
        uint8_t* xf = xhat + fr * n;
#pragma unroll
        for (int q = 0; q < VR; ++q) {
            const int v = var_of_slot[q * 64 + lane];
            if (v >= 0) xf[v] = (uint8_t)((xb >> q) & 1u);
        }
    }
}

template <typename T>
int upload_vec(const std::vector<T>& h, T** d) {
    LDPC_HIP_TRY(hipMalloc((void**)d, (h.size() + 1) * sizeof(T)));
    LDPC_HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return LDPC_OK;
}

struct ShapeEntry {
    int alg, DC, DV, CR, VR;
    const void* kernel;
};

template <int ALG, int DC, int DV, int CR, int VR>
constexpr ShapeEntry shape_entry() {
    return ShapeEntry{ALG, DC, DV, CR, VR, (const void*)k_fused_bp<ALG, DC, DV, CR, VR>};
}

// instantiated shapes: (3,6)-regular codes up to n = 512 and up to n = 1216, min-sum and sum-product (fp32)
const ShapeEntry kShapes[] = {shape_entry<ALG_MSA, 6, 3, 4, 8>(), shape_entry<ALG_MSA, 6, 3, 10, 19>(),
                              shape_entry<ALG_SPA, 6, 3, 4, 8>(), shape_entry<ALG_SPA, 6, 3, 10, 19>()};

}  // namespace

bool fused_supported(const Decoder* d) { return d->fused && d->fused->ok; }

int fused_info(const Decoder* d, double* out8) {
    const FusedPlan* p = d->fused;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!p || !p->ok) return LDPC_OK;
    out8[0] = 1;
    out8[1] = p->base_cycles;     // conflict-free LDS cycles of the gathers per sweep
    out8[2] = p->extra_identity;  // extra bank-conflict cycles per sweep, trivial placement
    out8[3] = p->extra_planned;   // ... with the planned placement
    out8[4] = p->waves_per_cu;
    out8[5] = (double)p->lds_bytes;
    out8[6] = p->CR;
    out8[7] = p->VR;
    return LDPC_OK;
}

int fused_plan_create(Decoder* d) {
    const Code* c = d->code;
    d->fused = new FusedPlan();
    FusedPlan* p = d->fused;
    if ((d->alg != ALG_MSA && d->alg != ALG_SPA) || d->dtype != DT_F32) return LDPC_OK;  // the rest stays on the streaming backend
    if (c->min_dc != c->max_dc) return LDPC_OK;
    const ShapeEntry* shape = nullptr;
    for (const ShapeEntry& s : kShapes)
        if (s.alg == d->alg && c->max_dc == s.DC && c->max_dv <= s.DV && c->m <= s.CR * 64 && c->n <= s.VR * 64) {
            shape = &s;
            break;
        }
    if (!shape) return LDPC_OK;
    const int DC = shape->DC, DV = shape->DV, CR = shape->CR, VR = shape->VR;
    p->DC = DC; p->DV = DV; p->CR = CR; p->VR = VR;
    const int NPAD = VR * 64;

    // ---- layout: check c -> slot (r, lane), variable v -> slot, edge positions (ldpc_layout.hpp)
    FusedLayout L;
    const char* mode = std::getenv("LDPC_FUSED_LAYOUT");
    const char* ms = std::getenv("LDPC_FUSED_PLAN_MS");
    if (mode && std::string(mode) == "identity") {
        identity_layout(*c, DC, DV, &L);
        L.base_cycles = 2.0 * (CR * DC + VR * DV);
        L.extra_cycles_identity = L.extra_cycles_planned = layout_extra_cycles(*c, DC, DV, CR, VR, L);
    } else {
        plan_fused_layout(*c, DC, DV, CR, VR, 0x1200u, ms ? atof(ms) * 1e-3 : 0.6, &L);
    }
    p->extra_identity = L.extra_cycles_identity;
    p->extra_planned = L.extra_cycles_planned;
    p->base_cycles = L.base_cycles;
    const std::vector<int>&chk_slot = L.chk_slot, &var_slot = L.var_slot, &edge_pos = L.edge_pos, &var_pos = L.var_pos;

    std::vector<uint32_t> cn_tab((size_t)((CR * DC + 1) / 2) * 64, 0), vn_tab((size_t)((VR * DV + 1) / 2) * 64, 0);
    std::vector<int32_t> var_of_slot((size_t)NPAD, -1);
    std::vector<u64> cn_active((size_t)CR, 0);
    auto put16 = [](std::vector<uint32_t>& tab, int k, int lane, uint32_t val) {
        uint32_t& w = tab[(size_t)(k >> 1) * 64 + lane];
        w = (k & 1) ? ((w & 0x0000ffffu) | (val << 16)) : ((w & 0xffff0000u) | val);
    };
    const uint32_t c2v_base = (uint32_t)NPAD * 4;
    std::vector<int64_t> cn_addr((size_t)CR * DC * 64, -1), vn_addr((size_t)VR * DV * 64, -1);  // byte offsets, -1 = padded lane
    for (int v = 0; v < c->n; ++v) var_of_slot[var_slot[v]] = v;
    for (int cc = 0; cc < c->m; ++cc) {
        const int r = chk_slot[cc] / 64, lane = chk_slot[cc] % 64;
        cn_active[r] |= 1ull << lane;
        for (int k = c->row_ptr[cc]; k < c->row_ptr[cc + 1]; ++k)
            cn_addr[(size_t)(r * DC + edge_pos[k]) * 64 + lane] = (int64_t)var_slot[c->edge_var[k]] * 4;
    }
    for (int v = 0; v < c->n; ++v) {
        const int q = var_slot[v] / 64, lane = var_slot[v] % 64;
        // a real variable with fewer than DV edges sums the always-zero row for the missing ones
        for (int j = 0; j < DV; ++j) vn_addr[(size_t)(q * DV + j) * 64 + lane] = c2v_base + (int64_t)(CR * DC * 64 + lane) * 4;
        for (int pidx = c->col_ptr[v]; pidx < c->col_ptr[v + 1]; ++pidx) {
            const int k = c->col_edge[pidx], cs = chk_slot[c->edge_chk[k]];
            vn_addr[(size_t)(q * DV + var_pos[k]) * 64 + lane] = c2v_base + (int64_t)(((cs / 64) * DC + edge_pos[k]) * 64 + cs % 64) * 4;
        }
    }
    // padded lanes may read anything: let them repeat an address of their own half-wave (LDS broadcast, no extra cycle)
    auto fill_padding = [](std::vector<int64_t>& addr, int64_t fallback) {
        for (size_t g0 = 0; g0 < addr.size(); g0 += 32) {
            int64_t rep = fallback;
            for (int l = 0; l < 32; ++l)
                if (addr[g0 + l] >= 0) { rep = addr[g0 + l]; break; }
            for (int l = 0; l < 32; ++l)
                if (addr[g0 + l] < 0) addr[g0 + l] = rep;
        }
    };
    if (DC % 2 == 0) {
        // even dc: a padded check lane reads ONE marginal dc times, so its sign parity is even and it never shows up in
        // the syndrome (no lane mask needed).  Pick, per half-wave, the slot that collides least with the real reads.
        for (int r = 0; r < CR; ++r)
            for (int h = 0; h < 2; ++h) {
                bool any_pad = false;
                for (int l = 0; l < 32; ++l) any_pad |= cn_addr[(size_t)(r * DC) * 64 + h * 32 + l] < 0;
                if (!any_pad) continue;
                int best_slot = 0, best_cost = 1 << 30;
                for (int slot = 0; slot < NPAD && best_cost > 0; ++slot) {
                    int cost = 0;
                    for (int j = 0; j < DC; ++j) {
                        bool clash = false, same = false;
                        for (int l = 0; l < 32; ++l) {
                            const int64_t a = cn_addr[(size_t)(r * DC + j) * 64 + h * 32 + l];
                            if (a < 0) continue;
                            if (a == (int64_t)slot * 4) same = true;
                            else if (((a / 4) & 31) == (slot & 31)) clash = true;
                        }
                        cost += (clash && !same) ? 1 : 0;
                    }
                    if (cost < best_cost) { best_cost = cost; best_slot = slot; }
                }
                for (int j = 0; j < DC; ++j)
                    for (int l = 0; l < 32; ++l) {
                        int64_t& a = cn_addr[(size_t)(r * DC + j) * 64 + h * 32 + l];
                        if (a < 0) a = (int64_t)best_slot * 4;
                    }
            }
    }
    fill_padding(cn_addr, 0);
    fill_padding(vn_addr, c2v_base);
    for (int k = 0; k < CR * DC; ++k)
        for (int lane = 0; lane < 64; ++lane) put16(cn_tab, k, lane, (uint32_t)cn_addr[(size_t)k * 64 + lane]);
    for (int k = 0; k < VR * DV; ++k)
        for (int lane = 0; lane < 64; ++lane) put16(vn_tab, k, lane, (uint32_t)vn_addr[(size_t)k * 64 + lane]);
    p->lds_bytes = (size_t)(NPAD + (CR * DC + 1) * 64) * 4;
    if (p->lds_bytes > 65535) return LDPC_OK;  // 16-bit offsets
    LDPC_HIP_TRY(hipSetDevice(c->device));
    LDPC_TRY(upload_vec(cn_tab, &p->d_cn_tab));
    LDPC_TRY(upload_vec(vn_tab, &p->d_vn_tab));
    LDPC_TRY(upload_vec(var_of_slot, &p->d_var_of_slot));
    LDPC_TRY(upload_vec(cn_active, &p->d_cn_active));
    LDPC_HIP_TRY(hipMalloc((void**)&p->d_next, 64));
    hipDeviceProp_t prop;
    LDPC_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    p->num_cu = prop.multiProcessorCount;
    const int by_lds = (int)((size_t)160 * 1024 / p->lds_bytes);
    int cap = 8;
    if (const char* w = std::getenv("LDPC_FUSED_WAVES")) cap = atoi(w) > 0 ? atoi(w) : 8;  // experiment knob: resident waves per CU
    p->waves_per_cu = by_lds < cap ? by_lds : cap;
    LDPC_HIP_TRY(hipFuncSetAttribute(shape->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    p->ok = p->waves_per_cu >= 1;
    return LDPC_OK;
}

void fused_plan_destroy(Decoder* d) {
    FusedPlan* p = d->fused;
    if (!p) return;
    for (void* q : {(void*)p->d_cn_tab, (void*)p->d_vn_tab, (void*)p->d_var_of_slot, (void*)p->d_cn_active, (void*)p->d_next})
        if (q) (void)hipFree(q);
    delete p;
    d->fused = nullptr;
}

int fused_decode(Decoder* d, const void* priors, const uint8_t* y0, int64_t B, int32_t max_iter, uint32_t flags,
                 uint8_t* xhat, int32_t* iters, hipStream_t st) {
    FusedPlan* p = d->fused;
    if (!p || !p->ok) {
        set_error("fused backend not available for this decoder");
        return LDPC_E_UNSUPPORTED;
    }
    if (B <= 0) return LDPC_OK;
    if (!priors) {
        set_error("priors pointer is null");
        return LDPC_E_ARG;
    }
    const ShapeEntry* shape = nullptr;
    for (const ShapeEntry& s : kShapes)
        if (s.alg == d->alg && s.DC == p->DC && s.DV == p->DV && s.CR == p->CR && s.VR == p->VR) shape = &s;
    const Code* c = d->code;
    LDPC_HIP_TRY(hipMemsetAsync(p->d_next, 0, 8, st));
    long long waves = (long long)p->num_cu * p->waves_per_cu;
    if (waves > B) waves = B;
    const float* pr = (const float*)priors;
    long long Bll = B;
    int n = c->n, mi = max_iter > 0 ? max_iter : 100000;
    unsigned fl = flags;
    void* args[] = {&pr, &y0, &Bll, &n, &mi, &fl, &p->d_cn_tab, &p->d_vn_tab, &p->d_var_of_slot, &p->d_cn_active, &xhat, &iters, &p->d_next};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (d->profile) {
        LDPC_TRY(prof_event(d, 0, &e0));
        LDPC_TRY(prof_event(d, 1, &e1));
        LDPC_HIP_TRY(hipEventRecord(e0, st));
    }
    LDPC_HIP_TRY(hipLaunchKernel(shape->kernel, dim3((unsigned)waves), dim3(64), args, p->lds_bytes, st));
    if (d->profile) {
        LDPC_HIP_TRY(hipEventRecord(e1, st));
        LDPC_HIP_TRY(hipStreamSynchronize(st));
        LDPC_TRY(prof_collect(d, {{2, e0, e1}}));
    }
    d->last_sweeps = max_iter;
    d->last_backend = BK_FUSED;
    return LDPC_OK;
}

}  // namespace ldpc
