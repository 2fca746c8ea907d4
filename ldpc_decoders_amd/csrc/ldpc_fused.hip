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

#include "ldpc_common.hpp"

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
};

namespace {

using u64 = unsigned long long;

template <int K>
__device__ __forceinline__ uint32_t half_of(const uint32_t (&tab)[(K + 1) / 2], int k) {
    const uint32_t w = tab[k >> 1];
    return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

__device__ __forceinline__ float lds_ld(const unsigned char* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(base + byte_off);
}

template <int DC, int DV, int CR, int VR>
__global__ __launch_bounds__(64, 2) void k_fused_msa(const float* __restrict__ priors, const uint8_t* __restrict__ y0,
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
                unsat |= par & cn_active[r];
            }
            left_at_0 = early && unsat == 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (!left_at_0) {
#pragma unroll
            for (int q = 0; q < VR; ++q) lds_marg[q * 64 + lane] = prior[q];
            __builtin_amdgcn_wave_barrier();
            for (;;) {
                if (max_iter > 0 && it >= max_iter) break;
                // ---------------- check phase (+ syndrome of the decisions of the previous sweep)
                u64 unsat = 0;
#pragma unroll
                for (int r = 0; r < CR; ++r) {
                    float v[DC], a[DC];
                    u64 par = 0, negpar = 0;
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        const float mg = lds_ld(smem, half_of<CR * DC>(cn_idx, r * DC + j));
                        par ^= __ballot(mg < 0.0f);
                        v[j] = mg - c2v_old[r][j];
                        a[j] = __builtin_fabsf(v[j]);
                        negpar ^= __ballot(v[j] < 0.0f);
                    }
                    unsat |= par & cn_active[r];
                    // leave-one-out minimum of |v|
                    float pre[DC], suf[DC];
                    pre[0] = __builtin_huge_valf();
#pragma unroll
                    for (int j = 1; j < DC; ++j) pre[j] = fminf(pre[j - 1], a[j - 1]);
                    suf[DC - 1] = __builtin_huge_valf();
#pragma unroll
                    for (int j = DC - 2; j >= 0; --j) suf[j] = fminf(suf[j + 1], a[j + 1]);
                    const bool row_neg = (negpar >> lane) & 1ull;
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        const float mag = fminf(pre[j], suf[j]);
                        const bool own_neg = !(v[j] >= 0.0f);
                        const float c = (row_neg != own_neg) ? -mag : mag;
                        c2v_old[r][j] = c;
                        lds_c2v[(r * DC + j) * 64 + lane] = c;
                    }
                }
                if (early && it > 0 && unsat == 0) break;
                __builtin_amdgcn_wave_barrier();
                // ---------------- variable phase
                xb = 0;
#pragma unroll
                for (int q = 0; q < VR; ++q) {
                    float s = lds_ld(smem, half_of<VR * DV>(vn_idx, q * DV));
#pragma unroll
                    for (int j = 1; j < DV; ++j) s += lds_ld(smem, half_of<VR * DV>(vn_idx, q * DV + j));
                    const float mg = prior[q] + s;
                    lds_marg[q * 64 + lane] = mg;
                    xb |= (mg < 0.0f) ? (1u << q) : 0u;
                }
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
    int DC, DV, CR, VR;
    const void* kernel;
};

template <int DC, int DV, int CR, int VR>
constexpr ShapeEntry shape_entry() {
    return ShapeEntry{DC, DV, CR, VR, (const void*)k_fused_msa<DC, DV, CR, VR>};
}

// instantiated shapes: (3,6)-regular codes up to n = 1216 and up to n = 512
const ShapeEntry kShapes[] = {shape_entry<6, 3, 4, 8>(), shape_entry<6, 3, 10, 19>()};

}  // namespace

bool fused_supported(const Decoder* d) { return d->fused && d->fused->ok; }

int fused_plan_create(Decoder* d) {
    const Code* c = d->code;
    d->fused = new FusedPlan();
    FusedPlan* p = d->fused;
    if (d->alg != ALG_MSA || d->dtype != DT_F32) return LDPC_OK;  // other combinations stay on the streaming backend
    if (c->min_dc != c->max_dc) return LDPC_OK;
    const ShapeEntry* shape = nullptr;
    for (const ShapeEntry& s : kShapes)
        if (c->max_dc == s.DC && c->max_dv <= s.DV && c->m <= s.CR * 64 && c->n <= s.VR * 64) {
            shape = &s;
            break;
        }
    if (!shape) return LDPC_OK;
    const int DC = shape->DC, DV = shape->DV, CR = shape->CR, VR = shape->VR;
    p->DC = DC; p->DV = DV; p->CR = CR; p->VR = VR;
    const int NPAD = VR * 64;

    // ---- layout: check c -> slot (r, lane), variable v -> slot s, edge position inside its check.
    std::vector<int> chk_slot(c->m), var_slot(c->n);
    std::iota(chk_slot.begin(), chk_slot.end(), 0);
    std::iota(var_slot.begin(), var_slot.end(), 0);
    std::vector<int> edge_pos(c->E);
    for (int cc = 0; cc < c->m; ++cc)
        for (int k = c->row_ptr[cc]; k < c->row_ptr[cc + 1]; ++k) edge_pos[k] = k - c->row_ptr[cc];

    std::vector<uint32_t> cn_tab((size_t)((CR * DC + 1) / 2) * 64, 0), vn_tab((size_t)((VR * DV + 1) / 2) * 64, 0);
    std::vector<int32_t> var_of_slot((size_t)NPAD, -1);
    std::vector<u64> cn_active((size_t)CR, 0);
    auto put16 = [](std::vector<uint32_t>& tab, int k, int lane, uint32_t val) {
        uint32_t& w = tab[(size_t)(k >> 1) * 64 + lane];
        w = (k & 1) ? ((w & 0x0000ffffu) | (val << 16)) : ((w & 0xffff0000u) | val);
    };
    for (int v = 0; v < c->n; ++v) var_of_slot[var_slot[v]] = v;
    // padded check slots read marg slot of variable 0 (any valid address) and are masked out of the syndrome
    for (int r = 0; r < CR; ++r)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < DC; ++j) put16(cn_tab, r * DC + j, lane, 0);
    for (int cc = 0; cc < c->m; ++cc) {
        const int r = chk_slot[cc] / 64, lane = chk_slot[cc] % 64;
        cn_active[r] |= 1ull << lane;
        for (int k = c->row_ptr[cc]; k < c->row_ptr[cc + 1]; ++k)
            put16(cn_tab, r * DC + edge_pos[k], lane, (uint32_t)(var_slot[c->edge_var[k]] * 4));
    }
    const uint32_t c2v_base = (uint32_t)NPAD * 4;
    for (int q = 0; q < VR; ++q)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < DV; ++j) put16(vn_tab, q * DV + j, lane, c2v_base + (uint32_t)(CR * DC * 64 + lane) * 4);  // zero row
    for (int v = 0; v < c->n; ++v) {
        const int q = var_slot[v] / 64, lane = var_slot[v] % 64;
        int j = 0;
        for (int pidx = c->col_ptr[v]; pidx < c->col_ptr[v + 1]; ++pidx, ++j) {
            const int k = c->col_edge[pidx];  // ascending edge order == the reference's summation order
            const int cs = chk_slot[c->edge_chk[k]];
            put16(vn_tab, q * DV + j, lane, c2v_base + (uint32_t)(((cs / 64) * DC + edge_pos[k]) * 64 + cs % 64) * 4);
        }
    }
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
    p->waves_per_cu = by_lds < 8 ? by_lds : 8;
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
        if (s.DC == p->DC && s.DV == p->DV && s.CR == p->CR && s.VR == p->VR) shape = &s;
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
