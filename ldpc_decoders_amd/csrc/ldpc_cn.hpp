// Check-node and variable-node arithmetic shared by the streaming and fused kernels.
//
// Every rule is stated per lane (lane == frame) on a register array holding the row's
// variable->check messages in edge order, and overwrites it with the check->variable messages.
//
//   min-sum      reference src/bpa.py:86-102 (+ src/math_utils.py:10,38-43,78-94)
//   sum-product  reference src/bpa.py:71-75  (+ src/math_utils.py:47-60)
//   (erasure decoder, src/bec.py:100-112: bit-sliced, ldpc_bec_planes.hpp / ldpc_bec_kernels.hpp / ldpc_bec_stream.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ldpc {

template <typename T>
struct real_traits;
template <>
struct real_traits<float> {
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
};
template <>
struct real_traits<double> {
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
};

// ---- min-sum ------------------------------------------------------------------------------------
// |extrinsic| = second minimum at the FIRST arg-min edge, first minimum elsewhere (== leave-one-out min);
// extrinsic sign = (-1)^(#(v<0) in the row) / sgn(v_own) with sgn(0) = +1.  Only compares and negations:
// bit-exact against the fp64 reference in any precision that represents the inputs.
template <typename T, int DCMAX>
__device__ __forceinline__ void cn_msa(T (&v)[DCMAX], int deg) {
    T min1 = real_traits<T>::inf(), min2 = real_traits<T>::inf();
    int arg1 = 0;
    bool parity = false;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            const T a = real_traits<T>::abs(v[j]);
            parity ^= (v[j] < T(0));
            if (a < min1) {
                min2 = min1;
                min1 = a;
                arg1 = j;
            } else if (a < min2) {
                min2 = a;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            const T mag = (j == arg1) ? min2 : min1;
            const bool own_neg = !(v[j] >= T(0));
            v[j] = (parity != own_neg) ? -mag : mag;
        }
    }
}

// ---- sum-product, fp64: the reference formula verbatim ---------------------------------------------
// t = tanh(v/2); row product = sign * exp(sum(log|t|)); extrinsic = product / t_own; 2*atanh with +-1 -> +-inf.
//
// The CHAIN is the reference's, value by value (each of tanh, log, exp, the quotient and atanh is rounded to a double before the next
// step -- that is what produces upstream's saturation artefacts: tanh == 1.0 from |v| ~ 38 on, quantised log|t| below that, q == +-1 ->
// +-inf, inf - inf -> NaN).  Three of the FUNCTIONS are this file's: the device library's tanh, log and atanh are double-double evaluations
// (166 / 94 / 160 VALU instructions, < 1 ulp); the ones below are single-double forms (tanh through expm1 and one division, correctly rounded where
// it saturates; fdlibm's log kernel; fdlibm's atanh through its log1p), 67 / 69 / ~95 instructions.  exp stays the library's.  numpy's own functions
// differ from either by an ulp here and there; agreement is held as measured decisions (tests/test_gpu_parity.py), unchanged: every golden
// case at its measured 100 % (a numpy model of these functions keeps all 2 770 golden frames: tests/test_spa64_functions_cpu.py); in a
// Monte-Carlo run one frame of 131 072 changes its word-error status against the library-function build.  The whole rule is BRANCH-FREE
// on purpose (NaN by arithmetic, selects after an empty asm on their operands): exec-mask regions keep a row's six edges from interleaving.
__device__ __forceinline__ double spa64_nan() { return __builtin_nan(""); }
// tanh(x / 2) with em = expm1(|x|), r = 1 / (em + 2):  1 - 2 r for |x| > 1/2 -- ONE rounding of an exactly representable 1 minus a tiny,
// accurately known term: correctly rounded wherever the saturation artefacts live (the row product is compared with +-1 for EQUALITY
// there) -- and em r below (small arguments, relative accuracy).  |x| clamped at 40 (tanh is exactly 1.0 from ~38.2 on); NaN stays NaN
__device__ __forceinline__ double spa64_tanh_half(double x) {
    double a = __builtin_fabs(x);
    a = (a > 40.0) ? 40.0 : a;
    const double em = expm1(a);
    const double r = 1.0 / (em + 2.0);
    const double t = (a > 0.5) ? (1.0 - (r + r)) : (em * r);
    return __builtin_copysign(t, x);
}
// log(t) for finite t >= 0 or NaN (fdlibm e_log.c, one formula for every range): t = 2^k (1 + f), s = f / (2 + f), log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2)))
__device__ __forceinline__ double spa64_log(double t) {
    const bool sub = t < 2.2250738585072014e-308;  // subnormal (or 0): scale by 2^54
    const double ts = sub ? t * 18014398509481984.0 : t;
    int hx = __double2hiint(ts);
    int k = (hx >> 20) - 1023 - (sub ? 54 : 0);
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;  // mantissa above sqrt(2): use t / 2
    const double m = __hiloint2double(hx | (i ^ 0x3ff00000), __double2loint(ts));
    k += i >> 20;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                        6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    double r = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
    r = (t == 0.0) ? -__builtin_huge_val() : r;
    return r + (t - t);  // NaN -> NaN by arithmetic (t - t is 0 for every finite t): a select here is compiled into a branch AROUND the whole
                         // evaluation, and thirty such regions per check phase keep the six edges of a row from being interleaved
}
// log1p(x) for x >= 0: fdlibm s_log1p.c (its range branches as selects; its two shortcut branches -- |f| < 2^-20 and |x| < 2^-29, which only
// save work there -- left out): below 0.41422 the argument itself is f (no rounding of 1 + x at all), otherwise u = fl(1 + x) = 2^k (1 + f) with
// the correction c = (rounding error of 1 + x) / u carried beside k ln2_lo (c is a correction of a correction: multiplied by rcp(u), not divided)
__device__ __forceinline__ double spa64_log1p(double x) {
    const bool small = __double2hiint(x) < 0x3FDA827A;
    const bool exact1 = x < 9007199254740992.0;  // 1 + x still has the bits of x
    const double u = exact1 ? 1.0 + x : x;
    int hu = __double2hiint(u);
    int k = (hu >> 20) - 1023;
    double c = ((k > 0) ? 1.0 - (u - x) : x - (u - 1.0)) * __builtin_amdgcn_rcp(u);
    c = exact1 ? c : 0.0;
    hu &= 0x000fffff;
    const bool lowm = hu < 0x6a09e;  // mantissa below sqrt(2)
    const double un = __hiloint2double(hu | (lowm ? 0x3ff00000 : 0x3fe00000), __double2loint(u));
    k = lowm ? k : k + 1;
    const double f = small ? x : un - 1.0;
    k = small ? 0 : k;
    c = small ? 0.0 : c;
    const double dk = (double)k;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, 1.479819860511658591e-01,
                     1.531383769920937332e-01), 1.818357216161805012e-01), 2.222219843214978396e-01), 2.857142874366239149e-01), 3.999999999940941908e-01),
                     6.666666666666735130e-01);  // the polynomial alone in fused multiply-adds (a correction term: its last bit never reaches the result's)
    double r0 = f - (hfsq - s * (hfsq + R));
    double rk = dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + (dk * 1.90821492927058770002e-10 + c))) - f);
    asm volatile("" : "+v"(r0), "+v"(rk));  // both evaluated, then selected: no branch (see spa64_log)
    return (k == 0) ? r0 : rk;
}
// atanh: fdlibm e_atanh.c (0.5 log1p(2a + 2a a / (1 - a)) below 1/2, 0.5 log1p(2a / (1 - a)) above; one division serves both).  NOT a 1-2
// ulp shortcut: over the BSC all priors are +-L and upstream's marginals cancel EXACTLY wherever 2 atanh(tanh(L / 2)) == L -- log1p through
// the plain log above lost that on 6 of the 300 frames of the reference's bsc-4_2_test case, on the GPU and in a numpy model of these
// functions alike; this form passes every golden case in the model and here.  |q| > 1 or NaN -> NaN, as atanh.
__device__ __forceinline__ double spa64_atanh(double q) {
    const double a = __builtin_fabs(q);
    const bool lo = a < 0.5;
    const double t2 = a + a;
    const double quot = (lo ? t2 * a : t2) / (1.0 - a);
    double t = 0.5 * spa64_log1p(lo ? t2 + quot : quot);
    asm volatile("" : "+v"(t));        // evaluated for every lane: no branch around it (see spa64_log)
    t = (a <= 1.0) ? t : spa64_nan();  // |q| > 1 or NaN (|q| == 1 is taken out by the caller)
    return __builtin_copysign(t, q);
}
template <int DCMAX>
__device__ __forceinline__ void cn_spa(double (&v)[DCMAX], int deg) {
    double t[DCMAX];
    double slog = 0.0;
    bool parity = false;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            t[j] = spa64_tanh_half(v[j]);
            parity ^= (t[j] < 0.0);
            slog += spa64_log(fabs(t[j]));
        }
    }
    const double prod = (parity ? -1.0 : 1.0) * exp(slog);
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            const double q = prod / t[j];
            double at = spa64_atanh(q);  // at |q| == 1 its value is discarded (2 / 0 = inf inside: no trap, no NaN test on it)
            asm volatile("" : "+v"(at));
            v[j] = 2.0 * ((fabs(q) == 1.0) ? (__builtin_huge_val() * q) : at);
        }
    }
}

// ---- sum-product, fp32: leave-one-out through the even / odd parts of prod (1 + u_i) ---------------------------------
// With u_i = e^-|v2c_i| the reference's factor is tanh(|v2c_i|/2) = (1 - u_i)/(1 + u_i) (src/bpa.py:71-75).  Write
// prod_{i != j} (1 + u_i x) = E_j + x O_j (E_j: even-degree, O_j: odd-degree elementary symmetric sums of the other edges'
// u_i); then prod (1 - u_i) = E_j - O_j, prod (1 + u_i) = E_j + O_j and
//     |c2v_j| = 2 atanh( (E_j - O_j) / (E_j + O_j) ) = ln(E_j / O_j).
// (E, O) start at (1, 0); an edge with weight u maps (E, O) -> (E + u O, O + u E); a prefix and a suffix combine as
// E = Ep Es + Op Os, O = Ep Os + Op Es.  Every term is positive -- no cancellation anywhere: when all other edges are
// reliable O ~ sum u_i keeps full relative precision (fp32 tanh would have saturated to 1 at |LLR| ~ 17), an unreliable edge
// (u = 1) makes E == O exactly, i.e. |c2v| = 0.  Exactly the reference's quantity (and the phi-domain statement of the oracle,
// bp_oracle.spa_phi_check_update); unlike the reference it has no 0/0 at v2c == 0 (src/bpa.py:74 TODO) and no +-inf / NaN
// artefacts: once O underflows (every other edge beyond |LLR| ~ 87) the message saturates at ~87 instead of becoming +inf,
// which would turn the next v2c = marginal - c2v into inf - inf.  Agreement with the fp64 reference is a TOLERANCE
// (tests/test_gpu_parity.py), not bit-exactness.  3 transcendental-unit operations per edge (exp2, log2, log2).
__device__ __forceinline__ float spa_u_of_llr(float a) {  // a = |v2c| >= 0  ->  e^-a on the exp2 unit (v_exp_f32, ~1 ulp)
    return __builtin_amdgcn_exp2f(a * -1.44269504088896340736f);
}
typedef float spa_f2 __attribute__((ext_vector_type(2)));
// Both recurrences are 2-wide: written on float2 so that they map to the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32:
// one instruction per pair; per component the same IEEE fma / multiply as the scalar form, so results are unchanged).
__device__ __forceinline__ void spa_eo_push(float& e, float& o, float u) {  // append one edge: (E + uO, O + uE)
    const spa_f2 r = __builtin_elementwise_fma(spa_f2{u, u}, spa_f2{o, e}, spa_f2{e, o});
    e = r.x;
    o = r.y;
}
__device__ __forceinline__ float spa_llr_of_eo(float ep, float op, float es, float os) {  // ln(E/O) of prefix x suffix
    const spa_f2 t = spa_f2{ep, ep} * spa_f2{es, os};
    const spa_f2 r = __builtin_elementwise_fma(spa_f2{op, op}, spa_f2{os, es}, t);  // (EpEs + OpOs, EpOs + OpEs)
    const float e = r.x;
    const float o = fmaxf(r.y, 1.17549435e-38f);
    return 0.69314718055994530942f * (__builtin_amdgcn_logf(e) - __builtin_amdgcn_logf(o));  // v_log_f32 is log2
}

// ---- the same rule in the BASE-2 LLR domain (LDS-resident fp32 kernels) ------------------------------------------------------
// Belief propagation is invariant under one positive scale on every LLR (sums are linear, decisions are signs, the check rule maps
// scaled inputs to the equally scaled output once exp / ln are replaced by the matching base): with every prior multiplied by log2(e)
// ONCE per frame, u_i = 2^-|v2c'_i| is the bare v_exp_f32 (negation and |.| are source modifiers: no multiply, no fabs) and the message
// |c2v'_j| = log2(E_j) - log2(O_j) the bare pair of v_log_f32 (no ln 2 multiply).  Soft outputs are scaled back by ln 2 where a kernel
// returns them.
// Degree 6 builds the six leave-one-out pairs from a TREE instead of prefix / suffix chains: three pairs P01, P23, P45 (one packed fma
// each), the three joins of two pairs (packed multiply + fma), and one push per edge -- (E_0, O_0) = (P23 x P45) pushed with u_1, ... --
// 15 packed instructions of depth 4 per row where prefix + suffix + join take 22 of depth 7.  Every term is positive, as before.
__device__ __forceinline__ float spa2_u_of_llr(float v) { return __builtin_amdgcn_exp2f(-__builtin_fabsf(v)); }  // 2^-|v|: one v_exp_f32
__device__ __forceinline__ spa_f2 spa2_push(spa_f2 eo, float u) {  // (E + uO, O + uE)
    return __builtin_elementwise_fma(spa_f2{u, u}, spa_f2{eo.y, eo.x}, eo);
}
__device__ __forceinline__ spa_f2 spa2_join(spa_f2 a, spa_f2 b) {  // (EaEb + OaOb, EaOb + OaEb)
    return __builtin_elementwise_fma(spa_f2{a.y, a.y}, spa_f2{b.y, b.x}, spa_f2{a.x, a.x} * b);
}
__device__ __forceinline__ float spa2_llr_of_eo(spa_f2 eo) {  // log2(E / O), O kept off zero (a saturated message, not +inf)
    return __builtin_amdgcn_logf(eo.x) - __builtin_amdgcn_logf(fmaxf(eo.y, 1.17549435e-38f));
}
// u[0..5] -> the six leave-one-out magnitudes log2(E_j / O_j)
__device__ __forceinline__ void spa2_loo6(const float (&u)[6], float (&mag)[6]) {
    const spa_f2 p01 = spa2_push(spa_f2{1.0f, u[0]}, u[1]), p23 = spa2_push(spa_f2{1.0f, u[2]}, u[3]), p45 = spa2_push(spa_f2{1.0f, u[4]}, u[5]);
    const spa_f2 q = spa2_join(p23, p45), r = spa2_join(p01, p45), s = spa2_join(p01, p23);
    mag[0] = spa2_llr_of_eo(spa2_push(q, u[1]));
    mag[1] = spa2_llr_of_eo(spa2_push(q, u[0]));
    mag[2] = spa2_llr_of_eo(spa2_push(r, u[3]));
    mag[3] = spa2_llr_of_eo(spa2_push(r, u[2]));
    mag[4] = spa2_llr_of_eo(spa2_push(s, u[5]));
    mag[5] = spa2_llr_of_eo(spa2_push(s, u[4]));
}
constexpr float SPA2_LOG2E = 1.44269504088896340736f, SPA2_LN2 = 0.69314718055994530942f;

template <int DCMAX>
__device__ __forceinline__ void cn_spa(float (&v)[DCMAX], int deg) {
    float u[DCMAX], pe[DCMAX], po[DCMAX];
    bool parity = false;
    float re = 1.0f, ro = 0.0f;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            u[j] = spa_u_of_llr(fabsf(v[j]));
            parity ^= (v[j] < 0.0f);
            pe[j] = re;  // (E, O) of the edges before j
            po[j] = ro;
            spa_eo_push(re, ro, u[j]);
        }
    }
    float se = 1.0f, so = 0.0f;  // (E, O) of the edges after j
#pragma unroll
    for (int j = DCMAX - 1; j >= 0; --j) {
        if (j < deg) {
            const float mag = spa_llr_of_eo(pe[j], po[j], se, so);
            spa_eo_push(se, so, u[j]);
            const bool own_neg = !(v[j] >= 0.0f);
            v[j] = (parity != own_neg) ? -mag : mag;
        }
    }
}

template <typename T, int ALG, int DCMAX>
__device__ __forceinline__ void cn_rule(T (&v)[DCMAX], int deg) {
    if constexpr (ALG == 0) {
        cn_msa<T, DCMAX>(v, deg);
    } else {
        static_assert(ALG == 1, "min-sum or sum-product (the erasure decoder is bit-sliced: ldpc_bec_planes.hpp)");
        cn_spa<DCMAX>(v, deg);
    }
}

}  // namespace ldpc
