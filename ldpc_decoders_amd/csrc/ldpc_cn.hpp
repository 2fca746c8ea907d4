// Check-node and variable-node arithmetic shared by the streaming and fused kernels.
//
// Every rule is stated per lane (lane == frame) on a register array holding the row's
// variable->check messages in edge order, and overwrites it with the check->variable messages.
//
//   min-sum      reference src/bpa.py:86-102 (+ src/math_utils.py:10,38-43,78-94)
//   sum-product  reference src/bpa.py:71-75  (+ src/math_utils.py:47-60)
//   erasure      reference src/bec.py:100-112
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
template <int DCMAX>
__device__ __forceinline__ void cn_spa(double (&v)[DCMAX], int deg) {
    double t[DCMAX];
    double slog = 0.0;
    bool parity = false;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            t[j] = tanh(v[j] / 2.0);
            parity ^= (t[j] < 0.0);
            slog += log(fabs(t[j]));
        }
    }
    const double prod = (parity ? -1.0 : 1.0) * exp(slog);
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            const double q = prod / t[j];
            v[j] = 2.0 * ((fabs(q) == 1.0) ? (__builtin_huge_val() * q) : atanh(q));
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

// ---- erasure channel (ternary messages in int8) -------------------------------------------------
template <int DCMAX>
__device__ __forceinline__ void cn_bec(int8_t (&v)[DCMAX], int deg) {
    int erased = 0, ones = 0;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            erased += (v[j] == 0);
            ones += (v[j] > 0);
        }
    }
    const int8_t fill = (int8_t)(2 * (ones & 1) - 1);
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            // 0 erasures: echo ; >1: nothing known ; exactly 1: the erased edge learns the parity of the others
            v[j] = erased == 0 ? v[j] : (erased > 1 ? (int8_t)0 : (v[j] == 0 ? fill : (int8_t)0));
        }
    }
}

template <typename T, int ALG, int DCMAX>
__device__ __forceinline__ void cn_rule(T (&v)[DCMAX], int deg) {
    if constexpr (ALG == 0) {
        cn_msa<T, DCMAX>(v, deg);
    } else if constexpr (ALG == 1) {
        cn_spa<DCMAX>(v, deg);
    } else {
        cn_bec<DCMAX>(v, deg);
    }
}

}  // namespace ldpc
