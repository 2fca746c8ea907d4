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

// ---- sum-product, fp32: leave-one-out in the "distance from certainty" domain -----------------------
// With t_i = tanh(|v2c_i|/2) the reference computes |c2v_j| = 2 atanh(prod_{i != j} t_i) (src/bpa.py:71-75).  fp32 tanh
// saturates to exactly 1 at |LLR| ~ 17, so the product is carried as D = 1 - prod instead, built from
// d_i = 1 - t_i = 2u/(1+u), u = e^-|v2c_i|, with the cancellation-free join  1-(1-x)(1-y) = x + y - xy :
//     |c2v_j| = ln((2 - D_j) / D_j),   D_j = join over i != j of d_i   (prefix/suffix joins, dc-1 each).
// Exactly the same quantity as the reference's tanh product (and as the phi-domain statement of the oracle,
// bp_oracle.spa_phi_check_update); small D (all other edges reliable) keeps full relative precision up to |LLR| ~ 88,
// D -> 1 (an unreliable edge) gives |c2v| -> 0 with absolute error ~1e-7.  Unlike the reference it has no 0/0 at
// v2c == 0 (src/bpa.py:74 TODO) and no +-inf / NaN artefacts: above |LLR| ~ 88 the check messages saturate (finite).
// Agreement with the fp64 reference is a TOLERANCE (tests/test_gpu_parity.py), not bit-exactness.
// 4 transcendental-unit operations per edge (exp2, rcp, rcp, log2).
__device__ __forceinline__ float spa_d_of_llr(float a) {  // a = |v2c| >= 0  ->  1 - tanh(a/2)
    const float u = __builtin_amdgcn_exp2f(a * -1.44269504088896340736f);  // e^-a on the exp2 unit (v_exp_f32, ~1 ulp)
    return (2.0f * u) * __builtin_amdgcn_rcpf(1.0f + u);                   // v_rcp_f32, ~1 ulp
}
__device__ __forceinline__ float spa_join(float x, float y) { return fmaf(-x, y, x) + y; }  // 1 - (1-x)(1-y)
__device__ __forceinline__ float spa_llr_of_d(float D) {  // 2 atanh(1 - D) = ln((2 - D) / D)
    // D underflows to 0 once every other edge of the check is beyond |LLR| ~ 88: saturate there (|c2v| <= 88.03) instead of
    // returning +inf, which would turn the next v2c = marginal - c2v into inf - inf = NaN and poison the frame
    D = fmaxf(D, 1.17549435e-38f);
    return 0.69314718055994530942f * __builtin_amdgcn_logf((2.0f - D) * __builtin_amdgcn_rcpf(D));  // v_log_f32 is log2
}

template <int DCMAX>
__device__ __forceinline__ void cn_spa(float (&v)[DCMAX], int deg) {
    float d[DCMAX];
    float pre[DCMAX];
    bool parity = false;
    float run = 0.0f;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            d[j] = spa_d_of_llr(fabsf(v[j]));
            parity ^= (v[j] < 0.0f);
            pre[j] = run;  // join over the edges before j
            run = spa_join(run, d[j]);
        }
    }
    float suf = 0.0f;  // join over the edges after j
#pragma unroll
    for (int j = DCMAX - 1; j >= 0; --j) {
        if (j < deg) {
            const float mag = spa_llr_of_d(spa_join(pre[j], suf));
            suf = spa_join(suf, d[j]);
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
