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

// ---- sum-product, fp32: phi-domain, leave-one-out -------------------------------------------------
// phi(x) = -log(tanh(x/2)) = log1p(2/expm1(x)) is its own inverse, so
//   |c2v_j| = phi( sum_{i != j} phi(|v2c_i|) ),   sign as in min-sum.
// Same quantity as the reference's tanh product, but it does not saturate at |LLR| ~ 17 the way
// fp32 tanh does and it avoids the reference's divide (0/0 at v2c == 0, src/bpa.py:74 TODO).
// Agreement with the fp64 reference is a TOLERANCE (tests/test_gpu_parity.py), not bit-exactness.
__device__ __forceinline__ float phi_f32(float x) { return log1pf(2.0f / expm1f(x)); }

template <int DCMAX>
__device__ __forceinline__ void cn_spa(float (&v)[DCMAX], int deg) {
    float ph[DCMAX];
    float pre[DCMAX];
    bool parity = false;
    float run = 0.0f;
#pragma unroll
    for (int j = 0; j < DCMAX; ++j) {
        if (j < deg) {
            ph[j] = phi_f32(fabsf(v[j]));
            parity ^= (v[j] < 0.0f);
            pre[j] = run;  // sum of phi over edges before j
            run += ph[j];
        }
    }
    float suf = 0.0f;  // sum of phi over edges after j
#pragma unroll
    for (int j = DCMAX - 1; j >= 0; --j) {
        if (j < deg) {
            const float mag = phi_f32(pre[j] + suf);
            suf += ph[j];
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
