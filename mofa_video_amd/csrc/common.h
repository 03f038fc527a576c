// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mofa_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MOFA_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return MOFA_ELAUNCH;           \
    } while (0)

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the
// GroupNorm + SiLU pass over 320-channel rows was VALU-limited by it (142 vs 115 us without the activation)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// erf GELU as diffusers' GEGLU uses it: gelu(x) = x * Phi(x), Phi(x) = 0.5 + 0.5 erf(x / sqrt2) ~= 0.5 + w Q(w^2) with
// w = x clamped to +-3 sqrt2.  Q is the odd degree-17 minimax polynomial of erf on [0, 3] (tools/fit_erf.py, |erf error|
// <= 2.9e-5) with the 1/sqrt2 argument scale and the factor 0.5 folded into its coefficients, so the sign needs no
// |x| / copysign handling and the whole thing is 1 v_med3 + 11 full-rate multiply-adds that hipcc packs two elements at a
// time (v_pk_fma_f32) -- no transcendental (v_exp / v_rcp are quarter rate, and the GEGLU epilogue of the implicit GEMM is
// bound by VALU issue: 4 * C of these per token).  |gelu error| <= 5.0e-5 absolute (fp16 rounding of an O(1) result: 2.4e-4).
__device__ __forceinline__ float gelu_phi_f(float x) {
    const float w = __builtin_amdgcn_fmed3f(x, -4.2426405f, 4.2426405f);
    const float u = w * w;
    float q = 5.626766414e-11f;
    q = fmaf(q, u, -5.371867839e-09f);
    q = fmaf(q, u, 2.268295702e-07f);
    q = fmaf(q, u, -5.646214049e-06f);
    q = fmaf(q, u, 9.359061369e-05f);
    q = fmaf(q, u, -1.109400182e-03f);
    q = fmaf(q, u, 9.818118997e-03f);
    q = fmaf(q, u, -6.634692103e-02f);
    q = fmaf(q, u, 3.989031613e-01f);
    return fmaf(w, q, 0.5f);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return x * gelu_phi_f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
