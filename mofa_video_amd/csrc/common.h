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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf GELU as diffusers' GEGLU uses it: gelu(x) = 0.5 * (x + |x| * erf(|x| / sqrt2)).  erf(z) on [0, 3] is an odd
// minimax polynomial z * P(z^2) of degree 17 (|error| <= 2.9e-5, fitted in tools/fit_erf.py) and 1 beyond: 14 full-rate
// VALU operations that hipcc packs two elements at a time (v_pk_fma_f32), no transcendental.  The previous
// Abramowitz-Stegun 7.1.26 form needed v_exp_f32 + v_rcp_f32 (quarter rate): the GEGLU epilogue evaluates this
// 4 * C times per token and was VALU-bound on it.  |gelu error| <= 6.1e-5 absolute (fp16 rounding of an O(1) result:
// 2.4e-4).
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float z = fminf(ax * 0.70710678118654752f, 3.0f);
    const float u = z * z;
    float p = 4.074214033e-08f;
    p = fmaf(p, u, -1.944823907e-06f);
    p = fmaf(p, u, 4.106053893e-05f);
    p = fmaf(p, u, -5.110369530e-04f);
    p = fmaf(p, u, 4.235427827e-03f);
    p = fmaf(p, u, -2.510286123e-02f);
    p = fmaf(p, u, 1.110793352e-01f);
    p = fmaf(p, u, -3.753148615e-01f);
    p = fmaf(p, u, 1.128268480e+00f);
    return 0.5f * fmaf(ax, z * p, x);
}

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
