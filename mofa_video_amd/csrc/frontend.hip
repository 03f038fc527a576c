// Image conditioning front end (the step before the denoising loop), the pieces that are not GEMMs:
//   filter1d_reflect   one pass of the separable Gaussian blur inside _resize_with_antialiasing
//                      (MOFA-Video-Traj/pipeline/pipeline.py:531-560 -> _gaussian_blur2d :632-645 -> _filter2d :587-610:
//                      reflect padding (k-1)/2 front, rest rear; x pass first, then y)
//   resize_bicubic_ac  F.interpolate(mode="bicubic", align_corners=True) (pipeline.py:562), A = -0.75, clamped taps
//   patchify           CLIP patch embedding operand: the stride = kernel = 14 convolution of
//                      transformers CLIPVisionEmbeddings.patch_embedding is a GEMM over [patches][3*14*14] rows
// fp32 planes in / out as in the reference (the image is cast to the encoder dtype only after the resize,
// pipeline.py:123-125).  All HBM-bound, once per clip.
#include <math.h>

#include "common.h"

__device__ __forceinline__ int reflect_idx(int i, const int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// axis 1: along W, axis 0: along H.  One thread per output element; taps accumulate in tap order (fp32 fma).
__global__ __launch_bounds__(256) void filter1d_reflect_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                               const float* __restrict__ taps, const long long total,
                                                               const int H, const int W, const int k, const int axis) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % W);
    const long long t = i / W;
    const int oy = (int)(t % H);
    const float* p = x + (t / H) * (long long)H * W;
    const int front = (k - 1) / 2;
    float acc = 0.0f;
    if (axis == 1) {
        p += (long long)oy * W;
        for (int j = 0; j < k; ++j) acc = fmaf(taps[j], p[reflect_idx(ox + j - front, W)], acc);
    } else {
        p += ox;
        for (int j = 0; j < k; ++j) acc = fmaf(taps[j], p[(long long)reflect_idx(oy + j - front, H) * W], acc);
    }
    out[i] = acc;
}

extern "C" int mofa_filter1d_reflect_f32(const float* x, float* out, const float* taps, int nplanes, int H, int W, int k,
                                         int axis, mofa_stream_t stream) {
    if (!x || !out || !taps || x == out || nplanes <= 0 || H <= 0 || W <= 0 || k <= 0 || axis < 0 || axis > 1)
        return MOFA_EINVAL;
    if (k - 1 - (k - 1) / 2 >= (axis == 1 ? W : H)) return MOFA_EINVAL;   // reflect padding needs pad < size (as F.pad)
    const long long total = (long long)nplanes * H * W;
    hipLaunchKernelGGL(filter1d_reflect_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, taps, total,
                       H, W, k, axis);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// cubic convolution weights (Keys, A = -0.75) for the taps at floor(s) - 1 .. floor(s) + 2, s = o * (in-1)/(out-1) in fp32
__device__ __forceinline__ void cubic_taps(const int o, const int nin, const int nout, int idx[4], float w[4]) {
    const float scale = nout > 1 ? (float)(nin - 1) / (float)(nout - 1) : 0.0f;
    float s = scale * (float)o;
    // s must be rounded before t = s - floor(s) (ATen rounds it): hipcc would otherwise fuse scale*o - floor(s) into one
    // fma, which moves t by up to an ulp of s (1.5e-5 at s = 255) -- seen as 1e-5 output error against the reference
    asm volatile("" : "+v"(s));
    const float f = floorf(s);
    const float t = fminf(fmaxf(s - f, 0.0f), 1.0f);
    const int i0 = (int)f;
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    w[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    w[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    w[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    w[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = min(max(i0 - 1 + j, 0), nin - 1);
}

__global__ __launch_bounds__(256) void resize_bicubic_ac_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                const long long total, const int Hin, const int Win,
                                                                const int Hout, const int Wout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wout);
    const long long t = i / Wout;
    const int oy = (int)(t % Hout);
    const float* p = x + (t / Hout) * (long long)Hin * Win;
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_taps(oy, Hin, Hout, iy, wy);
    cubic_taps(ox, Win, Wout, ix, wx);
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float* r = p + (long long)iy[a] * Win;
        const float row = r[ix[0]] * wx[0] + r[ix[1]] * wx[1] + r[ix[2]] * wx[2] + r[ix[3]] * wx[3];
        acc += row * wy[a];
    }
    out[i] = acc;
}

extern "C" int mofa_resize_bicubic_ac_f32(const float* x, float* out, int nplanes, int Hin, int Win, int Hout, int Wout,
                                          mofa_stream_t stream) {
    if (!x || !out || nplanes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return MOFA_EINVAL;
    const long long total = (long long)nplanes * Hout * Wout;
    hipLaunchKernelGGL(resize_bicubic_ac_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, total, Hin,
                       Win, Hout, Wout);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 [nimg][C][H][W] -> fp16 [nimg * (H/p) * (W/p)][ld]; column c*p*p + py*p + px (the order of
// conv.weight.reshape(N, C*p*p)), columns >= C*p*p zero.  One thread per output element.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, f16* __restrict__ out, const long long total,
                                                       const int C, const int H, const int W, const int p, const int ld) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % ld);
    const long long row = i / ld;
    const int gw = W / p, gh = H / p;
    const int gx = (int)(row % gw);
    const long long t = row / gw;
    const int gy = (int)(t % gh);
    const long long img = t / gh;
    float v = 0.0f;
    if (col < C * p * p) {
        const int c = col / (p * p), r = col - c * p * p;
        const int py = r / p, px = r - py * p;
        v = x[((img * C + c) * H + gy * p + py) * W + gx * p + px];
    }
    out[i] = (f16)v;
}

extern "C" int mofa_patchify_f16(const float* x, void* out, int nimg, int C, int H, int W, int p, int ld, mofa_stream_t stream) {
    if (!x || !out || nimg <= 0 || C <= 0 || H <= 0 || W <= 0 || p <= 0 || H % p || W % p || ld < C * p * p) return MOFA_EINVAL;
    const long long total = (long long)nimg * (H / p) * (W / p) * ld;
    hipLaunchKernelGGL(patchify_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (f16*)out, total, C, H, W,
                       p, ld);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
