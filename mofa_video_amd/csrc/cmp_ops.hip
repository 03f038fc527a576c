// Elementwise pieces of the CMP sparse-to-dense motion encoder that are not convolutions:
//   pool2d            nn.MaxPool2d / nn.AvgPool2d on token-major fp16 maps (resnet.py:108, shallownet.py:16-21,
//                     decoder.py:115,127,139)
//   resize_bilinear   F.interpolate(mode="bilinear", align_corners=True) on token-major fp16 maps (decoder.py:192-211)
//                     and on fp32 NCHW flow fields (..._norefine.py:58-60)
//   flow_expectation  Fuser.convert_flow: per axis softmax over nbins logits, expectation over the bin centres
//                     (cmp/utils/visualize_utils.py:6-19)
// All HBM-bound; 16-byte accesses along the channel axis.
#include <math.h>

#include "common.h"

// one thread = 8 channels of one output pixel.  mode 0: max (padding ignored, as torch), 1: average (no padding used)
__global__ __launch_bounds__(256) void pool2d_kernel(const f16* __restrict__ x, f16* __restrict__ out, const long long total,
                                                     const int Hin, const int Win, const int Hout, const int Wout,
                                                     const int C8, const int ldx, const int ldo, const int k,
                                                     const int stride, const int pad, const int mode) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    const long long pix = i / C8;
    const int ox = (int)(pix % Wout);
    const long long t = pix / Wout;
    const int oy = (int)(t % Hout);
    const long long img = t / Hout;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = mode == 0 ? -INFINITY : 0.0f;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * stride + ky - pad;
        if (iy < 0 || iy >= Hin) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * stride + kx - pad;
            if (ix < 0 || ix >= Win) continue;
            const f16x8 v = *(const f16x8*)(x + ((img * Hin + iy) * Win + ix) * ldx + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = mode == 0 ? fmaxf(acc[e], (float)v[e]) : acc[e] + (float)v[e];
        }
    }
    f16x8 o;
    const float inv = 1.0f / (float)(k * k);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)(mode == 0 ? acc[e] : acc[e] * inv);
    *(f16x8*)(out + pix * ldo + c8 * 8) = o;
}

extern "C" int mofa_pool2d_f16(const void* x, void* out, int nimg, int Hin, int Win, int C, int ldx, int ldo, int k,
                               int stride, int pad, int mode, mofa_stream_t stream) {
    if (!x || !out || nimg <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || C % 8 || ldx % 8 || ldo % 8 || k <= 0 || stride <= 0 ||
        pad < 0 || mode < 0 || mode > 1 || (mode == 1 && pad != 0))
        return MOFA_EINVAL;
    const int Hout = (Hin + 2 * pad - k) / stride + 1, Wout = (Win + 2 * pad - k) / stride + 1;
    if (Hout <= 0 || Wout <= 0) return MOFA_EINVAL;
    const long long total = (long long)nimg * Hout * Wout * (C / 8);
    hipLaunchKernelGGL(pool2d_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out, total,
                       Hin, Win, Hout, Wout, C / 8, ldx, ldo, k, stride, pad, mode);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// align_corners=True source coordinate: dst * (in - 1) / (out - 1)   (0 when out == 1), as ATen computes it in fp32
__device__ __forceinline__ void ac_coord(int d, int in, int out, int& i0, int& i1, float& f) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    float s = scale * (float)d;
    asm volatile("" : "+v"(s));   // keep s rounded: a fused scale*d - i0 would move f by up to an ulp of s (see frontend.hip)
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    f = s - (float)i0;
}

__global__ __launch_bounds__(256) void resize_bilinear_tok_kernel(const f16* __restrict__ x, f16* __restrict__ out,
                                                                  const long long total, const int Hin, const int Win,
                                                                  const int Hout, const int Wout, const int C8, const int ldx,
                                                                  const int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    const long long pix = i / C8;
    const int ox = (int)(pix % Wout);
    const long long t = pix / Wout;
    const int oy = (int)(t % Hout);
    const long long img = t / Hout;
    int y0, y1, x0, x1;
    float fy, fx;
    ac_coord(oy, Hin, Hout, y0, y1, fy);
    ac_coord(ox, Win, Wout, x0, x1, fx);
    const f16* b = x + img * Hin * Win * (long long)ldx + c8 * 8;
    const f16x8 v00 = *(const f16x8*)(b + ((long long)y0 * Win + x0) * ldx), v01 = *(const f16x8*)(b + ((long long)y0 * Win + x1) * ldx);
    const f16x8 v10 = *(const f16x8*)(b + ((long long)y1 * Win + x0) * ldx), v11 = *(const f16x8*)(b + ((long long)y1 * Win + x1) * ldx);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // ATen: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
        const float top = (1.0f - fx) * (float)v00[e] + fx * (float)v01[e];
        const float bot = (1.0f - fx) * (float)v10[e] + fx * (float)v11[e];
        o[e] = (f16)((1.0f - fy) * top + fy * bot);
    }
    *(f16x8*)(out + pix * ldo + c8 * 8) = o;
}

extern "C" int mofa_resize_bilinear_ac_f16(const void* x, void* out, int nimg, int Hin, int Win, int Hout, int Wout, int C,
                                           int ldx, int ldo, mofa_stream_t stream) {
    if (!x || !out || nimg <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || C % 8 || ldx % 8 || ldo % 8)
        return MOFA_EINVAL;
    const long long total = (long long)nimg * Hout * Wout * (C / 8);
    hipLaunchKernelGGL(resize_bilinear_tok_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                       (f16*)out, total, Hin, Win, Hout, Wout, C / 8, ldx, ldo);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ __launch_bounds__(256) void resize_bilinear_nchw_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                   const long long total, const int Hin, const int Win,
                                                                   const int Hout, const int Wout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % Wout);
    const long long t = i / Wout;
    const int oy = (int)(t % Hout);
    const long long plane = t / Hout;
    int y0, y1, x0, x1;
    float fy, fx;
    ac_coord(oy, Hin, Hout, y0, y1, fy);
    ac_coord(ox, Win, Wout, x0, x1, fx);
    const float* b = x + plane * Hin * Win;
    const float top = (1.0f - fx) * b[y0 * Win + x0] + fx * b[y0 * Win + x1];
    const float bot = (1.0f - fx) * b[y1 * Win + x0] + fx * b[y1 * Win + x1];
    out[i] = (1.0f - fy) * top + fy * bot;
}

extern "C" int mofa_resize_bilinear_ac_f32(const float* x, float* out, int nplanes, int Hin, int Win, int Hout, int Wout,
                                           mofa_stream_t stream) {
    if (!x || !out || nplanes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return MOFA_EINVAL;
    const long long total = (long long)nplanes * Hout * Wout;
    hipLaunchKernelGGL(resize_bilinear_nchw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, total, Hin,
                       Win, Hout, Wout);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// one wave per token: logits fp16 [tokens][ld], bins [0,nbins) = x axis, [nbins, 2 nbins) = y axis (nbins <= 128)
__global__ __launch_bounds__(256) void flow_expectation_kernel(const f16* __restrict__ logits, float* __restrict__ out,
                                                               const long long ntok, const int HW, const int ld,
                                                               const int nbins, const float fmax) {
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= ntok) return;
    const int lane = threadIdx.x & 63;
    const float step = 2.0f * fmax / (float)nbins;
    const f16* row = logits + tok * ld;
#pragma unroll
    for (int axis = 0; axis < 2; ++axis) {
        const int b0 = lane, b1 = lane + 64;
        const float l0 = b0 < nbins ? (float)row[axis * nbins + b0] : -INFINITY;
        const float l1 = b1 < nbins ? (float)row[axis * nbins + b1] : -INFINITY;
        const float mx = wave_max(fmaxf(l0, l1));
        const float e0 = b0 < nbins ? __expf(l0 - mx) : 0.0f, e1 = b1 < nbins ? __expf(l1 - mx) : 0.0f;
        const float c0 = (float)b0 * step - fmax + step * 0.5f, c1 = (float)b1 * step - fmax + step * 0.5f;
        const float den = wave_sum(e0 + e1);
        const float num = wave_sum(e0 * c0 + e1 * c1);
        if (lane == 0) out[((tok / HW) * 2 + axis) * HW + tok % HW] = num / den;
    }
}

extern "C" int mofa_flow_expectation_f16(const void* logits, float* flow_nchw, int nimg, int HW, int ld, int nbins, float fmax,
                                         mofa_stream_t stream) {
    if (!logits || !flow_nchw || nimg <= 0 || HW <= 0 || nbins <= 0 || nbins > 128 || ld < 2 * nbins) return MOFA_EINVAL;
    const long long ntok = (long long)nimg * HW;
    hipLaunchKernelGGL(flow_expectation_kernel, dim3(cdiv(ntok, 4)), dim3(256), 0, (hipStream_t)stream, (const f16*)logits,
                       flow_nchw, ntok, HW, ld, nbins, fmax);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
