// Output stage of the pipeline (the step right after the hot path):
//   * frames_postprocess: what the reference's tensor2vid does through diffusers' VaeImageProcessor.postprocess
//     (Traj/pipeline/pipeline.py:57-69, :518): denormalise (x / 2 + 0.5).clamp(0, 1), then keep NCHW fp32 ("pt"),
//     or NHWC fp32 ("np"), or NHWC uint8 = round-half-even(x * 255) ("pil", numpy's .round()).
//   * flow_to_image: the Middlebury colour coding of a flow field the reference's UIs display
//     (Traj/utils/flow_viz.py:196-277): unknown (> 1e7) entries zeroed, both components divided by (max radius + eps),
//     colour-wheel interpolation on the angle, saturation by the radius, in fp64 as numpy computes it.
// HBM-bound elementwise kernels; the flow maximum is a two-step reduction (per-block maxima, then one block).
#include "common.h"

__global__ __launch_bounds__(256) void frames_postprocess_kernel(const float* __restrict__ x, void* __restrict__ out,
                                                                 const long long npix, const int HW, const int mode) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // pixel index over frames * H * W
    if (i >= npix) return;
    const long long f = i / HW;
    const int p = (int)(i - f * HW);
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = x[(f * 3 + c) * HW + p] * 0.5f + 0.5f;   // x / 2 is exact, so this equals (x / 2 + 0.5) in fp32
        v[c] = fminf(fmaxf(t, 0.0f), 1.0f);
    }
    if (mode == MOFA_FRAMES_PT) {
        float* o = (float*)out;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(f * 3 + c) * HW + p] = v[c];
    } else if (mode == MOFA_FRAMES_NP) {
        float* o = (float*)out + i * 3;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
        unsigned char* o = (unsigned char*)out + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (unsigned char)rintf(__fmul_rn(v[c], 255.0f));
    }
}

extern "C" int mofa_frames_postprocess_f32(const float* frames_nchw, void* out, int nframes, int H, int W, int mode,
                                           mofa_stream_t stream) {
    if (!frames_nchw || !out || nframes <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return MOFA_EINVAL;
    const long long npix = (long long)nframes * H * W;
    hipLaunchKernelGGL(frames_postprocess_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, frames_nchw, out,
                       npix, H * W, mode);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- flow visualisation ---------------------------------------------------------------------------------------------
#define FLOW_UNKNOWN 1e7f
__device__ __forceinline__ void flow_uv(const float* flow, long long i, float& u, float& v) {
    u = flow[2 * i]; v = flow[2 * i + 1];
    if (fabsf(u) > FLOW_UNKNOWN || fabsf(v) > FLOW_UNKNOWN) { u = 0.0f; v = 0.0f; }
}

__global__ __launch_bounds__(256) void flow_maxrad_kernel(const float* __restrict__ flow, float* __restrict__ part,
                                                          const long long n, const int final_pass) {
    __shared__ float red[4];
    float m = -1.0f;                                        // the reference starts from max(-1, ...)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (final_pass) { m = fmaxf(m, flow[i]); continue; }
        float u, v;
        flow_uv(flow, i, u, v);
        m = fmaxf(m, sqrtf(u * u + v * v));                 // torch.sqrt(u**2 + v**2) in fp32
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// Middlebury colour wheel (make_color_wheel, flow_viz.py:146-193): 55 entries, built per thread on the fly
__device__ __forceinline__ double wheel(int k, int c) {    // k in [0, 55)
    const int RY = 15, YG = 6, GC = 4, CB = 11, BM = 13, MR = 6;
    double r = 0, g = 0, b = 0;
    if (k < RY) { r = 255; g = floor(255.0 * k / RY); }
    else if ((k -= RY) < YG) { r = 255 - floor(255.0 * k / YG); g = 255; }
    else if ((k -= YG) < GC) { g = 255; b = floor(255.0 * k / GC); }
    else if ((k -= GC) < CB) { g = 255 - floor(255.0 * k / CB); b = 255; }
    else if ((k -= CB) < BM) { b = 255; r = floor(255.0 * k / BM); }
    else { k -= BM; b = 255 - floor(255.0 * k / MR); r = 255; }
    return c == 0 ? r : (c == 1 ? g : b);
}

__global__ __launch_bounds__(256) void flow_to_image_kernel(const float* __restrict__ flow, const float* __restrict__ maxrad,
                                                            unsigned char* __restrict__ out, const long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float fu = flow[2 * i], fv = flow[2 * i + 1];
    const bool unknown = fabsf(fu) > FLOW_UNKNOWN || fabsf(fv) > FLOW_UNKNOWN;
    // u / (maxrad + eps): torch divides the fp32 tensor by a python float -> fp32 result
    const float den = (float)((double)maxrad[0] + 2.220446049250313e-16);
    const double u = unknown ? 0.0 : (double)(fu / den), v = unknown ? 0.0 : (double)(fv / den);
    const int ncols = 55;
    // numpy keeps these in fp32 (fp32 arrays combined with python scalars): rad, arctan2 / pi, fk; f = fk - k0 and the
    // colour interpolation are fp64 (int64 / fp64 operands)
    const float uf = (float)u, vf = (float)v;
    const float rad = sqrtf(__fadd_rn(__fmul_rn(uf, uf), __fmul_rn(vf, vf)));
    const float a = __fdiv_rn(atan2f(-vf, -uf), 3.14159274101257324f);
    const float fk = __fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(a, 1.0f), 2.0f), (float)(ncols - 1)), 1.0f);
    const int k0 = (int)floorf(fk);
    int k1 = k0 + 1;
    if (k1 == ncols + 1) k1 = 1;
    const double f = (double)fk - (double)k0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double col0 = wheel(k0 - 1, c) / 255.0, col1 = wheel(k1 - 1, c) / 255.0;
        double col = (1.0 - f) * col0 + f * col1;
        if (rad <= 1.0f) col = 1.0 - (double)rad * (1.0 - col);
        else col *= 0.75;
        out[i * 3 + c] = unknown ? 0 : (unsigned char)floor(255.0 * col);
    }
}

extern "C" int64_t mofa_flow_to_image_ws_bytes(int H, int W) { (void)H; (void)W; return 1025 * (int64_t)sizeof(float); }

extern "C" int mofa_flow_to_image_u8(const float* flow_hw2, unsigned char* out_hw3, int H, int W, void* workspace,
                                     mofa_stream_t stream) {
    if (!flow_hw2 || !out_hw3 || !workspace || H <= 0 || W <= 0) return MOFA_EINVAL;
    const long long n = (long long)H * W;
    float* part = (float*)workspace;                        // [1024] block maxima, then [1024] = the maximum
    const int nb = (int)(n < 1024LL * 256 ? cdiv(n, 256) : 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(flow_maxrad_kernel, dim3(nb), dim3(256), 0, st, flow_hw2, part, n, 0);
    hipLaunchKernelGGL(flow_maxrad_kernel, dim3(1), dim3(256), 0, st, (const float*)part, part + 1024, (long long)nb, 1);
    hipLaunchKernelGGL(flow_to_image_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, flow_hw2, (const float*)(part + 1024),
                       out_hw3, n);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
