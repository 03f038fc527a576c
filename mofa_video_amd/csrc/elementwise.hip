// HBM-bound element-wise / data-movement kernels (16-byte vector accesses along the channel axis).
#include "common.h"

static inline int ew_blocks(long long n) {
    long long nb = (n + 255) / 256;
    return (int)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb));
}

// y = a*x + b*y
__global__ __launch_bounds__(256) void axpby_kernel(const f16* __restrict__ x, f16* __restrict__ y, long long nvec, int CV,
                                                    int ldx, int ldy, float a, float b) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const f16x8 xv = *(const f16x8*)(x + (size_t)row * ldx + cv * 8);
        f16x8 yv = *(const f16x8*)(y + (size_t)row * ldy + cv * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) yv[e] = (f16)(a * (float)xv[e] + b * (float)yv[e]);
        *(f16x8*)(y + (size_t)row * ldy + cv * 8) = yv;
    }
}
extern "C" int mofa_axpby_f16(const void* x, void* y, int M, int C, int ldx, int ldy, float a, float b,
                              mofa_stream_t stream) {
    if (!x || !y || M <= 0 || C <= 0 || C % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0) return MOFA_EINVAL;
    const long long nvec = (long long)M * (C / 8);
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y,
                       nvec, C / 8, ldx, ldy, a, b);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// out = a*x + b*y (the same arithmetic, written elsewhere: a column slice of a concat buffer)
__global__ __launch_bounds__(256) void axpby_out_kernel(const f16* __restrict__ x, const f16* __restrict__ y, f16* __restrict__ out,
                                                        long long nvec, int CV, int ldx, int ldy, int ldo, float a, float b) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const f16x8 xv = *(const f16x8*)(x + (size_t)row * ldx + cv * 8);
        const f16x8 yv = *(const f16x8*)(y + (size_t)row * ldy + cv * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(a * (float)xv[e] + b * (float)yv[e]);
        *(f16x8*)(out + (size_t)row * ldo + cv * 8) = o;
    }
}
extern "C" int mofa_axpby_out_f16(const void* x, const void* y, void* out, int M, int C, int ldx, int ldy, int ldo, float a, float b,
                                  mofa_stream_t stream) {
    if (!x || !y || !out || M <= 0 || C <= 0 || C % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || ldo % 8 != 0) return MOFA_EINVAL;
    const long long nvec = (long long)M * (C / 8);
    hipLaunchKernelGGL(axpby_out_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (const f16*)y,
                       (f16*)out, nvec, C / 8, ldx, ldy, ldo, a, b);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// out[m][j] = x[m][j] * gelu(x[m][Ch + j])
__global__ __launch_bounds__(256) void geglu_kernel(const f16* __restrict__ x, f16* __restrict__ out, long long nvec, int CV,
                                                    int Ch, int ldx, int ldo) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const f16x8 v = *(const f16x8*)(x + (size_t)row * ldx + cv * 8);
        const f16x8 g = *(const f16x8*)(x + (size_t)row * ldx + Ch + cv * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)v[e] * gelu_erf_f((float)g[e]));
        *(f16x8*)(out + (size_t)row * ldo + cv * 8) = o;
    }
}
extern "C" int mofa_geglu_f16(const void* x, void* out, int M, int Ch, int ldx, int ldo, mofa_stream_t stream) {
    if (!x || !out || M <= 0 || Ch <= 0 || Ch % 8 != 0 || ldx % 8 != 0 || ldo % 8 != 0) return MOFA_EINVAL;
    const long long nvec = (long long)M * (Ch / 8);
    hipLaunchKernelGGL(geglu_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)out,
                       nvec, Ch / 8, Ch, ldx, ldo);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const f16* __restrict__ src, f16* __restrict__ dst, long long nvec,
                                                     int CV, int lds, int ldd) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        *(f16x8*)(dst + (size_t)row * ldd + cv * 8) = *(const f16x8*)(src + (size_t)row * lds + cv * 8);
    }
}
extern "C" int mofa_copy2d_f16(const void* src, void* dst, int M, int C, int lds, int ldd, mofa_stream_t stream) {
    if (!src || !dst || M <= 0 || C <= 0 || C % 8 != 0 || lds % 8 != 0 || ldd % 8 != 0) return MOFA_EINVAL;
    const long long nvec = (long long)M * (C / 8);
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)src,
                       (f16*)dst, nvec, C / 8, lds, ldd);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ void silu_f32_kernel(const float* x, float* y, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = silu_f(x[i]);
}
extern "C" int mofa_silu_f32(const float* x, float* y, int n, mofa_stream_t stream) {
    if (!x || !y || n <= 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(silu_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ void cast_f32_f16_kernel(const float* x, f16* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (f16)x[i];
}
__global__ void cast_f16_f32_kernel(const f16* x, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (float)x[i];
}
extern "C" int mofa_cast_f32_to_f16(const float* x, void* y, int64_t n, mofa_stream_t stream) {
    if (!x || !y || n <= 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, (f16*)y,
                       (long long)n);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
extern "C" int mofa_cast_f16_to_f32(const void* x, float* y, int64_t n, mofa_stream_t stream) {
    if (!x || !y || n <= 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, y,
                       (long long)n);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// NCHW fp32 -> token-major fp16 (through an LDS tile so both sides are coalesced)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, f16* __restrict__ y, int C, int HW,
                                                           int ldo, float scale) {
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, n = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) y[((size_t)n * HW + p) * ldo + c] = (f16)(tile[tx][r] * scale);
    }
}
extern "C" int mofa_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int C, int HW, int ldo, float scale,
                                         mofa_stream_t stream) {
    if (!x || !y || n <= 0 || C <= 0 || HW <= 0 || ldo < C) return MOFA_EINVAL;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), n);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (f16*)y, C, HW, ldo, scale);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* __restrict__ x, float* __restrict__ y, int C, int HW,
                                                           int ldx) {
    __shared__ float tile[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32, n = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && p < HW) ? (float)x[((size_t)n * HW + p) * ldx + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) y[((size_t)n * C + c) * HW + p] = tile[tx][r];
    }
}
extern "C" int mofa_nhwc_f16_to_nchw_f32(const void* x, float* y, int n, int C, int HW, int ldx, mofa_stream_t stream) {
    if (!x || !y || n <= 0 || C <= 0 || HW <= 0 || ldx < C) return MOFA_EINVAL;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), n);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, y, C, HW, ldx);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos(t*f_i), sin(t*f_i)], f_i = exp(-ln(1e4)*i/half)
__global__ void timestep_embedding_kernel(const float* t, float* out, int n, int dim) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[r] * freq;
    out[(size_t)r * dim + k] = cosf(a);
    out[(size_t)r * dim + half + k] = sinf(a);
}
extern "C" int mofa_timestep_embedding(const float* t, float* out, int n, int dim, mofa_stream_t stream) {
    if (!t || !out || n <= 0 || dim <= 0 || dim % 2 != 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv((long long)n * dim / 2, 256)), dim3(256), 0,
                       (hipStream_t)stream, t, out, n, dim);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// F.interpolate(flow, scale_factor=1/s) (nearest: src = floor(dst*s)) then / s
__global__ void flow_downscale_kernel(const float* __restrict__ flow, float* __restrict__ out, long long total, int H, int W,
                                      int s) {
    const int h = H / s, w = W / s;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w);
        const long long r = i / w;
        const int y = (int)(r % h);
        const long long plane = r / h;
        out[i] = flow[(plane * H + (long long)y * s) * W + (long long)x * s] / (float)s;
    }
}
extern "C" int mofa_flow_downscale_f32(const float* flow, float* out, int n, int H, int W, int s, mofa_stream_t stream) {
    if (!flow || !out || n <= 0 || s <= 0 || H % s != 0 || W % s != 0) return MOFA_EINVAL;
    const long long total = (long long)n * 2 * (H / s) * (W / s);
    hipLaunchKernelGGL(flow_downscale_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, flow, out, total,
                       H, W, s);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- scheduler math ---------------------------------------------------------------------------------
// `scal` != nullptr: the step's scalars come from device memory (a row of the per-clip step table, mofa_hip.h) -- the form a
// captured hipGraph needs, where by-value arguments would be frozen at capture time
__global__ void prepare_model_input_kernel(const float* __restrict__ lat, const float* __restrict__ img,
                                           f16* __restrict__ out, int T, int HW, int ldo, float inv_arg,
                                           const float* __restrict__ scal) {
    const float inv = scal ? scal[4] : inv_arg;
    const long long total = 2LL * T * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const long long r = i / HW;
        const int t = (int)(r % T);
        const int half = (int)(r / T);
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[c] = (f16)(lat[((size_t)t * 4 + c) * HW + p] * inv);
            o[4 + c] = (f16)img[((size_t)half * 4 + c) * HW + p];
        }
        *(f16x8*)(out + (size_t)i * ldo) = o;
    }
}
extern "C" int mofa_prepare_model_input(const float* latents, const float* image_latents, void* out, int T, int HW,
                                        int ldo, float sigma, mofa_stream_t stream) {
    if (!latents || !image_latents || !out || T <= 0 || HW <= 0 || ldo % 8 != 0 || ldo < 8) return MOFA_EINVAL;
    const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
    hipLaunchKernelGGL(prepare_model_input_kernel, dim3(ew_blocks(2LL * T * HW)), dim3(256), 0, (hipStream_t)stream,
                       latents, image_latents, (f16*)out, T, HW, ldo, inv, (const float*)nullptr);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
extern "C" int mofa_prepare_model_input_dev(const float* latents, const float* image_latents, void* out, int T, int HW,
                                            int ldo, const float* step_scalars, mofa_stream_t stream) {
    if (!latents || !image_latents || !out || !step_scalars || T <= 0 || HW <= 0 || ldo % 8 != 0 || ldo < 8) return MOFA_EINVAL;
    hipLaunchKernelGGL(prepare_model_input_kernel, dim3(ew_blocks(2LL * T * HW)), dim3(256), 0, (hipStream_t)stream,
                       latents, image_latents, (f16*)out, T, HW, ldo, 0.0f, step_scalars);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// cur[0 .. MOFA_STEP_SCALARS) = table[*counter][...]; ++*counter: the first node of a captured denoise step
__global__ void step_select_kernel(const float* __restrict__ table, int* __restrict__ counter, float* __restrict__ cur, int nsteps) {
    const int i = *counter;
    const int row = i < nsteps ? i : nsteps - 1;
    if (threadIdx.x < MOFA_STEP_SCALARS) cur[threadIdx.x] = table[(size_t)row * MOFA_STEP_SCALARS + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) *counter = i + 1;
}
extern "C" int mofa_step_select(const float* table, int* counter, float* cur, int nsteps, mofa_stream_t stream) {
    if (!table || !counter || !cur || nsteps <= 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(step_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, table, counter, cur, nsteps);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ void cfg_euler_kernel(float* __restrict__ lat, const f16* __restrict__ np, int T, int HW, int ldn, float sigma_arg,
                                 float sigma_next_arg, float gmin, float gmax, const float* __restrict__ scal) {
    const float sigma = scal ? scal[0] : sigma_arg, sigma_next = scal ? scal[1] : sigma_next_arg;
    const long long total = (long long)T * HW;
    const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
    const float c_skip = 1.0f / (sigma * sigma + 1.0f);
    const float dt = sigma_next - sigma;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int t = (int)(i / HW);
        const float g = (T > 1) ? gmin + (gmax - gmin) * (float)t / (float)(T - 1) : gmin;
        const f16x4 u = *(const f16x4*)(np + (size_t)i * ldn);
        const f16x4 cnd = *(const f16x4*)(np + ((size_t)total + i) * ldn);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float vu = (float)u[c], vc = (float)cnd[c];
            const float v = vu + g * (vc - vu);
            float* xp = lat + ((size_t)t * 4 + c) * HW + p;
            const float x = *xp;
            const float x0 = v * c_out + x * c_skip;
            const float d = (x - x0) / sigma;
            *xp = x + d * dt;
        }
    }
}
extern "C" int mofa_cfg_euler_step(float* latents, const void* noise_pred, int T, int HW, int ldn, float sigma,
                                   float sigma_next, float gmin, float gmax, mofa_stream_t stream) {
    if (!latents || !noise_pred || T <= 0 || HW <= 0 || ldn % 4 != 0 || sigma <= 0.f) return MOFA_EINVAL;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(ew_blocks((long long)T * HW)), dim3(256), 0, (hipStream_t)stream, latents,
                       (const f16*)noise_pred, T, HW, ldn, sigma, sigma_next, gmin, gmax, (const float*)nullptr);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
extern "C" int mofa_cfg_euler_step_dev(float* latents, const void* noise_pred, int T, int HW, int ldn, const float* step_scalars,
                                       float gmin, float gmax, mofa_stream_t stream) {
    if (!latents || !noise_pred || !step_scalars || T <= 0 || HW <= 0 || ldn % 4 != 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(ew_blocks((long long)T * HW)), dim3(256), 0, (hipStream_t)stream, latents,
                       (const f16*)noise_pred, T, HW, ldn, 1.0f, 0.0f, gmin, gmax, step_scalars);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- fp32 vector axpby (Keypoint window accumulation) -------------------------------------------------------------
__global__ void axpby_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float a, float b) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = (b == 0.0f) ? a * x[i] : a * x[i] + b * y[i];
}
extern "C" int mofa_axpby_f32(const float* x, float* y, int64_t n, float a, float b, mofa_stream_t stream) {
    if (!x || !y || n <= 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(axpby_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n, a, b);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int H, int W,
                                      int h, int w) {
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % w);
        const long long r = i / w;
        const int oy = (int)(r % h);
        const long long plane = r / h;
        int iy = (int)floorf((float)oy * sy), ix = (int)floorf((float)ox * sx);
        iy = iy < H - 1 ? iy : H - 1;
        ix = ix < W - 1 ? ix : W - 1;
        y[i] = x[(plane * H + iy) * W + ix];
    }
}
extern "C" int mofa_resize_nearest_f32(const float* x, float* y, int n, int H, int W, int h, int w,
                                       mofa_stream_t stream) {
    if (!x || !y || n <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return MOFA_EINVAL;
    const long long total = (long long)n * h * w;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, H,
                       W, h, w);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- Hybrid residual blend: out = a*w + b*(1-w), w per pixel -----------------------------------------------------------
__global__ __launch_bounds__(256) void mask_blend_kernel(const f16* __restrict__ a, const f16* __restrict__ b,
                                                         const float* __restrict__ w, f16* __restrict__ out,
                                                         long long nvec, int CV, int HW, int lda, int ldb, int ldo) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const float m = w[row % HW];
        const f16x8 av = *(const f16x8*)(a + (size_t)row * lda + cv * 8);
        const f16x8 bv = *(const f16x8*)(b + (size_t)row * ldb + cv * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)av[e] * m + (float)bv[e] * (1.0f - m));
        *(f16x8*)(out + (size_t)row * ldo + cv * 8) = o;
    }
}
extern "C" int mofa_mask_blend_f16(const void* a, const void* b, const float* w, void* out, int M, int C, int HW, int lda,
                                   int ldb, int ldo, mofa_stream_t stream) {
    if (!a || !b || !w || !out || M <= 0 || C <= 0 || HW <= 0 || C % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0 ||
        ldo % 8 != 0)
        return MOFA_EINVAL;
    const long long nvec = (long long)M * (C / 8);
    hipLaunchKernelGGL(mask_blend_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                       (const f16*)b, w, (f16*)out, nvec, C / 8, HW, lda, ldb, ldo);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- ForegroundMatting tail: out = warped*sigmoid(l) + matting*(1 - sigmoid(l)) --------------------------------------
__global__ __launch_bounds__(256) void matting_blend_kernel(const f16* __restrict__ wp, const f16* __restrict__ mt,
                                                            const f16* __restrict__ lg, f16* __restrict__ out,
                                                            float* __restrict__ mask_out, long long nvec, int CV,
                                                            int ldw, int ldm, int ldl, int ldo) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const float m = 1.0f / (1.0f + __expf(-(float)lg[(size_t)row * ldl]));
        if (mask_out && cv == 0) mask_out[row] = m;
        const f16x8 av = *(const f16x8*)(wp + (size_t)row * ldw + cv * 8);
        const f16x8 bv = *(const f16x8*)(mt + (size_t)row * ldm + cv * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)av[e] * m + (float)bv[e] * (1.0f - m));
        *(f16x8*)(out + (size_t)row * ldo + cv * 8) = o;
    }
}
extern "C" int mofa_matting_blend_f16(const void* warped, const void* matting, const void* logit, void* out,
                                      float* mask_out, int M, int C, int ldw, int ldm, int ldl, int ldo,
                                      mofa_stream_t stream) {
    if (!warped || !matting || !logit || !out || M <= 0 || C <= 0 || C % 8 != 0 || ldw % 8 != 0 || ldm % 8 != 0 ||
        ldo % 8 != 0 || ldl <= 0)
        return MOFA_EINVAL;
    const long long nvec = (long long)M * (C / 8);
    hipLaunchKernelGGL(matting_blend_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const f16*)warped, (const f16*)matting, (const f16*)logit, (f16*)out, mask_out, nvec, C / 8, ldw,
                       ldm, ldl, ldo);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- F.interpolate(x, scale_factor=1/s) (nearest) on token-major maps: out[(n, y, x)] = in[(n, y*s, x*s)] --------------
__global__ __launch_bounds__(256) void subsample_tokens_kernel(const f16* __restrict__ x, f16* __restrict__ y,
                                                               long long nvec, int CV, int H, int W, int s, int ldx,
                                                               int ldy) {
    const int h = H / s, w = W / s;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const int ox = (int)(row % w);
        const long long r = row / w;
        const int oy = (int)(r % h);
        const long long n = r / h;
        const long long src = (n * H + (long long)oy * s) * W + (long long)ox * s;
        *(f16x8*)(y + (size_t)row * ldy + cv * 8) = *(const f16x8*)(x + (size_t)src * ldx + cv * 8);
    }
}
extern "C" int mofa_subsample_tokens_f16(const void* x, void* y, int n, int H, int W, int s, int C, int ldx, int ldy,
                                         mofa_stream_t stream) {
    if (!x || !y || n <= 0 || s <= 0 || H % s != 0 || W % s != 0 || C % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0)
        return MOFA_EINVAL;
    const long long nvec = (long long)n * (H / s) * (W / s) * (C / 8);
    hipLaunchKernelGGL(subsample_tokens_kernel, dim3(ew_blocks(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                       (f16*)y, nvec, C / 8, H, W, s, ldx, ldy);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

extern "C" int mofa_version(void) { return 104; }
