// Forward splatting ("softsplat", mode 'avg') for the MOFA-Adapter warp.
//
// Reference semantics: MOFA-Video-Traj/models/softsplat.py:232-274 (append ones channel, splat, divide by
// splatted ones + 1e-7) and the CUDA kernel softsplat_out at :284-345 (every source pixel scatters to its
// 4 bilinear neighbours with fp32 atomicAdd; a non-finite target skips the pixel; each corner is bounds
// checked separately).
//
// mofa_softsplat_avg_f16 -- deterministic GATHER form.  The (target, weight) pairs depend only on the flow,
// not on the channel, so per flow frame we build once a CSR "target -> (corner*HW + source, weight)" and then
// every target row gathers its sources across all C channels with 16-byte loads along the channel axis
// (the feature map is token-major: one source = one contiguous C-vector).  The contributions of a target are
// summed in (corner, source-raster) order -- the same order as the CPU oracle -- so the result is
// run-to-run reproducible (the reference's atomicAdd order is not).  The ones-channel and the final division
// are fused: norm = sum of weights.
//
// mofa_softsplat_scatter_f32 -- the literal scatter/atomicAdd form on NCHW fp32 (for parity classing and as
// the baseline the gather form is measured against).
#include "common.h"

struct Corners {
    int t[4];
    float w[4];
};

// softsplat.py:298-334
__device__ __forceinline__ bool splat_corners(const float* __restrict__ flow, int s, int H, int W, Corners& c) {
    const int HW = H * W;
    const int y = s / W, x = s - y * W;
    const float fx = (float)x + flow[s];
    const float fy = (float)y + flow[HW + s];
    if (!isfinite(fx) || !isfinite(fy)) return false;
    const float flx = floorf(fx), fly = floorf(fy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float x1f = (float)(x0 + 1), y1f = (float)(y0 + 1), x0f = (float)x0, y0f = (float)y0;
    c.w[0] = (x1f - fx) * (y1f - fy);  // NW
    c.w[1] = (fx - x0f) * (y1f - fy);  // NE
    c.w[2] = (x1f - fx) * (fy - y0f);  // SW
    c.w[3] = (fx - x0f) * (fy - y0f);  // SE
    const int cx[4] = {x0, x0 + 1, x0, x0 + 1};
    const int cy[4] = {y0, y0, y0 + 1, y0 + 1};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        c.t[k] = (cx[k] >= 0 && cx[k] < W && cy[k] >= 0 && cy[k] < H) ? cy[k] * W + cx[k] : -1;
    return true;
}

// workspace per flow frame: count[HW] | offset[HW+1] | cursor[HW] | keys[4HW] | wts[4HW]
static inline int64_t ws_ints_per_flow(int HW) { return (int64_t)HW * 3 + 1 + (int64_t)HW * 8; }
extern "C" int64_t mofa_softsplat_ws_bytes(int nflows, int H, int W) {
    return ws_ints_per_flow(H * W) * 4 * (int64_t)nflows + 64;
}

__global__ void ss_count_kernel(const float* __restrict__ flow, int* __restrict__ ws, int H, int W, long long per) {
    const int HW = H * W, i = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= HW) return;
    Corners c;
    if (!splat_corners(flow + (size_t)i * 2 * HW, s, H, W, c)) return;
    int* count = ws + per * i;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c.t[k] >= 0) atomicAdd(&count[c.t[k]], 1);
}

// exclusive scan of count[HW] -> offset[HW+1]; one 1024-thread block per flow frame
__global__ __launch_bounds__(1024) void ss_scan_kernel(int* __restrict__ ws, int HW, long long per) {
    __shared__ int sums[1024];
    int* count = ws + per * blockIdx.x;
    int* offset = count + HW;
    const int tid = threadIdx.x;
    const int seg = (HW + 1023) / 1024;
    const int a = tid * seg;
    int b = a + seg;
    b = b < HW ? b : HW;
    int local = 0;
    for (int j = a; j < b; ++j) local += count[j];
    sums[tid] = local;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = (tid >= o) ? sums[tid - o] : 0;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    int run = sums[tid] - local;
    for (int j = a; j < b; ++j) {
        offset[j] = run;
        run += count[j];
    }
    if (tid == 1023) offset[HW] = sums[1023];
}

__global__ void ss_fill_kernel(const float* __restrict__ flow, int* __restrict__ ws, int H, int W, long long per) {
    const int HW = H * W, i = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= HW) return;
    Corners c;
    if (!splat_corners(flow + (size_t)i * 2 * HW, s, H, W, c)) return;
    int* base = ws + per * i;
    const int* offset = base + HW;
    int* cursor = base + 2 * HW + 1;
    int* keys = base + 3 * HW + 1;
    float* wts = (float*)(base + 3 * HW + 1 + 4 * HW);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (c.t[k] >= 0) {
            const int pos = offset[c.t[k]] + atomicAdd(&cursor[c.t[k]], 1);
            keys[pos] = k * HW + s;
            wts[pos] = c.w[k];
        }
}

// order every target's segment by key = corner*HW + source (keys are distinct).  Segments are short for real flows (a
// handful of sources per pixel): insertion sort.  A strongly convergent flow can put up to 4 HW entries on one pixel, where
// the quadratic sort of one thread would run for seconds -- beyond SS_INSERTION_MAX entries the segment is heap-sorted in
// place instead (O(n log n): 36 864 entries ~ 1 M steps).  Same result either way (a total order on distinct keys).
#define SS_INSERTION_MAX 48
__device__ __forceinline__ void ss_sift_down(int* keys, float* wts, int root, int n) {
    const int kr = keys[root];
    const float wr = wts[root];
    int hole = root;
    for (;;) {
        int child = 2 * hole + 1;
        if (child >= n) break;
        if (child + 1 < n && keys[child + 1] > keys[child]) ++child;
        if (keys[child] <= kr) break;
        keys[hole] = keys[child];
        wts[hole] = wts[child];
        hole = child;
    }
    keys[hole] = kr;
    wts[hole] = wr;
}
__global__ void ss_sort_kernel(int* __restrict__ ws, int HW, long long per) {
    const int i = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= HW) return;
    int* base = ws + per * i;
    const int* offset = base + HW;
    int* keys = base + 3 * HW + 1;
    float* wts = (float*)(base + 3 * HW + 1 + 4 * HW);
    const int a = offset[t], b = offset[t + 1];
    if (b - a <= SS_INSERTION_MAX) {
        for (int j = a + 1; j < b; ++j) {
            const int kj = keys[j];
            const float wj = wts[j];
            int m = j - 1;
            while (m >= a && keys[m] > kj) {
                keys[m + 1] = keys[m];
                wts[m + 1] = wts[m];
                --m;
            }
            keys[m + 1] = kj;
            wts[m + 1] = wj;
        }
    } else {
        int* k = keys + a;
        float* w = wts + a;
        const int n = b - a;
        for (int r = n / 2 - 1; r >= 0; --r) ss_sift_down(k, w, r, n);
        for (int end = n - 1; end > 0; --end) {
            const int kt = k[0]; k[0] = k[end]; k[end] = kt;
            const float wt = w[0]; w[0] = w[end]; w[end] = wt;
            ss_sift_down(k, w, 0, end);
        }
    }
}

// one wave per target pixel; lanes span the channel vectors
__global__ __launch_bounds__(256) void ss_gather_kernel(const f16* __restrict__ feat, const int* __restrict__ ws,
                                                        f16* __restrict__ out, int HW, int C, int ldf, int ldo,
                                                        long long per) {
    const int i = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave;
    if (t >= HW) return;
    const int* base = ws + per * i;
    const int* offset = base + HW;
    const int* keys = base + 3 * HW + 1;
    const float* wts = (const float*)(base + 3 * HW + 1 + 4 * HW);
    const int a = offset[t], b = offset[t + 1];
    const int CV = C >> 3;
    float norm = 0.f;
    for (int j = a; j < b; ++j) norm += wts[j];
    const float inv = 1.0f / (norm + 0.0000001f);
    f16* op = out + ((size_t)i * HW + t) * ldo;
    for (int cv = lane; cv < CV; cv += 64) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int j = a; j < b; ++j) {
            const int src = keys[j] % HW;
            const float w = wts[j];
            const f16x8 v = *(const f16x8*)(feat + (size_t)src * ldf + cv * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e] * w;
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(acc[e] * inv);
        *(f16x8*)(op + cv * 8) = o;
    }
}

extern "C" int mofa_softsplat_avg_f16(const void* feat, const float* flow, void* out, void* ws, int nflows, int H, int W,
                                      int C, int ldf, int ldo, mofa_stream_t stream) {
    if (!feat || !flow || !out || !ws || nflows <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || ldf % 8 != 0 ||
        ldo % 8 != 0)
        return MOFA_EINVAL;
    const int HW = H * W;
    const long long per = ws_ints_per_flow(HW);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, (size_t)per * 4 * nflows, st) != hipSuccess) return MOFA_ELAUNCH;
    dim3 gpix(cdiv(HW, 256), nflows);
    hipLaunchKernelGGL(ss_count_kernel, gpix, dim3(256), 0, st, flow, (int*)ws, H, W, per);
    hipLaunchKernelGGL(ss_scan_kernel, dim3(nflows), dim3(1024), 0, st, (int*)ws, HW, per);
    hipLaunchKernelGGL(ss_fill_kernel, gpix, dim3(256), 0, st, flow, (int*)ws, H, W, per);
    hipLaunchKernelGGL(ss_sort_kernel, gpix, dim3(256), 0, st, (int*)ws, HW, per);
    hipLaunchKernelGGL(ss_gather_kernel, dim3(cdiv(HW, 4), nflows), dim3(256), 0, st, (const f16*)feat, (const int*)ws,
                       (f16*)out, HW, C, ldf, ldo, per);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// literal form of the reference kernel (NCHW fp32, atomicAdd): out_sum must be zero-initialised by the caller
__global__ void ss_scatter_kernel(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ out,
                                  long long total, int C, int H, int W) {
    const int HW = H * W;
    for (long long idx = (long long)blockIdx.x * 512 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 512) {
        const int s = (int)(idx % HW);
        const long long r = idx / HW;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        Corners cr;
        if (!splat_corners(flow + (size_t)n * 2 * HW, s, H, W, cr)) continue;
        const float v = in[idx];
        float* o = out + ((size_t)n * C + c) * HW;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cr.t[k] >= 0) atomicAdd(&o[cr.t[k]], v * cr.w[k]);
    }
}
extern "C" int mofa_softsplat_scatter_f32(const float* in, const float* flow, float* out_sum, int N, int C, int H, int W,
                                          mofa_stream_t stream) {
    if (!in || !flow || !out_sum || N <= 0 || C <= 0 || H <= 0 || W <= 0) return MOFA_EINVAL;
    const long long total = (long long)N * C * H * W;
    long long nb = (total + 511) / 512;
    nb = nb > 65535 ? 65535 : nb;
    hipLaunchKernelGGL(ss_scatter_kernel, dim3((int)nb), dim3(512), 0, (hipStream_t)stream, in, flow, out_sum, total, C,
                       H, W);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- the metric-weighted modes of the reference wrapper (Traj/models/softsplat.py:243-270), off the inference path: 'linear' /
//      'soft' splat [in * w | w] with w = metric / exp(metric) and divide by the splatted last channel ------------------------------
__global__ __launch_bounds__(256) void splat_weight_kernel(const float* __restrict__ in, const float* __restrict__ metric,
                                                           float* __restrict__ out, int C, long long HW, long long total, int mode) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW, r = i / HW;               // r = n * (C + 1) + c
        const int c = (int)(r % (C + 1));
        const long long n = r / (C + 1);
        float w = metric[n * HW + p];
        if (mode == 2) w = expf(w);
        out[i] = c < C ? in[(n * C + c) * HW + p] * w : w;
    }
}
extern "C" int mofa_softsplat_weight_f32(const float* in, const float* metric, float* out, int N, int C, int H, int W, int mode,
                                         mofa_stream_t stream) {
    if (!in || !metric || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (mode != 1 && mode != 2)) return MOFA_EINVAL;
    const long long HW = (long long)H * W, total = (long long)N * (C + 1) * HW;
    long long nb = (total + 255) / 256;
    nb = nb > 16384 ? 16384 : nb;
    hipLaunchKernelGGL(splat_weight_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, in, metric, out, C, HW, total, mode);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// out[n][c] = summed[n][c] / norm(summed[n][C]); eps_mode 0: + 1e-7 ('' / 'addeps'), 1: 0 -> 1 ('zeroeps'), 2: max(., 1e-7)
// ('clipeps'), 3: as it is (any other suffix: the reference leaves the channel untouched)
__global__ __launch_bounds__(256) void splat_normalize_kernel(const float* __restrict__ summed, float* __restrict__ out, int C,
                                                              long long HW, long long total, int eps_mode) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i % HW, r = i / HW;               // r = n * C + c
        const int c = (int)(r % C);
        const long long n = r / C;
        float d = summed[(n * (C + 1) + C) * HW + p];
        if (eps_mode == 0) d = d + 0.0000001f;
        else if (eps_mode == 1) d = d == 0.0f ? 1.0f : d;
        else if (eps_mode == 2) d = fmaxf(d, 0.0000001f);
        out[i] = summed[(n * (C + 1) + c) * HW + p] / d;
    }
}
extern "C" int mofa_softsplat_normalize_f32(const float* summed, float* out, int N, int C, int H, int W, int eps_mode,
                                            mofa_stream_t stream) {
    if (!summed || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || eps_mode < 0 || eps_mode > 3) return MOFA_EINVAL;
    const long long HW = (long long)H * W, total = (long long)N * C * HW;
    long long nb = (total + 255) / 256;
    nb = nb > 16384 ? 16384 : nb;
    hipLaunchKernelGGL(splat_normalize_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, summed, out, C, HW, total, eps_mode);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
