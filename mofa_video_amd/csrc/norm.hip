// GroupNorm(32) / LayerNorm for token-major fp16 activations (HBM-bound kernels).
//
// GroupNorm is two launches (r03; three before):
//   1. gn_partial : per (frame, row-chunk, 256-channel block) fp32 sum / sum-of-squares per group
//   2. gn_apply   : every workgroup combines the partials of ITS statistics set itself (per frame, or per clip of T frames
//                   for TemporalResnetBlock whose statistics span T*H*W; fp64, fixed order) into scale = rstd*gamma,
//                   shift = beta - mean*scale in LDS, then y = x*scale + shift, optional SiLU, over its rows of that frame.
//                   The partials are a few KB..100 KB and L2 resident; recomputing them per workgroup costs less than the
//                   18 us launch-to-launch latency of a separate one-workgroup-per-set finalize kernel (4 054 per clip).
//   (gn_finalize / affine_act remain as separate entry points: the frame-sharded path all-reduces the sums in between.)
// Deterministic (no float atomics across workgroups; no cross-workgroup hand-off).
#include "common.h"

#define GN_MAX_CHUNKS 128

// row chunks per frame.  One partial entry (32 groups x {sum, sum of squares}) per (frame, chunk), ALL channels: few enough
// entries that the applying kernel can combine a clip's worth itself (25 frames x 16 chunks = 400 entries = 100 KB), enough
// workgroups (frames x chunks) to stream at HBM rate with 4 row loads in flight per thread.
static inline int gn_nchunks(int HW) {
    const int n = HW >= 9216 ? cdiv(HW, 576) : cdiv(HW, 144);
    const int cap = HW >= 9216 ? GN_MAX_CHUNKS : 16;
    return n > cap ? cap : (n < 1 ? 1 : n);
}
extern "C" int mofa_gn_nparts(int HW, int C) { (void)C; return gn_nchunks(HW); }

// thread layout: cols = min(C / 8, 256) column threads (8 channels = 16 B each) x rl = 256 / cols row lanes; channels beyond
// 2048 are covered by a second pass over the columns.  Per-thread fp32 sums over <= rows_per_chunk / rl rows, LDS tree over
// the row lanes in fixed order, then ONE thread per group adds its channels in channel order: deterministic.
__global__ __launch_bounds__(256) void gn_partial_kernel(const f16* __restrict__ x, float* __restrict__ part, int HW,
                                                         int C, int ldx, int rows_per_chunk, int nparts) {
    __shared__ float sS[2048], sQ[2048];                   // [row lane][column thread][8] of the current pass
    __shared__ float cS[8192 / 4], cQ[8192 / 4];           // per-channel totals of one pass (<= 2048 channels)
    __shared__ float gS[32], gQ[32];
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, frame = blockIdx.y;
    const int CV = C >> 3;
    const int cols = CV < 256 ? CV : 256;
    const int rl = 256 / cols;                             // row lanes (threads beyond cols * rl idle)
    const int cx = tid % cols, ry = tid / cols;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    r1 = r1 < HW ? r1 : HW;
    const int cpg = C / 32;
    if (tid < 32) { gS[tid] = 0.f; gQ[tid] = 0.f; }
    for (int cv0 = 0; cv0 < CV; cv0 += 256) {              // (one pass unless C > 2048)
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        const int cv = cv0 + cx;
        if (ry < rl && cv < CV) {
            const f16* base = x + (size_t)frame * HW * ldx + cv * 8;
            int r = r0 + ry;
            for (; r + 3 * rl < r1; r += 4 * rl) {         // 4 independent 16-byte loads in flight
                f16x8 a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = *(const f16x8*)(base + (size_t)(r + u * rl) * ldx);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (float)a[u][e];
                        s[e] += v;
                        q[e] = fmaf(v, v, q[e]);
                    }
            }
            for (; r < r1; r += rl) {
                const f16x8 a = *(const f16x8*)(base + (size_t)r * ldx);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (float)a[e];
                    s[e] += v;
                    q[e] = fmaf(v, v, q[e]);
                }
            }
        }
        __syncthreads();                                   // (previous pass's readers of sS / cS are done)
        if (ry < rl) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sS[(ry * cols + cx) * 8 + e] = s[e]; sQ[(ry * cols + cx) * 8 + e] = q[e]; }
        }
        __syncthreads();
        const int nch = cols * 8;                          // channels of this pass
        for (int c = tid; c < nch; c += 256) {
            float ts = 0.f, tq = 0.f;
            for (int r = 0; r < rl; ++r) { ts += sS[r * nch + c]; tq += sQ[r * nch + c]; }
            cS[c] = ts;
            cQ[c] = tq;
        }
        __syncthreads();
        if (tid < 32) {                                    // group tid: its channels that fall into this pass, in order
            int ca = tid * cpg - cv0 * 8, cb = ca + cpg;
            ca = ca < 0 ? 0 : ca;
            cb = cb > nch ? nch : cb;
            float a = 0.f, b = 0.f;
            for (int c = ca; c < cb; ++c) { a += cS[c]; b += cQ[c]; }
            gS[tid] += a;
            gQ[tid] += b;
        }
    }
    if (tid < 32) {
        float* p = part + (((size_t)frame * nparts + chunk) * 32 + tid) * 2;
        p[0] = gS[tid];
        p[1] = gQ[tid];
    }
}

extern "C" int mofa_gn_partial_f16(const void* x, float* part, int nframes, int HW, int C, int ldx,
                                   mofa_stream_t stream) {
    if (!x || !part || nframes <= 0 || HW <= 0 || C <= 0 || C % 32 != 0 || C % 8 != 0 || C > 4096 || ldx % 8 != 0) return MOFA_EINVAL;
    const int nch = gn_nchunks(HW);
    const int rpc = cdiv(HW, nch);
    dim3 grid(nch, nframes);
    hipLaunchKernelGGL(gn_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)x, part, HW, C, ldx, rpc, nch);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// `part` entries from the pair sums an implicit-GEMM epilogue emitted (mofa_igemm_args.stats: fp32 [frames * HW / 64][C], element 2 p =
// sum, 2 p + 1 = sum of squares of columns 2 p, 2 p + 1 over a block of 64 rows): grid = (nparts, frames); the HW / 64 row blocks of
// a frame are dealt to its nparts entries in order; thread (group g, sub-lane u) adds pairs u, u + 8, .. of its group over the
// entry's blocks in block order, then one thread per group adds the 8 sub-lanes in order: deterministic.
__global__ __launch_bounds__(256) void gn_from_stats_kernel(const float* __restrict__ st, float* __restrict__ part, int HW, int C,
                                                            int nparts) {
    __shared__ float sS[8][32], sQ[8][32];
    const int tid = threadIdx.x, sub = tid & 7, g = tid >> 3;
    const int e = blockIdx.x, frame = blockIdx.y;
    const int nb = HW >> 6;
    const int b0 = (int)(((long long)e * nb) / nparts), b1 = (int)(((long long)(e + 1) * nb) / nparts);
    const int cpg = C / 32, ppg = cpg >> 1;
    float s = 0.f, q = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float* row = st + ((size_t)frame * nb + b) * C + g * cpg;
        for (int j = sub; j < ppg; j += 8) {
            const float2 v = *(const float2*)(row + 2 * j);
            s += v.x;
            q += v.y;
        }
    }
    sS[sub][g] = s;
    sQ[sub][g] = q;
    __syncthreads();
    if (tid < 32) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a += sS[r][tid]; b += sQ[r][tid]; }
        float* p = part + (((size_t)frame * nparts + e) * 32 + tid) * 2;
        p[0] = a;
        p[1] = b;
    }
}

extern "C" int mofa_gn_partial_from_stats(const float* stats, float* part, int nframes, int HW, int C, mofa_stream_t stream) {
    if (!stats || !part || nframes <= 0 || HW <= 0 || HW % 64 != 0 || C <= 0 || C % 64 != 0 || C > 4096) return MOFA_EINVAL;
    const int nch = gn_nchunks(HW);
    hipLaunchKernelGGL(gn_from_stats_kernel, dim3(nch, nframes), dim3(256), 0, (hipStream_t)stream, stats, part, HW, C, nch);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift, int HW, int C, int fps, int nparts,
                                                          float eps) {
    __shared__ double dS[8][32], dQ[8][32];
    __shared__ float sMean[32], sRstd[32];
    const int tid = threadIdx.x, g = tid & 31, l8 = tid >> 5;
    const int stat = blockIdx.x;  // one statistics set = fps consecutive frames
    const int total = fps * nparts;
    double a = 0.0, b = 0.0;
    const float* p0 = part + (size_t)stat * fps * nparts * 64;
    for (int i = l8; i < total; i += 8) {
        a += (double)p0[(size_t)i * 64 + g * 2];
        b += (double)p0[(size_t)i * 64 + g * 2 + 1];
    }
    dS[l8][g] = a;
    dQ[l8][g] = b;
    __syncthreads();
    if (tid < 32) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < 8; ++r) { s += dS[r][tid]; q += dQ[r][tid]; }
        const double cnt = (double)fps * (double)HW * (double)(C / 32);
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sMean[tid] = (float)mean;
        sRstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / 32;
    for (int idx = tid; idx < fps * C; idx += 256) {
        const int f = idx / C, c = idx - f * C;
        const int gg = c / cpg;
        const float sc = sRstd[gg] * gamma[c];
        const size_t o = ((size_t)stat * fps + f) * C + c;
        scale[o] = sc;
        shift[o] = beta[c] - sMean[gg] * sc;
    }
}

extern "C" int mofa_gn_finalize(const float* part, const float* gamma, const float* beta, float* scale, float* shift,
                                int nframes, int HW, int C, int frames_per_stat, float eps, mofa_stream_t stream) {
    if (!part || !gamma || !beta || !scale || !shift || nframes <= 0 || frames_per_stat <= 0 ||
        nframes % frames_per_stat != 0 || C % 32 != 0)
        return MOFA_EINVAL;
    const int nparts = mofa_gn_nparts(HW, C);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nframes / frames_per_stat), dim3(256), 0, (hipStream_t)stream, part,
                       gamma, beta, scale, shift, HW, C, frames_per_stat, nparts, eps);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- split form for frame-sharded clips: local partials -> fp64 sums per (stat, group); the caller all-reduces the
//      sums over the ranks that hold the clip's other frames, then finalizes with the GLOBAL element count ----------
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ part, double* __restrict__ sums, int fps,
                                                        int nparts) {
    __shared__ double dS[8][32], dQ[8][32];
    const int tid = threadIdx.x, g = tid & 31, l8 = tid >> 5;
    const int stat = blockIdx.x;
    const int total = fps * nparts;
    double a = 0.0, b = 0.0;
    const float* p0 = part + (size_t)stat * fps * nparts * 64;
    for (int i = l8; i < total; i += 8) {
        a += (double)p0[(size_t)i * 64 + g * 2];
        b += (double)p0[(size_t)i * 64 + g * 2 + 1];
    }
    dS[l8][g] = a;
    dQ[l8][g] = b;
    __syncthreads();
    if (tid < 32) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < 8; ++r) { s += dS[r][tid]; q += dQ[r][tid]; }
        sums[((size_t)stat * 32 + tid) * 2] = s;
        sums[((size_t)stat * 32 + tid) * 2 + 1] = q;
    }
}
extern "C" int mofa_gn_reduce(const float* part, double* sums, int nframes, int HW, int C, int frames_per_stat,
                              mofa_stream_t stream) {
    if (!part || !sums || nframes <= 0 || frames_per_stat <= 0 || nframes % frames_per_stat != 0 || C % 32 != 0)
        return MOFA_EINVAL;
    hipLaunchKernelGGL(gn_reduce_kernel, dim3(nframes / frames_per_stat), dim3(256), 0, (hipStream_t)stream, part, sums,
                       frames_per_stat, mofa_gn_nparts(HW, C));
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ __launch_bounds__(256) void gn_finalize_sums_kernel(const double* __restrict__ sums,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ scale,
                                                               float* __restrict__ shift, int C, int fps, double cnt,
                                                               float eps) {
    __shared__ float sMean[32], sRstd[32];
    const int tid = threadIdx.x, stat = blockIdx.x;
    if (tid < 32) {
        const double s = sums[((size_t)stat * 32 + tid) * 2], q = sums[((size_t)stat * 32 + tid) * 2 + 1];
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sMean[tid] = (float)mean;
        sRstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / 32;
    for (int idx = tid; idx < fps * C; idx += 256) {
        const int f = idx / C, c = idx - f * C;
        const int gg = c / cpg;
        const float sc = sRstd[gg] * gamma[c];
        const size_t o = ((size_t)stat * fps + f) * C + c;
        scale[o] = sc;
        shift[o] = beta[c] - sMean[gg] * sc;
    }
}
extern "C" int mofa_gn_finalize_sums(const double* sums, const float* gamma, const float* beta, float* scale,
                                     float* shift, int nframes, int C, int frames_per_stat, double count_per_group,
                                     float eps, mofa_stream_t stream) {
    if (!sums || !gamma || !beta || !scale || !shift || nframes <= 0 || frames_per_stat <= 0 ||
        nframes % frames_per_stat != 0 || C % 32 != 0 || count_per_group <= 0)
        return MOFA_EINVAL;
    hipLaunchKernelGGL(gn_finalize_sums_kernel, dim3(nframes / frames_per_stat), dim3(256), 0, (hipStream_t)stream, sums,
                       gamma, beta, scale, shift, C, frames_per_stat, count_per_group, eps);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

__global__ __launch_bounds__(256) void affine_act_kernel(const f16* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, f16* __restrict__ y,
                                                         long long nvec, int HW, int C, int ldx, int ldy, int silu) {
    const int CV = C >> 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const long long row = i / CV;
        const int cv = (int)(i - row * CV);
        const int frame = (int)(row / HW);
        const f16x8 a = *(const f16x8*)(x + (size_t)row * ldx + cv * 8);
        const float* sp = scale + (size_t)frame * C + cv * 8;
        const float* hp = shift + (size_t)frame * C + cv * 8;
        const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
        const f32x4 h0 = *(const f32x4*)hp, h1 = *(const f32x4*)(hp + 4);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = fmaf((float)a[e], e < 4 ? s0[e] : s1[e - 4], e < 4 ? h0[e] : h1[e - 4]);
            if (silu) v = silu_f(v);
            o[e] = (f16)v;
        }
        *(f16x8*)(y + (size_t)row * ldy + cv * 8) = o;
    }
}

extern "C" int mofa_affine_act_f16(const void* x, const float* scale, const float* shift, void* y, int nframes, int HW,
                                   int C, int ldx, int ldy, int silu, mofa_stream_t stream) {
    if (!x || !scale || !shift || !y || nframes <= 0 || HW <= 0 || C % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0)
        return MOFA_EINVAL;
    const long long nvec = (long long)nframes * HW * (C / 8);
    long long nb = (nvec + 255) / 256;
    nb = nb > 16384 ? 16384 : nb;
    hipLaunchKernelGGL(affine_act_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, (const f16*)x, scale, shift,
                       (f16*)y, nvec, HW, C, ldx, ldy, silu);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---- fused finalize + apply (single-rank path) ---------------------------------------------------------------------------
// grid = (row chunks of a frame, frames).  Prologue: thread t sums partial value (group t >> 1, s / q = t & 1) ... 64 values
// per partial entry, entries strided over the 4 quarter-workgroups, fp64, then a fixed-order combine of the 4 quarters.
// The summation tree depends only on (fps, nparts): bit-identical run to run and across workgroups of one set.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       f16* __restrict__ y, int HW, int C, int ldx, int ldy, int fps, int total,
                                                       double cnt, int rows_per_wg, float eps, int silu) {
    // fps frames share one statistics set of `total` partial entries (consecutive in `part`) over `cnt` elements per group
    extern __shared__ __attribute__((aligned(16))) char gn_smem[];
    float* sScale = (float*)gn_smem;                         // [C]
    float* sShift = sScale + C;                              // [C]
    __shared__ double dAcc[16][64];
    __shared__ float sMean[32], sRstd[32];
    const int tid = threadIdx.x, frame = blockIdx.y;
    const int stat = frame / fps;
    {
        // 16 threads per entry (16 B = 2 groups x {s, q} each), 16 entries per sweep, 4 sweeps in flight
        const int slot = tid & 15, el = tid >> 4;
        const f32x4* p0 = (const f32x4*)(part + (size_t)stat * total * 64) + slot;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        int i = el;
        for (; i + 48 < total; i += 64) {
            const f32x4 v0 = p0[(size_t)i * 16], v1 = p0[(size_t)(i + 16) * 16], v2 = p0[(size_t)(i + 32) * 16],
                        v3 = p0[(size_t)(i + 48) * 16];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += ((double)v0[k] + (double)v1[k]) + ((double)v2[k] + (double)v3[k]);
        }
        for (; i < total; i += 16) {
            const f32x4 v = p0[(size_t)i * 16];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += (double)v[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dAcc[el][slot * 4 + k] = a[k];
    }
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += dAcc[r][tid];
        dAcc[0][tid] = t;                                    // (row 0 is read only by its own writer thread above)
    }
    __syncthreads();
    if (tid < 32) {
        const double s = dAcc[0][2 * tid], q = dAcc[0][2 * tid + 1];
        const double mean = s / cnt;
        double var = q / cnt - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        sMean[tid] = (float)mean;
        sRstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cpg = C / 32;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float sc = sRstd[g] * gamma[c];
        sScale[c] = sc;
        sShift[c] = beta[c] - sMean[g] * sc;
    }
    __syncthreads();
    const int CV = C >> 3;
    const int r0 = blockIdx.x * rows_per_wg;
    int r1 = r0 + rows_per_wg;
    r1 = r1 < HW ? r1 : HW;
    const int nrows = r1 - r0;
    const f16* xb = x + ((size_t)frame * HW + r0) * ldx;
    f16* yb = y + ((size_t)frame * HW + r0) * ldy;
    // (row, column vector) of this thread, advanced by 256 vectors per iteration without divisions
    int row = tid / CV, cv = tid - row * CV;
    const int drow = 256 / CV, dcv = 256 - drow * CV;
    while (row < nrows) {
        const f16x8 a = *(const f16x8*)(xb + (size_t)row * ldx + cv * 8);
        const f32x4 s0 = *(const f32x4*)(sScale + cv * 8), s1 = *(const f32x4*)(sScale + cv * 8 + 4);
        const f32x4 h0 = *(const f32x4*)(sShift + cv * 8), h1 = *(const f32x4*)(sShift + cv * 8 + 4);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = fmaf((float)a[e], e < 4 ? s0[e] : s1[e - 4], e < 4 ? h0[e] : h1[e - 4]);
            if (silu) v = silu_f(v);
            o[e] = (f16)v;
        }
        *(f16x8*)(yb + (size_t)row * ldy + cv * 8) = o;
        cv += dcv;
        row += drow;
        if (cv >= CV) { cv -= CV; ++row; }
    }
}

// rows per workgroup of gn_apply: ONE round of resident workgroups -- (workgroups that fit the chip) / frames chunks per frame, at
// least 16 rows each.  (Until r04b: about 64 K elements per workgroup, which at level 0 made 2 250 workgroups for 2 048 slots: a
// second round one tenth full, 4.5 TB/s where the same kernel streams 5.2 TB/s on 4 500 or 1 150 workgroups.)
static int gn_apply_rows_per_wg(int HW, int C, int nframes, int total_entries) {
    // launch geometry of this chip, taken ONCE: a function-local static with an initialiser is initialised thread-safely (C++11), so the
    // virtual-rank threads of the tests / several host threads cannot race on it (round-4 advice); every GPU of a node is the same part
    struct Chip { int cus, slots; };
    static const Chip chip = [] {
        int dev = 0, cus = 0, nb = 0;
        if (!(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
              cus > 0))
            cus = 256;
        // (4 KB of dynamic LDS: the occupancy of this kernel is bound by its 8 waves per SIMD, not by C * 8 bytes of LDS, up to C = 1280)
        if (!(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gn_apply_kernel, 256, 4096) == hipSuccess && nb > 0)) nb = 8;
        return Chip{cus, cus * nb};
    }();
    const int slots = chip.slots, n_cu_ = chip.cus;
    // wide inputs (decoder concat buffers, C = 1920 / 2560) are bound by LDS instead: C * 8 B of scale / shift + 8.5 KB static of 160 KB
    const int by_lds = 163840 / (C * 8 + 8704);
    const int per_cu = slots / n_cu_;
    int chunks = (by_lds < per_cu ? n_cu_ * (by_lds < 1 ? 1 : by_lds) : slots) / nframes;
    chunks = chunks < 1 ? 1 : chunks;
    int rpw = cdiv(HW, chunks);
    // (a floor of total_entries * 256 / C rows, so that a workgroup's re-read of its set's partial entries stays below a quarter of its
    // own bytes, was measured too: level-1 temporal norms 56 -> 70 us with 750 workgroups; the entries come from L2, the rows do not)
    (void)total_entries;
    return rpw < 16 ? 16 : rpw;
}

extern "C" int mofa_gn_apply_f16(const void* x, const float* part, const float* gamma, const float* beta, void* y, int nframes,
                                 int HW, int C, int ldx, int ldy, int frames_per_stat, float eps, int silu,
                                 mofa_stream_t stream) {
    if (!x || !part || !gamma || !beta || !y || nframes <= 0 || HW <= 0 || frames_per_stat <= 0 ||
        nframes % frames_per_stat != 0 || C % 32 != 0 || C % 8 != 0 || C > 4096 || ldx % 8 != 0 || ldy % 8 != 0)
        return MOFA_EINVAL;                                  // C <= 4096 as mofa_gn_partial_f16 (C * 8 B of dynamic + 8.5 KB static LDS)
    const int nparts = mofa_gn_nparts(HW, C);
    const int rpw = gn_apply_rows_per_wg(HW, C, nframes, frames_per_stat * nparts);
    const int chunks = cdiv(HW, rpw);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, nframes), dim3(256), (size_t)C * 8, (hipStream_t)stream, (const f16*)x, part,
                       gamma, beta, (f16*)y, HW, C, ldx, ldy, frames_per_stat, frames_per_stat * nparts,
                       (double)frames_per_stat * (double)HW * (double)(C / 32), rpw, eps, silu);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// Frame-sharded clips (mofa_video_amd/parallel.py): the statistics set spans frames that live on other ranks.  Every rank
// all-gathers the PARTIALS (a few KB per rank: [frames][nparts][32][2] fp32, zero entries for the padding frames of the
// shorter shards) and this entry applies the normalisation of that ONE set to the `nframes` frames given -- the rank's own
// frames, or a halo frame received raw from a neighbour shard: every workgroup combines the `nentries` gathered entries
// itself in entry order (fp64: the same result on every rank), `count_per_group` = elements per group over the WHOLE clip.
extern "C" int mofa_gn_apply_gathered_f16(const void* x, const float* part_all, int nentries, double count_per_group,
                                          const float* gamma, const float* beta, void* y, int nframes, int HW, int C, int ldx,
                                          int ldy, float eps, int silu, mofa_stream_t stream) {
    if (!x || !part_all || !gamma || !beta || !y || nframes <= 0 || HW <= 0 || nentries <= 0 || nentries > 4096 ||
        count_per_group <= 0 || C % 32 != 0 || C % 8 != 0 || C > 4096 || ldx % 8 != 0 || ldy % 8 != 0)
        return MOFA_EINVAL;
    // (a rank's few frames against the whole clip's gathered entries: about 64 K elements per workgroup, at least 1 024 workgroups)
    int rpw = (65536 + C - 1) / C;
    while (rpw > 16 && (long long)cdiv(HW, rpw) * nframes < 1024) rpw = (rpw + 1) / 2;
    const int chunks = cdiv(HW, rpw);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(chunks, nframes), dim3(256), (size_t)C * 8, (hipStream_t)stream, (const f16*)x, part_all,
                       gamma, beta, (f16*)y, HW, C, ldx, ldy, nframes, nentries, count_per_group, rpw, eps, silu);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// LayerNorm: LPR lanes per token row (16, or 8 where C / 8 = 40 vectors divide evenly by 8 but not by 16: C = 320, the level-0
// width -- with 16 lanes the third vector of a row kept half the lanes idle), the row held in registers (C <= 1280); reductions
// are xor-shuffles inside the lane group.  gamma / beta are staged in LDS once per workgroup (r03: loading them per row was 4 x the
// row's own load instructions and bound the kernel on the CU's texture path at 3.6 TB/s).  r04b: the row stays PACKED (fp16, 4
// registers per vector instead of 8) unless a row vector is added first -- C = 1280 went from 160 to under 100 VGPRs, 3 -> 5 waves
// per SIMD --, and a workgroup walks `nrr` consecutive passes of 256 / LPR rows chosen by the launcher so that the grid is ONE round of
// resident workgroups (3 600 workgroups on 1 792 slots = 1.76 rounds left the second round 3/4 empty at level 0).
#define LN_MAXIT 10
template <int MAXIT, int LPR, bool RV>
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, f16* __restrict__ y, int M,
                                                        int C, int ldx, int ldy, float eps,
                                                        const float* __restrict__ rowvec, int rv_div, int rv_mod, int nrr) {
    __shared__ __attribute__((aligned(16))) float sG[LPR * 8 * MAXIT], sB[LPR * 8 * MAXIT];
    constexpr int GROUPS = 256 / LPR;                              // rows per pass
    const int lg = threadIdx.x & (LPR - 1);
    const int CV = C >> 3;
    for (int c = threadIdx.x; c < C; c += 256) { sG[c] = gamma[c]; sB[c] = beta[c]; }
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    const int row0 = (int)blockIdx.x * nrr * GROUPS + (int)(threadIdx.x / LPR);
#pragma unroll 1
    for (int rr = 0; rr < nrr; ++rr) {
        const int row = row0 + rr * GROUPS;
        if (row >= M) break;
        const f16* xp = x + (size_t)row * ldx;
        f16x8 a[MAXIT];
        float v[RV ? MAXIT : 1][8];                                // (row vector kinds: x + rowvec in fp32, as the reference adds it)
        const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int cv = lg + LPR * it;
            a[it] = cv < CV ? *(const f16x8*)(xp + cv * 8) : zero8;
        }
        float sum = 0.f;
        if constexpr (RV) {
            const float* rv = rowvec + (size_t)((row / rv_div) % rv_mod) * C;
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int cv = lg + LPR * it;
                if (cv < CV) {
                    const f32x4 r0 = *(const f32x4*)(rv + cv * 8), r1 = *(const f32x4*)(rv + cv * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = (float)a[it][e] + (e < 4 ? r0[e] : r1[e - 4]);
                        v[it][e] = t;
                        sum += t;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)a[it][e];     // (vectors beyond C are zero)
        }
        auto val = [&](int it, int e) __attribute__((always_inline)) -> float {
            if constexpr (RV) return v[it][e];
            else return (float)a[it][e];
        };
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum * inv_c;
        if constexpr (!RV) {                                       // keep the row PACKED between the passes: without this hipcc converts it
#pragma unroll                                                     // once and holds 8 fp32 registers per vector instead of 4
            for (int it = 0; it < MAXIT; ++it) asm volatile("" : "+v"(a[it]));
        }
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int cv = lg + LPR * it;
            if (cv < CV) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = val(it, e) - mean;
                    sq = fmaf(d, d, sq);
                }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq * inv_c + eps);
        if constexpr (!RV) {
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) asm volatile("" : "+v"(a[it]));
        }
        f16* yp = y + (size_t)row * ldy;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int cv = lg + LPR * it;
            if (cv < CV) {
                const f32x4 g0 = *(const f32x4*)(sG + cv * 8), g1 = *(const f32x4*)(sG + cv * 8 + 4);
                const f32x4 b0 = *(const f32x4*)(sB + cv * 8), b1 = *(const f32x4*)(sB + cv * 8 + 4);
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (f16)fmaf((val(it, e) - mean) * rstd, e < 4 ? g0[e] : g1[e - 4], e < 4 ? b0[e] : b1[e - 4]);
                *(f16x8*)(yp + cv * 8) = o;
            }
        }
    }
}

template <int MAXIT, int LPR>
static void launch_layernorm(const void* x, const float* gamma, const float* beta, void* y, int M, int C, int ldx, int ldy, float eps,
                             const float* rowvec, int rv_div, int rv_mod, int slots, hipStream_t st) {
    // one round of resident workgroups: slots = CUs x workgroups per CU of this instantiation (occupancy API, cached)
    static const int wpc_rv = [] {
        int nb = 0;
        return (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)layernorm_kernel<MAXIT, LPR, true>, 256, 0) == hipSuccess && nb > 0) ? nb : 4;
    }();
    static const int wpc_plain = [] {
        int nb = 0;
        return (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)layernorm_kernel<MAXIT, LPR, false>, 256, 0) == hipSuccess && nb > 0) ? nb : 4;
    }();
    const int wpc = rowvec ? wpc_rv : wpc_plain;
    slots = slots * wpc;
    const int passes = cdiv(M, 256 / LPR);
    const int nrr = cdiv(passes, slots);
    const int grid = cdiv(passes, nrr);
    if (rowvec)
        hipLaunchKernelGGL((layernorm_kernel<MAXIT, LPR, true>), dim3(grid), dim3(256), 0, st, (const f16*)x, gamma, beta, (f16*)y, M, C,
                           ldx, ldy, eps, rowvec, rv_div, rv_mod, nrr);
    else
        hipLaunchKernelGGL((layernorm_kernel<MAXIT, LPR, false>), dim3(grid), dim3(256), 0, st, (const f16*)x, gamma, beta, (f16*)y, M, C,
                           ldx, ldy, eps, rowvec, rv_div, rv_mod, nrr);
}

extern "C" int mofa_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int M, int C, int ldx,
                                  int ldy, float eps, const float* rowvec, int rv_div, int rv_mod, mofa_stream_t stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || C <= 0 || C % 8 != 0 || C > 16 * 8 * LN_MAXIT || ldx % 8 != 0 || ldy % 8 != 0)
        return MOFA_EINVAL;
    if (rowvec && (rv_div <= 0 || rv_mod <= 0)) return MOFA_EINVAL;
    const int CV = C / 8;
    static const int n_cu = [] {
        int dev = 0, cus = 0;
        return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                cus > 0) ? cus : 256;
    }();
    hipStream_t st = (hipStream_t)stream;
    if (CV <= 40 && CV % 8 == 0) launch_layernorm<5, 8>(x, gamma, beta, y, M, C, ldx, ldy, eps, rowvec, rv_div, rv_mod, n_cu, st);
    else if (CV <= 48) launch_layernorm<3, 16>(x, gamma, beta, y, M, C, ldx, ldy, eps, rowvec, rv_div, rv_mod, n_cu, st);
    else if (CV <= 80) launch_layernorm<5, 16>(x, gamma, beta, y, M, C, ldx, ldy, eps, rowvec, rv_div, rv_mod, n_cu, st);
    else launch_layernorm<LN_MAXIT, 16>(x, gamma, beta, y, M, C, ldx, ldy, eps, rowvec, rv_div, rv_mod, n_cu, st);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
