// Implicit-GEMM on MFMA for gfx950: every linear / 1x1 conv / 3x3 conv (s1, s2, nearest-2x input) /
// temporal (3,1,1) conv of the SVD UNet, the MOFA-Adapter trunk and the temporal VAE decoder.
//
//   out[m, n] = act( s_acc * (sum_tap sum_k X[src(m,tap), k] * W[n, tap*Cin + k] + bias[n] + rowvec[idx(m), n])
//                    + s1 * R1[m, n] + s2 * R2[m, n] )
//
// Tiling: 128(M) x 128(N) x 64(K) per 256-thread workgroup (4 waves in 2x2, each 64x64 = 2x2 MFMA
// 32x32x16 tiles, fp32 accumulators).  The MFMA "A" operand is the WEIGHT tile and the "B" operand the
// ACTIVATION tile, so an accumulator lane owns one output row m (= lane & 31) and 4 consecutive
// output columns per register quad -> 8-byte epilogue loads/stores along the channel axis.
// K is walked tap-major: zero padding of the convolution is realised by zero-filling the staged
// activation rows of an out-of-image tap.  Global -> register -> LDS staging, double-buffered LDS
// (rows padded to 144 B: ds_read_b128 fragment reads are bank-conflict free), one barrier per K step,
// next tile's global loads issued before the current tile's MFMAs.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define LDSS 72  // LDS row stride in halves (64 + 8 pad) = 144 B
#define IGEMM_LDS_BYTES (2 * (BM + BN) * LDSS * 2)

struct RowGeo {
    int img, oy, ox;  // conv3x3: image index and output pixel; convT3: (unused, t, unused)
    int m;            // global output row (or -1 when beyond M)
};

__device__ __forceinline__ const f16* x_src(const mofa_igemm_args& a, const RowGeo& g, int tap) {
    if (g.m < 0) return nullptr;
    const f16* x = (const f16*)a.x;
    if (a.mode == MOFA_MODE_PLAIN) {
        return x + (size_t)g.m * a.ldx;
    } else if (a.mode == MOFA_MODE_CONV3X3) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int vy = g.oy * a.stride + ky - 1;
        const int vx = g.ox * a.stride + kx - 1;
        if (vy < 0 || vx < 0 || vy >= a.Hin * a.up || vx >= a.Win * a.up) return nullptr;
        const int iy = (a.up == 2) ? (vy >> 1) : vy;
        const int ix = (a.up == 2) ? (vx >> 1) : vx;
        return x + ((size_t)(g.img * a.Hin + iy) * a.Win + ix) * a.ldx;
    } else {  // MOFA_MODE_CONVT3
        const int tt = g.oy + tap - 1;
        if (tt < 0 || tt >= a.T) return nullptr;
        return x + ((size_t)g.m + (size_t)(tap - 1) * a.HW) * a.ldx;
    }
}

__global__ __launch_bounds__(256, 2) void igemm_f16_kernel(const mofa_igemm_args a, const int tilesN, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* sX = (f16*)smem;             // [2][BM][LDSS]
    f16* sW = sX + 2 * BM * LDSS;     // [2][BN][LDSS]

    // XCD-aware (bijective) workgroup remap: consecutive tile ids (same activation row block, successive
    // weight column blocks) land on the same XCD so the activation tile is served from that XCD's L2.
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tilesN, tn = bid - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = (a.mode == MOFA_MODE_CONV3X3) ? 9 : (a.mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const int kpt = a.Cin / BK;          // K steps per tap
    const int nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;

    // ---- loader mapping: thread -> (row lrow + 32*i, 16-byte chunk lcol) -------------------------------
    const int lcol = tid & 7, lrow = tid >> 3;
    RowGeo geo[4];
    const f16* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lrow + 32 * i;
        geo[i].m = (m < a.M) ? m : -1;
        geo[i].img = 0; geo[i].oy = 0; geo[i].ox = 0;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int hw = a.Hout * a.Wout;
            const int img = m / hw, rem = m - img * hw;
            geo[i].img = img; geo[i].oy = rem / a.Wout; geo[i].ox = rem - geo[i].oy * a.Wout;
        } else if (a.mode == MOFA_MODE_CONVT3) {
            geo[i].oy = (m / a.HW) % a.T;
        }
        int n = n0 + lrow + 32 * i;
        n = n < a.N ? n : a.N - 1;
        wrow[i] = (const f16*)a.w + (size_t)n * Ktot + lcol * 8;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const f16* xs[4];
    f16x8 gx[4], gw[4];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tile = [&](int ks) {
        const int tap = ks / kpt, kc = ks - tap * kpt;
        if (kc == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[i] = x_src(a, geo[i], tap);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gx[i] = xs[i] ? *(const f16x8*)(xs[i] + kc * BK + lcol * 8) : zero8;
            gw[i] = *(const f16x8*)(wrow[i] + (size_t)ks * BK);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f16x8*)&sX[(buf * BM + lrow + 32 * i) * LDSS + lcol * 8] = gx[i];
            *(f16x8*)&sW[(buf * BN + lrow + 32 * i) * LDSS + lcol * 8] = gw[i];
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tile(ks + 1);
        const f16* bx = sX + (buf * BM + wm * 64 + l31) * LDSS + lh * 8;
        const f16* bw = sW + (buf * BN + wn * 64 + l31) * LDSS + lh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f16x8 xf[2], wf[2];
            xf[0] = *(const f16x8*)(bx + kk * 16);
            xf[1] = *(const f16x8*)(bx + 32 * LDSS + kk * 16);
            wf[0] = *(const f16x8*)(bw + kk * 16);
            wf[1] = *(const f16x8*)(bw + 32 * LDSS + kk * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns row m = l31 (per m-tile) and columns 8q + 4*lh + e (per n-tile) -----------
    const f16* r1 = (const f16*)a.r1;
    const f16* r2 = (const f16*)a.r2;
    f16* out = (f16*)a.out;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wm * 64 + i * 32 + l31;
        if (m >= a.M) continue;
        const float* rv = nullptr;
        if (a.rowvec) {
            const int idx = ((m / a.rv_div) * a.rv_mul + (m % a.rv_mod_in)) % a.rv_mod_out;
            rv = a.rowvec + (size_t)idx * a.N;
        }
        if (a.act == MOFA_ACT_GEGLU_PAIR) {
            // n-tile 0 of the wave = value columns, n-tile 1 = the matching gate columns (weights interleaved by 32)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nv = n0 + wn * 64 + 8 * q + 4 * lh;  // value column in the interleaved N space
                if (nv >= a.N) continue;
                const int ng = nv + 32;
                const int no = (n0 + wn * 64) / 2 + 8 * q + 4 * lh;
                f32x4 bv = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
                if (a.bias) { bv = *(const f32x4*)(a.bias + nv); bg = *(const f32x4*)(a.bias + ng); }
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = a.s_acc * (acc[i][0][4 * q + e] + bv[e]);
                    const float g = a.s_acc * (acc[i][1][4 * q + e] + bg[e]);
                    o[e] = (f16)(v * gelu_erf_f(g));
                }
                *(f16x4*)(out + (size_t)m * a.ldo + no) = o;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * lh;
                if (n >= a.N) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (a.bias) { const f32x4 b = *(const f32x4*)(a.bias + n); v += b; }
                if (rv) { const f32x4 b = *(const f32x4*)(rv + n); v += b; }
                v *= a.s_acc;
                if (r1) {
                    const f16x4 t = *(const f16x4*)(r1 + (size_t)m * a.ldr1 + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a.s1 * (float)t[e];
                }
                if (r2) {
                    const f16x4 t = *(const f16x4*)(r2 + (size_t)m * a.ldr2 + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += a.s2 * (float)t[e];
                }
                if (a.act == MOFA_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)v[e];
                *(f16x4*)(out + (size_t)m * a.ldo + n) = o;
            }
        }
    }
}

extern "C" int mofa_igemm_f16(const mofa_igemm_args* a, mofa_stream_t stream) {
    if (!a || !a->x || !a->w || !a->out) return MOFA_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->Cin <= 0) return MOFA_EINVAL;
    if (a->Cin % BK != 0 || a->N % 4 != 0) return MOFA_EINVAL;
    if (a->mode < 0 || a->mode > 2) return MOFA_EINVAL;
    if (a->mode == MOFA_MODE_CONV3X3) {
        if (a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0) return MOFA_EINVAL;
        if ((a->stride != 1 && a->stride != 2) || (a->up != 1 && a->up != 2)) return MOFA_EINVAL;
        if (a->M % (a->Hout * a->Wout) != 0) return MOFA_EINVAL;
    }
    if (a->mode == MOFA_MODE_CONVT3 && (a->T <= 0 || a->HW <= 0 || a->M % (a->T * a->HW) != 0)) return MOFA_EINVAL;
    if (a->rowvec && (a->rv_div <= 0 || a->rv_mod_in <= 0 || a->rv_mod_out <= 0)) return MOFA_EINVAL;
    if (a->act == MOFA_ACT_GEGLU_PAIR && (a->N % 64 != 0 || a->r1 || a->r2 || a->rowvec)) return MOFA_EINVAL;
    if (a->ldx % 8 != 0 || a->ldo % 4 != 0) return MOFA_EINVAL;
    if ((a->r1 && a->ldr1 % 4 != 0) || (a->r2 && a->ldr2 % 4 != 0)) return MOFA_EINVAL;

    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)igemm_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                IGEMM_LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
        attr_set = true;
    }
    const int tilesM = cdiv(a->M, BM), tilesN = cdiv(a->N, BN);
    const int nwg = tilesM * tilesN;
    hipLaunchKernelGGL(igemm_f16_kernel, dim3(nwg), dim3(256), IGEMM_LDS_BYTES, (hipStream_t)stream, *a, tilesN, nwg);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
