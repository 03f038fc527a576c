// Implicit-GEMM on MFMA for gfx950: every linear / 1x1 conv / 3x3 conv (s1, s2, nearest-2x input) /
// temporal (3,1,1) conv of the SVD UNet, the MOFA-Adapter trunk and the temporal VAE decoder.
//
//   out[m, n] = act( s_acc * (sum_tap sum_k X[src(m,tap), k] * W[n, tap*Cin + k] + bias[n] + rowvec[idx(m), n])
//                    + s1 * R1[m, n] + s2 * R2[m, n] )
//
// Output tile 128(M) x 128(N) per 256-thread workgroup (4 waves in 2x2, each 64x64 = 2x2 MFMA 32x32x16 f16
// tiles, fp32 accumulators).  The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile, so an
// accumulator lane owns one output row m (= lane & 31) and 4 consecutive output columns per register quad -> 8-byte
// epilogue loads/stores along the channel axis.  K is walked tap-major; zero padding of the convolution is realised
// by sourcing the rows of an out-of-image tap from a zero page.
//
// igemm_f16_kernel (default): the K loop is a 4-stage LDS ring filled by direct-to-LDS DMA
//   (global_load_lds_dwordx4, 16 B per lane, no VGPR staging), K step 32.  Three K tiles are kept in flight; each
//   iteration waits with a COUNTED s_waitcnt vmcnt (never a drain in steady state) + one raw s_barrier, issues the
//   tile three steps ahead, then runs 8 MFMAs per wave.  The LDS image is lane-linear (64-byte rows); bank conflicts
//   of the ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE address
//   (chunk ^= (row>>2)&3) and again on the read (cdna guide, rule 21).
// igemm_regstage_kernel: the first-generation variant (global -> VGPR -> LDS, 2 buffers, K step 64), kept for A/B
//   (MOFA_IGEMM_REGSTAGE=1).
#include <stdlib.h>

#include "common.h"

#define BM 128
#define BN 128

struct RowGeo {
    int img, oy, ox;  // conv3x3: image index and output pixel; convT3: oy = frame index within its clip
    int m;            // global output row (or -1 when beyond M)
};

__device__ __attribute__((aligned(128))) f16 g_zero_page[128];  // source of out-of-image taps / rows beyond M

__device__ __forceinline__ RowGeo make_geo(const mofa_igemm_args& a, int m) {
    RowGeo g;
    g.m = (m < a.M) ? m : -1;
    g.img = 0; g.oy = 0; g.ox = 0;
    if (a.mode == MOFA_MODE_CONV3X3) {
        const int hw = a.Hout * a.Wout;
        const int img = m / hw, rem = m - img * hw;
        g.img = img; g.oy = rem / a.Wout; g.ox = rem - g.oy * a.Wout;
    } else if (a.mode == MOFA_MODE_CONVT3) {
        g.oy = a.T > 0 ? (m / a.HW) % a.T : 0;
    }
    return g;
}

__device__ __forceinline__ const f16* x_src(const mofa_igemm_args& a, const RowGeo& g, int tap) {
    if (g.m < 0) return nullptr;
    const f16* x = (const f16*)a.x;
    if (a.mode == MOFA_MODE_PLAIN) {
        return x + (size_t)g.m * a.ldx;
    } else if (a.mode == MOFA_MODE_CONV3X3) {
        const int ks = a.ksize > 0 ? a.ksize : 3;
        const int ky = tap / ks, kx = tap - ky * ks;
        const int vy = g.oy * a.stride + ky - (ks >> 1);
        const int vx = g.ox * a.stride + kx - (ks >> 1);
        if (vy < 0 || vx < 0 || vy >= a.Hin * a.up || vx >= a.Win * a.up) return nullptr;
        const int iy = (a.up == 2) ? (vy >> 1) : vy;
        const int ix = (a.up == 2) ? (vx >> 1) : vx;
        return x + ((size_t)(g.img * a.Hin + iy) * a.Win + ix) * a.ldx;
    } else {  // MOFA_MODE_CONVT3
        const int tt = g.oy + tap - 1;
        if (a.T > 0 && (tt < 0 || tt >= a.T)) return nullptr;   // T == 0: unclipped, caller supplies halo frames
        return x + ((size_t)g.m + (size_t)(tap - 1) * a.HW) * a.ldx;
    }
}

// XCD-aware (bijective) workgroup remap: consecutive tile ids (same activation row block, successive weight column
// blocks) land on the same XCD so the activation tile is served from that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- epilogue -----------------------------------------------------------------------------------------------------
// The accumulator layout (lane = one row, 4 consecutive columns per register quad) would give 8-byte global accesses
// scattered over 32 rows per instruction.  Instead each wave transposes its 32 x (NJ*32) block through a private LDS
// slab (the K ring is free by then): phase 1 writes s_acc*(acc + bias + rowvec) in fp32 in fragment layout
// (ds_write_b128, rows padded to a 16-byte-odd stride), phase 2 re-reads it row-wise so that a lane owns 8 consecutive
// columns of one row: residuals are loaded and the result stored with 16-byte accesses, consecutive lanes covering
// consecutive 16-byte pieces of the same output row (whole 64/128-byte segments per row).
// (mrow0, ncol0) = origin of this wave's MI x NJ block of 32x32 accumulator tiles; slab = this wave's LDS slab.
#define EPI_COLS_MAX 64
#define EPI_STRIDE (EPI_COLS_MAX * 4 + 16)              // bytes per staged row (fp32) -- 272: conflict-free both ways
#define EPI_SLAB_BYTES (32 * EPI_STRIDE)

template <int MI, int NJ>
__device__ __forceinline__ void igemm_epilogue(const mofa_igemm_args& a, f32x16 (&acc)[MI][NJ], int mrow0, int ncol0,
                                               int lane, char* slab) {
    static_assert(NJ == 2, "epilogue slab is sized for 64 staged columns");
    const int l31 = lane & 31, lh = lane >> 5;
    const f16* r1 = (const f16*)a.r1;
    const f16* r2 = (const f16*)a.r2;
    f16* out = (f16*)a.out;
    const bool geglu = a.act == MOFA_ACT_GEGLU_PAIR;
    const int ocols = geglu ? 32 : 64;                   // staged output columns of this wave
    const int ocol0 = geglu ? ncol0 / 2 : ncol0;         // first output column
    const int nout = geglu ? a.N / 2 : a.N;              // number of valid output columns
    // 16-byte path needs every row start 16-byte aligned
    const bool wide = ((a.ldo & 7) == 0) && (!r1 || (a.ldr1 & 7) == 0) && (!r2 || (a.ldr2 & 7) == 0) &&
                      ((((size_t)a.out) & 15) == 0) && ((((size_t)a.r1) & 15) == 0) && ((((size_t)a.r2) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        // ---------------- phase 1: fragment layout -> slab ----------------
        {
            const int m = mrow0 + i * 32 + l31;
            const float* rv = nullptr;
            if (a.rowvec && m < a.M) {
                const int idx = ((m / a.rv_div) * a.rv_mul + (m % a.rv_mod_in)) % a.rv_mod_out;
                rv = a.rowvec + (size_t)idx * a.N;
            }
            if (geglu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nv = ncol0 + 8 * q + 4 * lh;          // value column in the interleaved N space
                    f32x4 bv = {0, 0, 0, 0}, bg = {0, 0, 0, 0};
                    if (a.bias && nv < a.N) { bv = *(const f32x4*)(a.bias + nv); bg = *(const f32x4*)(a.bias + nv + 32); }
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = a.s_acc * (acc[i][0][4 * q + e] + bv[e]);
                        const float g = a.s_acc * (acc[i][1][4 * q + e] + bg[e]);
                        o[e] = v * gelu_erf_f(g);
                    }
                    *(f32x4*)(slab + l31 * EPI_STRIDE + (8 * q + 4 * lh) * 4) = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = ncol0 + j * 32 + 8 * q + 4 * lh;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                        if (n < a.N) {
                            if (a.bias) { const f32x4 b = *(const f32x4*)(a.bias + n); v += b; }
                            if (rv) { const f32x4 b = *(const f32x4*)(rv + n); v += b; }
                        }
                        v *= a.s_acc;
                        *(f32x4*)(slab + l31 * EPI_STRIDE + (j * 32 + 8 * q + 4 * lh) * 4) = v;
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave DS ops are ordered; make the data visible
        // ---------------- phase 2: row-wise, 8 columns per lane per pass ----------------
        {
            const int cpr = ocols >> 3;                      // 16-byte (8-column) pieces per row: 8 or 4
            const int rpp = 64 / cpr;                        // rows covered per pass: 8 or 16
            const int lrow = lane / cpr, lcol = (lane - lrow * cpr) * 8;
            for (int r0 = 0; r0 < 32; r0 += rpp) {
                const int row = r0 + lrow;
                const int m = mrow0 + i * 32 + row;
                const int n = ocol0 + lcol;
                if (m >= a.M || n >= nout) continue;
                const f32x4 v0 = *(const f32x4*)(slab + row * EPI_STRIDE + lcol * 4);
                const f32x4 v1 = *(const f32x4*)(slab + row * EPI_STRIDE + lcol * 4 + 16);
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                const bool full = wide && (n + 8 <= nout);
                const int cnt = (n + 8 <= nout) ? 8 : 4;     // N % 4 == 0: a piece is whole, half, or empty
                if (r1) {
                    const f16* p = r1 + (size_t)m * a.ldr1 + n;
                    if (full) {
                        const f16x8 t = *(const f16x8*)p;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += a.s1 * (float)t[e];
                    } else {
                        const f16x4 t = *(const f16x4*)p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += a.s1 * (float)t[e];
                        if (cnt == 8) {
                            const f16x4 u = *(const f16x4*)(p + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 + e] += a.s1 * (float)u[e];
                        }
                    }
                }
                if (r2) {
                    const f16* p = r2 + (size_t)m * a.ldr2 + n;
                    if (full) {
                        const f16x8 t = *(const f16x8*)p;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += a.s2 * (float)t[e];
                    } else {
                        const f16x4 t = *(const f16x4*)p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += a.s2 * (float)t[e];
                        if (cnt == 8) {
                            const f16x4 u = *(const f16x4*)(p + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 + e] += a.s2 * (float)u[e];
                        }
                    }
                }
                if (a.act == MOFA_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                } else if (a.act == MOFA_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                f16* po = out + (size_t)m * a.ldo + n;
                if (full) {
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
                    *(f16x8*)po = o;
                } else {
                    f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                    *(f16x4*)po = o;
                    if (cnt == 8) {
                        f16x4 o2 = {(f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
                        *(f16x4*)(po + 4) = o2;
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // slab is rewritten by the next m-tile
    }
}

// =====================================================================================================================
// default kernel: LDS ring fed by global_load_lds, counted vmcnt.  Templated on the wave grid (WM x WN waves) and the
// per-wave block of MFMA tiles (MI x NJ of 32x32): <2,2,2,2> = 128x128 tile / 256 threads, <2,4,4,2> = 256x256 / 512.
// =====================================================================================================================
__device__ __forceinline__ void glds16(const f16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BKS = K per ring stage (32: 64-byte LDS rows, one DMA instruction = 16 rows; 64: 128-byte rows = whole cache lines,
// one DMA instruction = 8 rows).  NST = ring stages (NST-1 tiles in flight).
template <int WM, int WN, int MI, int NJ, int BKS, int NST>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_f16_kernel(const mofa_igemm_args a, const int tilesN,
                                                                    const int nwg) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
    constexpr int RB = BKS * 2;                                    // LDS row bytes
    constexpr int RPI = 1024 / RB;                                 // rows per DMA instruction
    constexpr int SPR = RB / 16;                                   // 16-byte slots per row
    constexpr int SWS = (BKS == 32) ? 2 : 1;                       // swizzle term = (row >> SWS) & (SPR-1)
    constexpr int XI = TBM / RPI / NW, WI = TBN / RPI / NW;        // DMA instructions per wave per stage
    constexpr int SXB = TBM * RB, STB = SXB + TBN * RB;            // stage bytes: X part, total
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY shared object (cdna guide trap (a))

    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid / tilesN, tn = bid - tm * tilesN;
    const int m0 = tm * TBM, n0 = tn * TBN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = (a.mode == MOFA_MODE_CONV3X3) ? (a.ksize > 0 ? a.ksize * a.ksize : 9) : (a.mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const int kpt = a.Cin / BKS;         // K stages per tap
    const int nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;

    // ---- DMA mapping: one global_load_lds instruction fills RPI rows (lane -> row lane / SPR, slot lane % SPR).
    //      The slot a lane fills holds source chunk  c = slot ^ ((row >> SWS) & (SPR-1))  (swizzle on the SOURCE
    //      address; the LDS image stays lane-linear) -> conflict-free ds_read_b128 fragment reads.
    RowGeo geo[XI];
    int xoff[XI];
    const f16* wsrc[WI];
#pragma unroll
    for (int q = 0; q < XI; ++q) {
        const int row = (wave * XI + q) * RPI + lane / SPR;
        geo[q] = make_geo(a, m0 + row);
        xoff[q] = ((lane % SPR) ^ ((row >> SWS) & (SPR - 1))) * 8;
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) {
        const int row = (wave * WI + q) * RPI + lane / SPR;
        int n = n0 + row;
        n = n < a.N ? n : a.N - 1;
        wsrc[q] = (const f16*)a.w + (size_t)n * Ktot + ((lane % SPR) ^ ((row >> SWS) & (SPR - 1))) * 8;
    }
    const f16* xs[XI];
#pragma unroll
    for (int q = 0; q < XI; ++q) xs[q] = nullptr;
    int itap = 0, ikc = 0;   // position of the NEXT tile to issue

    auto issue = [&](int ks, int stage) {
        if (ikc == 0) {
#pragma unroll
            for (int q = 0; q < XI; ++q) xs[q] = x_src(a, geo[q], itap);
        }
        char* sb = smem + stage * STB;
#pragma unroll
        for (int q = 0; q < XI; ++q) {
            const f16* s = xs[q] ? xs[q] + ikc * BKS + xoff[q] : (const f16*)g_zero_page;
            glds16(s, sb + (wave * XI + q) * 1024);
        }
#pragma unroll
        for (int q = 0; q < WI; ++q) glds16(wsrc[q] + (size_t)ks * BKS, sb + SXB + (wave * WI + q) * 1024);
        if (++ikc == kpt) { ikc = 0; ++itap; }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int D = NST - 1;
    constexpr int OPS = XI + WI;                          // DMA ops per tile per wave
    for (int t = 0; t < D && t < nk; ++t) issue(t, t);

    const int fsw = (l31 >> SWS) & (SPR - 1);             // read-side swizzle term is lane-constant
    const int xrow = (wm * MI * 32 + l31) * RB;           // byte offsets of this lane's fragment rows
    const int wrow = SXB + (wn * NJ * 32 + l31) * RB;

    for (int ks = 0; ks < nk; ++ks) {
        // tile ks must have landed; tiles ks+1 .. ks+D-1 (if they exist) may stay in flight
        const int rem = nk - 1 - ks;
        if (D >= 3 && rem >= 2) wait_vmcnt<2 * OPS>();
        else if (D >= 2 && rem >= 1) wait_vmcnt<OPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                     // all waves' parts of tile ks landed; buffer (ks-1)%NST is free
        if (ks + D < nk) issue(ks + D, (ks + D) % NST);
        const char* sb = smem + (ks % NST) * STB;
#pragma unroll
        for (int kk = 0; kk < BKS / 16; ++kk) {
            const int slot = ((kk * 2 + lh) ^ fsw) * 16;
            f16x8 xf[MI], wf[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) xf[i] = *(const f16x8*)(sb + xrow + i * 32 * RB + slot);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[j] = *(const f16x8*)(sb + wrow + j * 32 * RB + slot);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                      // every wave is done with the ring: reuse it as epilogue slabs
    igemm_epilogue<MI, NJ>(a, acc, m0 + wm * MI * 32, n0 + wn * NJ * 32, lane, smem + wave * EPI_SLAB_BYTES);
}

// =====================================================================================================================
// first-generation kernel: register-staged, 2 LDS buffers, K step 64 (kept for A/B: MOFA_IGEMM_REGSTAGE=1)
// =====================================================================================================================
#define BK 64
#define LDSS 72  // LDS row stride in halves (64 + 8 pad) = 144 B
#define REGSTAGE_LDS_BYTES (2 * (BM + BN) * LDSS * 2)

__global__ __launch_bounds__(256, 2) void igemm_regstage_kernel(const mofa_igemm_args a, const int tilesN, const int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16* sX = (f16*)smem;             // [2][BM][LDSS]
    f16* sW = sX + 2 * BM * LDSS;     // [2][BN][LDSS]

    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid / tilesN, tn = bid - tm * tilesN;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = (a.mode == MOFA_MODE_CONV3X3) ? (a.ksize > 0 ? a.ksize * a.ksize : 9) : (a.mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const int kpt = a.Cin / BK;
    const int nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;

    const int lcol = tid & 7, lrow = tid >> 3;
    RowGeo geo[4];
    const f16* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        geo[i] = make_geo(a, m0 + lrow + 32 * i);
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
    igemm_epilogue<2, 2>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * EPI_SLAB_BYTES);
}

extern "C" int mofa_igemm_f16(const mofa_igemm_args* a, mofa_stream_t stream) {
    if (!a || !a->x || !a->w || !a->out) return MOFA_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->Cin <= 0) return MOFA_EINVAL;
    if (a->Cin % BK != 0 || a->N % 4 != 0) return MOFA_EINVAL;
    if (a->mode < 0 || a->mode > 2) return MOFA_EINVAL;
    if (a->mode == MOFA_MODE_CONV3X3) {
        if (a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0) return MOFA_EINVAL;
        if ((a->stride != 1 && a->stride != 2) || (a->up != 1 && a->up != 2)) return MOFA_EINVAL;
        if (a->ksize != 0 && a->ksize != 3 && a->ksize != 7) return MOFA_EINVAL;
        if (a->M % (a->Hout * a->Wout) != 0) return MOFA_EINVAL;
    }
    if (a->mode == MOFA_MODE_CONVT3 && (a->T < 0 || a->HW <= 0 || (a->T > 0 && a->M % (a->T * a->HW) != 0)))
        return MOFA_EINVAL;
    if (a->rowvec && (a->rv_div <= 0 || a->rv_mod_in <= 0 || a->rv_mod_out <= 0)) return MOFA_EINVAL;
    if (a->act == MOFA_ACT_GEGLU_PAIR && (a->N % 64 != 0 || a->r1 || a->r2 || a->rowvec)) return MOFA_EINVAL;
    if (a->ldx % 8 != 0 || a->ldo % 4 != 0) return MOFA_EINVAL;
    if ((a->r1 && a->ldr1 % 4 != 0) || (a->r2 && a->ldr2 % 4 != 0)) return MOFA_EINVAL;

    // kernel configurations of the LDS-DMA ring: {tile, K per stage, stages}
    typedef void (*kern_t)(const mofa_igemm_args, const int, const int);
    struct Cfg { kern_t k; int tm, tn, threads, lds; };
    static const Cfg cfgs[] = {
        {igemm_f16_kernel<2, 2, 2, 2, 32, 4>, 128, 128, 256, 4 * 256 * 64},    // 0: 128^2, 64-B rows, 3 tiles in flight
        {igemm_f16_kernel<2, 4, 4, 2, 32, 4>, 256, 256, 512, 4 * 512 * 64},    // 1: 256^2, 64-B rows, 3 tiles in flight
        {igemm_f16_kernel<2, 2, 2, 2, 64, 2>, 128, 128, 256, 2 * 256 * 128},   // 2: 128^2, 128-B rows, 1 tile in flight
        {igemm_f16_kernel<2, 4, 4, 2, 64, 2>, 256, 256, 512, 2 * 512 * 128},   // 3: 256^2, 128-B rows, 1 tile in flight
        {igemm_f16_kernel<2, 2, 2, 2, 64, 3>, 128, 128, 256, 3 * 256 * 128},   // 4: 128^2, 128-B rows, 2 tiles in flight
        {igemm_f16_kernel<4, 2, 2, 2, 64, 3>, 256, 128, 512, 3 * 384 * 128},   // 5: 256x128, 128-B rows, 2 tiles in flight
        {igemm_f16_kernel<2, 4, 2, 2, 64, 3>, 128, 256, 512, 3 * 384 * 128},   // 6: 128x256, 128-B rows, 2 tiles in flight
    };
    static int variant = -1;   // -2 = register-staged kernel, -1 unset, otherwise forced cfg (or 100 = auto)
    if (variant == -1) {
        const char* e = getenv("MOFA_IGEMM_REGSTAGE");
        const char* e2 = getenv("MOFA_IGEMM_CFG");
        variant = (e && e[0] == '1') ? -2 : (e2 ? atoi(e2) : 100);
        for (const Cfg& c : cfgs)
            if (hipFuncSetAttribute((const void*)c.k, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds) != hipSuccess) {
                variant = -1;
                return MOFA_ELAUNCH;
            }
        if (hipFuncSetAttribute((const void*)igemm_regstage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                REGSTAGE_LDS_BYTES) != hipSuccess) {
            variant = -1;
            return MOFA_ELAUNCH;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (variant == -2) {
        const int tilesM = cdiv(a->M, BM), tilesN = cdiv(a->N, BN);
        hipLaunchKernelGGL(igemm_regstage_kernel, dim3(tilesM * tilesN), dim3(256), REGSTAGE_LDS_BYTES, st, *a, tilesN,
                           tilesM * tilesN);
    } else {
        int ci = variant;
        if (variant == 100) {
            // 256x256 tiles halve the global->LDS fill traffic per flop; use them when the column count fills them
            // (<= 1/8 padding waste) and the grid still covers the chip
            // measured on MI355X (profiles/r01_igemm_config_sweep.md): 128-byte LDS rows (whole cache lines per DMA
            // row) beat the deeper 64-byte-row ring everywhere; 256x256 tiles win when N is wide relative to K
            // (GEGLU / QKV projections), 128x128 (2 workgroups per CU) otherwise.
            const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
            const long long Ktot = (long long)taps * a->Cin;
            const int t256m = cdiv(a->M, 256), t256n = cdiv(a->N, 256);
            const bool big = (long long)t256n * 256 * 8 <= (long long)a->N * 9 && t256m * t256n >= 256 &&
                             (long long)a->N >= 2 * Ktot;
            ci = big ? 3 : 2;
        }
        const Cfg& c = cfgs[ci];
        const int tilesM = cdiv(a->M, c.tm), tilesN = cdiv(a->N, c.tn);
        hipLaunchKernelGGL(c.k, dim3(tilesM * tilesN), dim3(c.threads), c.lds, st, *a, tilesN, tilesM * tilesN);
    }
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
