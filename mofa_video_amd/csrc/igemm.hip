// Implicit-GEMM on MFMA for gfx950: every linear / 1x1 conv / kxk conv (s1, s2, nearest-2x input) /
// temporal (3,1,1) conv of the SVD UNet, the MOFA-Adapter trunk and the temporal VAE decoder.
//
//   out[m, n] = act( s_acc * (sum_tap sum_k X[src(m,tap), k] * W[n, tap*Cin + k] + bias[n] + rowvec[idx(m), n])
//                    + s1 * R1[m, n] + s2 * R2[m, n] )
//
// One PERSISTENT workgroup per CU slot walks output tiles (128x128 or 192x128 with 4 waves, 2 workgroups per CU; the
// 8-wave 256x256 tile is igemm8.hip; the launcher picks by a small cost model).  Each wave owns MI x 2 MFMA 32x32x16 f16
// tiles with fp32 accumulators.  The MFMA "A" operand is
// the WEIGHT tile and the "B" operand the ACTIVATION tile, so an accumulator lane owns one output row.  K is walked
// tap-major in steps of 64 (128-byte LDS rows = whole cache lines per row); zero padding of the convolution is realised
// by sourcing the rows of an out-of-image tap from a zero page.
//
// K loop: a 2-stage LDS ring filled by direct-to-LDS DMA (global_load_lds_dwordx4, 16 B per lane, no VGPR staging), the
// DMA instructions of the next stage interleaved with the four MFMA groups of the current one.
// The LDS image is lane-linear; bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle applied
// on the SOURCE address and again on the read (cdna guide, rule 21).  The stream of stages runs ACROSS tiles: the last
// K step of a tile issues stage 0 of the workgroup's next tile, so the DMA latency of the next tile hides under the
// epilogue of the current one.  The first wait of the next tile is a COUNTED s_waitcnt vmcnt(T), T = the VMEM
// operations the epilogue issued after that DMA (its output stores are not drained); T is exact only for interior
// tiles on the 16-byte path, every other tile waits with vmcnt(0).
//
// Epilogue: each wave transposes its 32 x 64 accumulator block through a private 8 KB XOR-swizzled LDS slab (inside the
// ring stage the K loop has just finished with) so that a lane owns 8 consecutive columns of one row; bias, row vector,
// residuals, GEGLU product and activation are applied in that layout with 16-byte global accesses.  Residual /
// row-vector loads are issued a group of passes at a time, on 128-row tiles one group ahead of the stores (vmcnt retires
// in order on gfx9: a load issued before a store never waits for it).  The kernel is templated on the epilogue kind
// (residuals / row vector / GEGLU) so the memory operations per pass are static.
// Tile choice: mofa_igemm_args.tile (MOFA_TILE_*; 0 = the launcher's cost model); the environment variable
// MOFA_IGEMM_CFG=2|4|5 forces the 128x128 | 192x128 | 256x256 phase-pipelined (igemm8.hip) tile for every launch that
// leaves `tile` at 0.
// Earlier variants (register staging, 64-byte rows, deeper rings, 256x128 tiles) and their measurements:
// profiles/archive/r01_igemm_config_sweep.md.
#include <stdlib.h>

#include "igemm_common.h"


// wait until at most t VMEM operations are outstanding, t rounded DOWN to an encodable step (waiting longer is safe)
__device__ __forceinline__ void wait_vmcnt_le(int t) {
    if (t >= 48) wait_vmcnt<48>();
    else if (t >= 32) wait_vmcnt<32>();
    else if (t >= 24) wait_vmcnt<24>();
    else if (t >= 16) wait_vmcnt<16>();
    else if (t >= 8) wait_vmcnt<8>();
    else if (t >= 4) wait_vmcnt<4>();
    else wait_vmcnt<0>();
}

// ---- epilogue -----------------------------------------------------------------------------------------------------
// slab: 32 rows x 64 fp32 columns = 256 B per row, 16-byte chunk c of row r stored at chunk c ^ (r & 15): the
// fragment-layout ds_write_b128 (lanes = 32 rows x 2 adjacent chunks) and the row-wise ds_read_b128 (lanes = 8 rows x
// 8 even chunks) are both bank-conflict free.
#define EPI_SLAB_BYTES 8192
__device__ __forceinline__ int slab_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

struct PassLoads {
    f16x8 t1, t2;
    f32x4 rv0, rv1;
};

// lane-constant bias registers: plain: bias[n .. n+8); GEGLU: value columns [0..1], gate columns [2..3].
// Loaded one tile AHEAD (before the prologue / after the previous tile's epilogue) and "touched" (bias_touch) at the
// tile's last K step, where the K loop's own vmcnt(0) has just drained everything: hipcc's waitcnt insertion then
// sees them as complete and puts no vmcnt(0) into the epilogue (which would also drain the next tile's DMA).
template <int EPI>
__device__ __forceinline__ void load_bias(const mofa_igemm_args& a, int ncol0, int lane, f32x4 (&b)[4]) {
    constexpr bool GEGLU = (EPI & EPI_GEGLU) != 0;
    const f32x4 z = {0, 0, 0, 0};
    b[0] = z; b[1] = z; b[2] = z; b[3] = z;
    if (!a.bias) return;
    if (GEGLU) {
        // value / gate rows interleaved in blocks of 16 (weights.interleave_geglu): this lane's 8 outputs (lane & 3) * 8 .. + 7
        // of the wave's 32 have their value columns at 32 t + o, their gate columns 16 further (t = tile, o = 0 or 8)
        int nv = ncol0 + ((lane & 3) >> 1) * 32 + (lane & 1) * 8;  // N is a multiple of 64 here
        nv = nv < a.N ? nv : 0;                                    // columns beyond N are never stored
        b[0] = *(const f32x4*)(a.bias + nv); b[1] = *(const f32x4*)(a.bias + nv + 4);
        b[2] = *(const f32x4*)(a.bias + nv + 16); b[3] = *(const f32x4*)(a.bias + nv + 20);
    } else {
        const int n = ncol0 + (lane & 7) * 8;
        const int n_lo = n + 4 <= a.N ? n : 0;                     // columns beyond N are never stored
        const int n_hi = n + 8 <= a.N ? n + 4 : n_lo;
        b[0] = *(const f32x4*)(a.bias + n_lo); b[1] = *(const f32x4*)(a.bias + n_hi);
    }
}
template <int EPI>
__device__ __forceinline__ void bias_touch(f32x4 (&b)[4]) {
    if (EPI & EPI_GEGLU) asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    else asm volatile("" : "+v"(b[0]), "+v"(b[1]));
}

// (mrow0, ncol0) = origin of this wave's MI x 2 block of 32x32 accumulator tiles; slab = this wave's LDS slab.
// WIDE: every row start of out / r1 / r2 is 16-byte aligned and N % 8 == 0 (a piece is whole or empty).
template <int MI, int EPI, bool WIDE>
__device__ __forceinline__ void igemm_epilogue(const mofa_igemm_args& a, f32x16 (&acc)[MI][2], const int mrow0,
                                               const int ncol0, const int lane, char* slab, f32x4 (&bias)[4]) {
    constexpr bool GEGLU = (EPI & EPI_GEGLU) != 0, R1 = (EPI & EPI_R1) != 0, R2 = (EPI & EPI_R2) != 0,
                   RV = (EPI & EPI_RV) != 0;
    constexpr int CPR = GEGLU ? 4 : 8;          // 8-column pieces per output row of this wave
    constexpr int RPP = 64 / CPR;               // rows per pass
    constexpr int P = 32 / RPP;                 // passes per 32-row accumulator tile
    constexpr int S = MI * P;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lrow = lane / CPR, lc = lane % CPR;
    const f16* r1 = (const f16*)a.r1;
    const f16* r2 = (const f16*)a.r2;
    f16* out = (f16*)a.out;
    const int nout = GEGLU ? a.N / 2 : a.N;
    const int n = (GEGLU ? ncol0 / 2 : ncol0) + lc * 8;       // first of this lane's 8 output columns
    const bool c8 = n + 8 <= nout, c4 = n + 4 <= nout;        // N % 4 == 0: a piece is whole, half or empty

    // Residual / row-vector loads: UNCONDITIONAL 16-byte (8-byte off the wide path) loads from clamped addresses (rows
    // beyond M and pieces beyond N are never stored), G passes at a time (a whole 32-row accumulator tile when the
    // registers allow), waited for explicitly (settle) before the first consumer.  On 128-row tiles the next group is
    // loaded before this group's stores go out (vmcnt retires in order: those loads never wait for the stores).
    constexpr int RPPASS = (R1 ? 4 : 0) + (R2 ? 4 : 0) + (RV ? 8 : 0);      // VGPRs of one pass's loads
    constexpr int BUDGET = MI == 2 ? 64 : 16;                               // the 192 / 256-row tiles have few spare VGPRs
    constexpr int G = RPPASS == 0 ? P : (RPPASS * P <= BUDGET ? P : (2 * RPPASS <= BUDGET && P % 2 == 0 ? 2 : 1));
    constexpr int NG = P / G;                                               // load groups per accumulator tile
    constexpr bool AHEAD = MI == 2 && RPPASS > 0 && 2 * RPPASS * G <= BUDGET;
    const int n_lo = c4 ? n : 0, n_hi = c8 ? n + 4 : n_lo;
    auto issue_group_loads = [&](int grp, PassLoads (&L)[G]) {             // grp = i * NG + gi
#pragma unroll
        for (int p = 0; p < G; ++p) {
            int m = mrow0 + (grp / NG) * 32 + ((grp % NG) * G + p) * RPP + lrow;
            m = m < a.M ? m : a.M - 1;
            if (R1) {
                const f16* q = r1 + (size_t)m * a.ldr1;
                if (WIDE) L[p].t1 = *(const f16x8*)(q + n_lo);
                else {
                    const f16x4 lo = *(const f16x4*)(q + n_lo), hi = *(const f16x4*)(q + n_hi);
                    L[p].t1 = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            if (R2) {
                const f16* q = r2 + (size_t)m * a.ldr2;
                if (WIDE) L[p].t2 = *(const f16x8*)(q + n_lo);
                else {
                    const f16x4 lo = *(const f16x4*)(q + n_lo), hi = *(const f16x4*)(q + n_hi);
                    L[p].t2 = (f16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
            if (RV) {
                const int idx = ((m / a.rv_div) * a.rv_mul + (m % a.rv_mod_in)) % a.rv_mod_out;
                const float* q = a.rowvec + (size_t)idx * a.N;
                L[p].rv0 = *(const f32x4*)(q + n_lo);
                L[p].rv1 = *(const f32x4*)(q + n_hi);
            }
        }
    };
    auto settle = [&](PassLoads (&L)[G]) {               // all loads of the group have landed
#pragma unroll
        for (int p = 0; p < G; ++p) {
            if (R1 && R2 && RV) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t1), "+v"(L[p].t2), "+v"(L[p].rv0), "+v"(L[p].rv1) :: "memory");
            else if (R1 && R2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t1), "+v"(L[p].t2) :: "memory");
            else if (R1 && RV) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t1), "+v"(L[p].rv0), "+v"(L[p].rv1) :: "memory");
            else if (R2 && RV) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t2), "+v"(L[p].rv0), "+v"(L[p].rv1) :: "memory");
            else if (R1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t1) :: "memory");
            else if (R2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].t2) :: "memory");
            else if (RV) asm volatile("s_waitcnt vmcnt(0)" : "+v"(L[p].rv0), "+v"(L[p].rv1) :: "memory");
        }
    };

    // The three scale factors live in VGPRs on purpose.  As SGPR operands hipcc folds (s_acc, s1) into one SGPR pair
    // feeding v_pk_mul_f32 / v_pk_fma_f32 with op_sel cross terms, and on MI355X (two workgroups per CU) that code
    // intermittently dropped the s1 * r1 term of output column 4 of a piece in lanes 48..63 (tools/igemm_det.hip: 7 of 7
    // repeat launches differed; with VGPR operands 0 of 7 for every epilogue kind).
    float s1v = a.s1, s2v = a.s2, saccv = a.s_acc;
    asm volatile("" : "+v"(s1v), "+v"(s2v), "+v"(saccv));
    PassLoads LA[G], LB[G];                               // this group's / (AHEAD) the next group's loads
    if (RPPASS > 0) issue_group_loads(0, LA);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        // ---- phase 1: raw accumulators, fragment layout -> slab (same-wave DS operations execute in order) ----
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                *(f32x4*)(slab + slab_off(l31, j * 8 + 2 * q + lh)) = v;
            }
        // ---- phase 2: P passes of RPP rows in NG load groups; a lane owns 8 consecutive output columns of one row ----
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int grp = i * NG + gi;
            PassLoads (&L)[G] = (AHEAD && (grp & 1)) ? LB : LA;
            if (RPPASS > 0) {
                if (!AHEAD && grp > 0) issue_group_loads(grp, LA);
                settle(L);
                if (AHEAD && grp + 1 < MI * NG) issue_group_loads(grp + 1, (grp & 1) ? LA : LB);
            }
#pragma unroll
            for (int pg = 0; pg < G; ++pg) {
                const int p = gi * G + pg;
                const int row = p * RPP + lrow;
                const int m = mrow0 + i * 32 + row;
                float v[8];
                // (GEGLU: value chunks of outputs 8 lc .. 8 lc + 7 = 16-byte chunks 8 t + 2 o, + 1 of the 64-column slab row,
                //  t = lc >> 1, o = lc & 1; the gate chunks follow 4 chunks = 16 columns later)
                const int vch = GEGLU ? 8 * (lc >> 1) + 2 * (lc & 1) : 2 * lc;
                {
                    const f32x4 v0 = *(const f32x4*)(slab + slab_off(row, vch));
                    const f32x4 v1 = *(const f32x4*)(slab + slab_off(row, vch + 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
                }
                if (GEGLU) {
                    const f32x4 g0 = *(const f32x4*)(slab + slab_off(row, vch + 4));
                    const f32x4 g1 = *(const f32x4*)(slab + slab_off(row, vch + 5));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = saccv * (v[e] + bias[0][e]) * gelu_erf_f(saccv * (g0[e] + bias[2][e]));
                        v[4 + e] = saccv * (v[4 + e] + bias[1][e]) * gelu_erf_f(saccv * (g1[e] + bias[3][e]));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0 = v[e] + bias[0][e], x1 = v[4 + e] + bias[1][e];
                        if (RV) { x0 += L[pg].rv0[e]; x1 += L[pg].rv1[e]; }
                        x0 *= saccv; x1 *= saccv;
                        if (R1 || R2) { x0 = (float)(f16)x0; x1 = (float)(f16)x1; }   // fp16 before the residual add, on every tile kernel
                        if (R1) { x0 += s1v * (float)L[pg].t1[e]; x1 += s1v * (float)L[pg].t1[4 + e]; }
                        if (R2) { x0 += s2v * (float)L[pg].t2[e]; x1 += s2v * (float)L[pg].t2[4 + e]; }
                        v[e] = x0; v[4 + e] = x1;
                    }
                    if (a.act == MOFA_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                    } else if (a.act == MOFA_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                    } else if (a.act == MOFA_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
                    }
                }
                if (m < a.M && c4) {
                    f16* po = out + (size_t)m * a.ldo + n;
                    if (WIDE) {
                        f16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
                        *(f16x8*)po = o;
                    } else {
                        const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                        *(f16x4*)po = o;
                        if (c8) {
                            const f16x4 o2 = {(f16)v[4], (f16)v[5], (f16)v[6], (f16)v[7]};
                            *(f16x4*)(po + 4) = o2;
                        }
                    }
                }
            }
        }
    }
}

// =====================================================================================================================
// the kernel.  WM x WN waves, each MI x 2 MFMA tiles: <2,2,2> = 128x128 tile / 256 threads, <2,4,4> = 256x256 / 512.
// =====================================================================================================================
template <int WM, int WN, int MI, int EPI>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_f16_kernel(const mofa_igemm_args a, const int tilesN,
                                                                 const int ntiles) {
    constexpr int NJ = 2;
    constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32, NW = WM * WN;
    constexpr int BKS = 64, RB = 128;                              // K per stage, LDS row bytes
    constexpr int XI = TBM / 8 / NW, WI = TBN / 8 / NW;            // DMA instructions (8 rows each) per wave per stage
    constexpr int SXB = TBM * RB, STB = SXB + TBN * RB;            // stage bytes: X part, total
    static_assert(NW * EPI_SLAB_BYTES <= STB, "epilogue slabs must fit in one ring stage");
    constexpr bool GEGLU = (EPI & EPI_GEGLU) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY shared object: 2 stages

    TileWalk walk;
    walk.init(ntiles);
    if (walk.local >= walk.count) return;                          // whole workgroup idle
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = (a.mode == MOFA_MODE_CONV3X3) ? (a.ksize > 0 ? a.ksize * a.ksize : 9) : (a.mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const int kpt = a.Cin / BKS;         // K stages per tap
    const int nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;
    const bool wide = ((a.ldo & 7) == 0) && ((a.N & 7) == 0) && (!(EPI & EPI_R1) || (a.ldr1 & 7) == 0) &&
                      (!(EPI & EPI_R2) || (a.ldr2 & 7) == 0) && ((((size_t)a.out) & 15) == 0) &&
                      ((((size_t)a.r1) & 15) == 0) && ((((size_t)a.r2) & 15) == 0);
    // VMEM operations an interior tile's epilogue issues (per pass: residual / row-vector loads + 1 store)
    constexpr int EPI_PASSES = MI * (GEGLU ? 2 : 4);
    const int epi_ops = EPI_PASSES * (1 + ((EPI & EPI_R1) ? 1 : 0) + ((EPI & EPI_R2) ? 1 : 0) + ((EPI & EPI_RV) ? 2 : 0));

    // ---- DMA mapping: one global_load_lds instruction fills 8 rows (lane -> row lane / 8, 16-byte slot lane % 8).
    //      The slot a lane fills holds source chunk  c = slot ^ ((row >> 1) & 7)  (swizzle on the SOURCE address; the
    //      LDS image stays lane-linear) -> conflict-free ds_read_b128 fragment reads.
    RowGeo geo[XI];
    int xoff[XI];
    const f16* wsrc[WI];
    const f16* xs[XI];
    int m0 = 0, n0 = 0, itap = 0, ikc = 0, ksw = 0;   // tile origin; position of the NEXT stage to issue
#pragma unroll
    for (int q = 0; q < XI; ++q) {
        const int row = (wave * XI + q) * 8 + (lane >> 3);
        xoff[q] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        xs[q] = nullptr;
    }
    auto setup = [&](int tile) {
        const int tm = tile / tilesN, tn = tile - tm * tilesN;
        m0 = tm * TBM; n0 = tn * TBN;
#pragma unroll
        for (int q = 0; q < XI; ++q) geo[q] = make_geo(a, m0 + (wave * XI + q) * 8 + (lane >> 3));
#pragma unroll
        for (int q = 0; q < WI; ++q) {
            const int row = (wave * WI + q) * 8 + (lane >> 3);
            int n = n0 + row;
            n = n < a.N ? n : a.N - 1;
            wsrc[q] = (const f16*)a.w + (size_t)n * Ktot + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
        }
        itap = 0; ikc = 0; ksw = 0;
    };
    auto issue = [&](int stage) {
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
        for (int q = 0; q < WI; ++q) glds16(wsrc[q] + (size_t)ksw * BKS, sb + SXB + (wave * WI + q) * 1024);
        ++ksw;
        if (++ikc == kpt) { ikc = 0; ++itap; }
    };

    const int fsw = (l31 >> 1) & 7;                       // read-side swizzle term is lane-constant
    const int xrow = (wm * MI * 32 + l31) * RB;           // byte offsets of this lane's fragment rows
    const int wrow = SXB + (wn * NJ * 32 + l31) * RB;

    f32x4 bias[4];
    setup(walk.start + walk.local);
    {
        int lane_b = lane;
        asm volatile("" : "+v"(lane_b));                  // (recomputed per tile, not kept in VGPRs across the K loop)
        load_bias<EPI>(a, n0 + wn * NJ * 32, lane_b, bias);
    }
    issue(0);
    int par = 0;            // ring stage holding K step 0 of the current tile
    int allow = 0;          // VMEM operations that may stay outstanding at the first wait of the current tile
    bool first = true;
    for (;;) {
        const int m0c = m0, n0c = n0;                     // the tile being computed (setup() moves on to the next)
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        auto compute = [&](int cur) {
            const char* sb = smem + cur * STB;
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
        };
        // refill of stage `nxt` spread over the four MFMA groups of stage `cur`: 2 DMA instructions, then 6 ds_reads and
        // 4 (8) MFMAs, four times -- the TA sees a steady trickle instead of a burst and the first MFMA does not wait
        // for the address arithmetic of all eight DMA instructions
        auto compute_and_issue = [&](int cur, int nxt) {
            if (ikc == 0) {
#pragma unroll
                for (int q = 0; q < XI; ++q) xs[q] = x_src(a, geo[q], itap);
            }
            const char* sb = smem + cur * STB;
            char* nb = smem + nxt * STB;
            constexpr int OPS = XI + WI, NG = BKS / 16;      // DMA instructions per stage, MFMA groups per stage
            constexpr int SPREAD = 4;                        // the refill is issued over all four MFMA groups
#pragma unroll
            for (int kk = 0; kk < NG; ++kk) {
#pragma unroll
                for (int o = (kk * OPS / SPREAD < OPS ? kk * OPS / SPREAD : OPS);
                     o < ((kk + 1) * OPS / SPREAD < OPS ? (kk + 1) * OPS / SPREAD : OPS); ++o) {
                    if (o < XI) {
                        const f16* sp = xs[o] ? xs[o] + ikc * BKS + xoff[o] : (const f16*)g_zero_page;
                        glds16(sp, nb + (wave * XI + o) * 1024);
                    } else {
                        glds16(wsrc[o - XI] + (size_t)ksw * BKS, nb + SXB + (wave * WI + (o - XI)) * 1024);
                    }
                }
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
            ++ksw;
            if (++ikc == kpt) { ikc = 0; ++itap; }
        };
        // K steps 0 .. nk-2: wait for the only DMA in flight (at step 0 the previous epilogue's stores may stay
        // outstanding), one barrier (all waves' parts landed; the other stage is free), refill, multiply
        for (int ks = 0; ks < nk - 1; ++ks) {
            if (ks == 0) wait_vmcnt_le(allow); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const int cur = (par + ks) & 1;
            compute_and_issue(cur, cur ^ 1);
        }
        // last K step (peeled: the set-up of the next tile stays out of the steady-state loop)
        const int last = (par + nk - 1) & 1;
        if (nk == 1) wait_vmcnt_le(allow); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        bias_touch<EPI>(bias);                            // landed long ago (loaded a tile ahead); see load_bias
        walk.local += walk.stride;
        const bool has_next = walk.local < walk.count;
        if (has_next) {                                   // stage 0 of the next tile flies during the epilogue
            setup(walk.start + walk.local);
            issue(last ^ 1);
        } else {                                          // same VMEM operation count on both paths (static waitcnts)
#pragma unroll
            for (int q = 0; q < XI + WI; ++q)
                glds16((const f16*)g_zero_page, smem + (last ^ 1) * STB + (wave * (XI + WI) + q) * 1024);
        }
        asm volatile("" ::: "memory");                    // epilogue loads stay behind the DMA (the count assumes it)
        compute(last);
        wait_lds();
        __builtin_amdgcn_s_barrier();                     // every wave is done reading the last stage: it becomes slabs
        char* slab = smem + last * STB + wave * EPI_SLAB_BYTES;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if (wide) igemm_epilogue<MI, EPI, true>(a, acc, m0c + wm * MI * 32, n0c + wn * NJ * 32, lane_e, slab, bias);
        else igemm_epilogue<MI, EPI, false>(a, acc, m0c + wm * MI * 32, n0c + wn * NJ * 32, lane_e, slab, bias);
        first = false;
        if (!has_next) break;
        // counted wait only when every epilogue memory instruction certainly executed with at least one lane
        {                                                 // bias of the next tile (n0 is already the next tile's)
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            load_bias<EPI>(a, n0 + wn * NJ * 32, lane_b, bias);
        }
        allow = (wide && m0c + TBM <= a.M && n0c + TBN <= a.N) ? epi_ops + (a.bias ? (GEGLU ? 4 : 2) : 0) : 0;
        par = last ^ 1;
    }
    // the wave must not retire with an LDS DMA (the last tile's dummy prefetch) in flight: the LDS allocation could be
    // handed to another workgroup while the DMA still writes into it
    wait_vmcnt<0>();
}

extern "C" int mofa_igemm_f16(const mofa_igemm_args* a, mofa_stream_t stream) {
    if (!a || !a->x || !a->w || !a->out) return MOFA_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->Cin <= 0) return MOFA_EINVAL;
    if (a->Cin % 64 != 0 || a->N % 4 != 0) return MOFA_EINVAL;
    if (a->mode < 0 || a->mode > 2 || a->act < 0 || a->act > MOFA_ACT_GELU) return MOFA_EINVAL;
    if (a->mode == MOFA_MODE_CONV3X3) {
        if (a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0) return MOFA_EINVAL;
        if ((a->stride != 1 && a->stride != 2) || (a->up != 1 && a->up != 2)) return MOFA_EINVAL;
        if (a->ksize != 0 && a->ksize != 1 && a->ksize != 3 && a->ksize != 5 && a->ksize != 7) return MOFA_EINVAL;
        if (a->dil < 0 || (a->dil > 1 && a->up != 1)) return MOFA_EINVAL;
        if (a->pad != MOFA_PAD_SAME && a->pad != MOFA_PAD_TRAILING) return MOFA_EINVAL;
        if (a->M % (a->Hout * a->Wout) != 0 || a->Hout > 65535 || a->Wout > 65535) return MOFA_EINVAL;
    }
    if (a->mode == MOFA_MODE_CONVT3 && (a->T < 0 || a->HW <= 0 || (a->T > 0 && a->M % (a->T * a->HW) != 0)))
        return MOFA_EINVAL;
    if (a->rowvec && (a->rv_div <= 0 || a->rv_mod_in <= 0 || a->rv_mod_out <= 0)) return MOFA_EINVAL;
    if (a->act == MOFA_ACT_GEGLU_PAIR && (a->N % 64 != 0 || a->r1 || a->r2 || a->rowvec)) return MOFA_EINVAL;
    if (a->ldx % 8 != 0 || a->ldo % 4 != 0) return MOFA_EINVAL;
    if ((a->r1 && a->ldr1 % 4 != 0) || (a->r2 && a->ldr2 % 4 != 0)) return MOFA_EINVAL;

    // two 4-wave configurations x nine epilogue kinds (bit 0 r1, bit 1 r2, bit 2 row vector; 8 = GEGLU pair); the third
    // choice, the 8-wave phase-pipelined 256x256 tile, lives in igemm8.hip
    struct Cfg { igemm_kern_t k[9]; int tm, tn, threads, lds, wg_per_cu; };
#define IGEMM_KINDS(WM, WN, MI)                                                                                        \
    {igemm_f16_kernel<WM, WN, MI, 0>, igemm_f16_kernel<WM, WN, MI, 1>, igemm_f16_kernel<WM, WN, MI, 2>,                \
     igemm_f16_kernel<WM, WN, MI, 3>, igemm_f16_kernel<WM, WN, MI, 4>, igemm_f16_kernel<WM, WN, MI, 5>,                \
     igemm_f16_kernel<WM, WN, MI, 6>, igemm_f16_kernel<WM, WN, MI, 7>, igemm_f16_kernel<WM, WN, MI, 8>}
    static const Cfg cfgs[2] = {
        {IGEMM_KINDS(2, 2, 2), 128, 128, 256, 2 * 256 * 128, 2},   // 128^2 tile, 2 workgroups per CU
        {IGEMM_KINDS(2, 2, 3), 192, 128, 256, 2 * 320 * 128, 2},   // 192x128 tile: 2 x 80 KB = the whole LDS of a CU
    };
    static bool ready = false;   // one-time set-up: LDS opt-in of every instantiation, CU count
    static int n_cu = 256;
    if (!ready) {
        for (const Cfg& c : cfgs)
            for (igemm_kern_t k : c.k)
                if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds) != hipSuccess)
                    return MOFA_ELAUNCH;
        if (igemm8_init() != MOFA_OK || igemm320_init() != MOFA_OK) return MOFA_ELAUNCH;
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            n_cu = cus >= 8 ? (cus / 8) * 8 : 8;              // persistent grids are multiples of 8 (one per XCD): every tile /
        ready = true;                                         // round / split-K count below uses the SAME rounded figure
    }
    if (a->tile != 0 && a->tile != MOFA_TILE_128X128 && a->tile != MOFA_TILE_192X128 && a->tile != MOFA_TILE_256X256 &&
        a->tile != MOFA_TILE_256X320)
        return MOFA_EINVAL;
    const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const long long Ktot = (long long)taps * a->Cin;
    const int kind = a->act == MOFA_ACT_GEGLU_PAIR ? 8 : ((a->r1 ? 1 : 0) | (a->r2 ? 2 : 0) | (a->rowvec ? 4 : 0));
    int choice = a->tile;                                     // MOFA_TILE_* or 0 = cost model
    if (a->stats) {                                           // GroupNorm pair sums from the epilogue: the 256x320 tile only
        if (!igemm320_stats_ok(a)) return MOFA_EINVAL;
        choice = MOFA_TILE_256X320;
    }
    if (choice == 0) {
        // Tile choice = the cheapest of  rounds of resident workgroups x CU time of one round, the latter modelled as
        //   workgroups per CU x tile area x relative K-loop cost per flop x (1 + epilogue / K loop),  epilogue in K tiles:
        //     128x128   1.00   two workgroups per CU
        //     192x128   0.86   two workgroups fill the CU's 160 KB of LDS exactly; 17 % less fill / LDS-read traffic per
        //                      flop, 50 % more MFMA work per barrier
        //     256x256   0.62   8 waves, phase-pipelined K loop (igemm8.hip; needs 16-byte aligned rows)
        // fitted to tools/igemm_tiles_bench.py on the denoise step's shapes (profiles/r02_igemm_tiles.md).  Rounds count the
        // padding waste of partial tiles and the idle slots of the last round (e.g. N = 320: 2 x 256 wastes 37 %, 3 x 128
        // 17 %; M = 7200: 570 tiles of 128x128 need two rounds of 512 slots, 380 tiles of 192x128 one).
        static const double rel[2] = {1.00, 0.86};
        static const int ids[2] = {MOFA_TILE_128X128, MOFA_TILE_192X128};
        // epilogue cost in K tiles per epilogue kind (index = kind): 4-wave tiles (their epilogue overlaps the co-resident
        // workgroup's K loop) / the 8-wave tile
        static const double epi4[9] = {2.0, 3.5, 4.0, 4.0, 3.5, 4.0, 4.5, 4.5, 4.0};
        static const double epi8[9] = {2.2, 5.2, 5.2, 6.0, 5.6, 5.5, 5.5, 6.3, 2.4};   // ([4]: fitted to the N = 320 choice, not the 2.6 measured)
        const double nk = (double)(Ktot / 64);
        double best = 0;
        for (int k = 0; k < 2; ++k) {
            const long long t = (long long)cdiv(a->M, cfgs[k].tm) * cdiv(a->N, cfgs[k].tn);
            const long long slots_k = (long long)n_cu * cfgs[k].wg_per_cu;
            const double cost = (double)((t + slots_k - 1) / slots_k) * cfgs[k].tm * cfgs[k].tn * rel[k] * cfgs[k].wg_per_cu *
                                (1.0 + epi4[kind] / nk);
            if (choice == 0 || cost < best) { best = cost; choice = ids[k]; }
        }
        {
            const long long t = (long long)cdiv(a->M, 256) * cdiv(a->N, 256);
            const double cost = (double)((t + n_cu - 1) / n_cu) * 256 * 256 * 0.62 * (1.0 + epi8[kind] / nk);
            if (cost < best) { best = cost; choice = MOFA_TILE_256X256; }
        }
        if (kind != 7) {
            // 256x320 (igemm320.hip): 0.58 per area (10 % less LDS-DMA, 7 % fewer fragment reads per flop than 256x256); its
            // epilogues move 25 % more outputs per tile
            static const double epi320[9] = {3.0, 7.5, 8.5, 9.5, 3.5, 7.9, 9.0, 12.0, 5.0};   // fitted: profiles/archive/r03_igemm_tiles_bench.log; r04: the
            // residual kinds [1] [2] [3] [5] [6] lowered by 1.5-2 with the fp16-transposed residual epilogue (profiles/r04_res16_tiles_ab.log);
            // [1], [5] lowered from 10.0 / 10.5 in r03c so that the K = N = 320 residual launches leave the 192x128 tile (alone a
            // draw: 352-414 against 372-403 TF/s by box; in the clip - 0.27 %, profiles/archive/r03c_cost_model_and_splitk_ab.log:
            // beside a second stream the 17 % of padded columns of 3 x 128 are no longer free)
            const long long t = (long long)cdiv(a->M, 256) * cdiv(a->N, 320);
            // a partial last round split along K (igemm320_split) costs 1 / S of a round + the fix-up pass
            const int S = kind == 8 ? 1 : igemm320_split(t, (int)nk, n_cu, a->workspace ? a->workspace_bytes : 0);
            const double rounds = S > 1 ? (double)(t / n_cu) + 1.0 / S + 0.12 : (double)((t + n_cu - 1) / n_cu);
            // wide outputs (many column tiles: every CU of an XCD streams its own weight tile through the fabric) run
            // relatively slower on this tile than on 256x256 (profiles/archive/r03_igemm_tiles_bench_geglu320.log: N = 3840 / 10240
            // 7 / 15 % behind): +2 % per column tile beyond 6, at most +25 %
            const int tn320 = cdiv(a->N, 320);
            const double wide = 1.0 + (tn320 > 6 ? (tn320 - 6 > 12 ? 0.25 : 0.02 * (tn320 - 6)) : 0.0);
            const double cost = rounds * 256 * 320 * 0.58 * wide * (1.0 + epi320[kind] / nk);
            if (cost < best) { best = cost; choice = MOFA_TILE_256X320; }
        }
    }
    if (choice == MOFA_TILE_256X320) {
        const int rc = igemm320_launch(a, kind, n_cu, (hipStream_t)stream);
        if (rc <= 0) return rc;                               // launched (0) or failed (< 0)
        if (a->tile == MOFA_TILE_256X320 || a->stats) return MOFA_EINVAL; // explicitly requested but not eligible
        choice = MOFA_TILE_256X256;                           // chosen by the model but not eligible: next best pipeline tile
    }
    if (choice == MOFA_TILE_256X256) {
        const int rc = igemm8_launch(a, kind, n_cu, (hipStream_t)stream);
        if (rc <= 0) return rc;                               // launched (0) or failed (< 0)
        if (a->tile == MOFA_TILE_256X256) return MOFA_EINVAL; // explicitly requested but not eligible (alignment)
        choice = MOFA_TILE_192X128;                           // chosen by the model but not eligible: fall back
    }
    const Cfg* sel = &cfgs[choice == MOFA_TILE_128X128 ? 0 : 1];
    const Cfg& c = *sel;
    const int tilesM = cdiv(a->M, c.tm), tilesN = cdiv(a->N, c.tn);
    const long long nt = (long long)tilesM * tilesN;
    if (nt > 0x7fffffffLL) return MOFA_EINVAL;
    const int slots = n_cu * c.wg_per_cu;                     // resident workgroups (a multiple of 8 on gfx950)
    int grid = (int)(nt < slots ? ((nt + 7) / 8) * 8 : (slots / 8) * 8);
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(c.k[kind], dim3(grid), dim3(c.threads), c.lds, (hipStream_t)stream, *a, tilesN, (int)nt);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

extern "C" int mofa_igemm_stats_ok(const mofa_igemm_args* a) { return igemm320_stats_ok(a) ? 1 : 0; }
