// Implicit GEMM, 256x256 output tile, 8 waves, phase-pipelined K loop (the workhorse tile of the denoise loop).
//
// Same contract as igemm.hip (see mofa_hip.h: out = act(s_acc * (conv + bias + rowvec) + s1 * r1 + s2 * r2)) and the
// same operand roles (MFMA A = weight tile, B = activation tile: an accumulator lane owns one output row), LDS image
// (128-byte rows, XOR swizzle on the DMA source address and on the fragment read) and persistent XCD-aware tile walk.
// What differs is the K loop, built after the 8-phase schedule of the CDNA4 guide (cdna_hip_programming.md "The 256^2
// 8-phase template", T3+T4+T5):
//
//   * wave (wm, wn) of 2 x 4 owns 128 (M) x 64 (N) outputs = 4 x 2 accumulator tiles of 32x32.  A K tile (64 deep) is
//     consumed in FOUR PHASES, one quadrant (2 x 1 accumulator tiles, 8 MFMAs = 256 matrix-pipe cycles) each:
//         phase 1: X half 0 x W half 0      reads X0 (8 ds_read_b128) + W0 (4)
//         phase 2: X half 0 x W half 1      reads W1 (4)                       (X0 stays in registers)
//         phase 3: X half 1 x W half 1      reads X1 (8)                       (W1 stays)
//         phase 4: X half 1 x W half 0      reads nothing                      (W0 was kept since phase 1)
//     A phase is  {fragment reads, 2 LDS-DMA instructions, s_waitcnt vmcnt(8)} s_barrier {8 MFMAs} s_barrier.
//   * the two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so on every
//     SIMD one wave is in its MFMA segment while its partner reads fragments and issues DMA; s_setprio(1) around the
//     MFMA cluster lets the matrix pipe win the issue arbitration.
//   * the LDS ring is 2 K tiles x 4 half tiles (X0, X1, W0, W1: 128 rows x 128 B = 16 KB each).  A half tile is
//     released as soon as its fragments are in registers (X0, W0 after phase 1, W1 after 2, X1 after 3) and refilled
//     two phases later; every phase issues exactly one half tile (each wave 2 DMA instructions of 8 rows):
//         phase 1: W1 of K tile t+1     phase 2: X1 of t+1     phase 3: X0 of t+2     phase 4: W0 of t+2
//     so every half tile has FOUR phases of flight before the single counted wait `vmcnt(8)` (= the DMA of the last
//     four phases may stay outstanding) retires it, and it is read no earlier than the phase after that wait (the
//     wait precedes the phase's first barrier; with the two wave groups one barrier apart that is the barrier after
//     which the other group's reads start -- guide: "read a staged buffer one phase AFTER the wait that retires it").
//     vmcnt is never drained inside the loop, and the stream of half tiles runs on ACROSS output tiles: while a tile's
//     epilogue runs, the first one and a half K tiles of the workgroup's next output tile are already in flight.
//   * two producer cursors (one per half-tile pair) walk the same persistent tile sequence as the consumer, 1 and 2 K
//     tiles ahead; past the end of the walk they source the zero page so that the instruction count stays static.
//
// Epilogue: private 4 KB LDS scratch per wave OUTSIDE the ring (160 KB = 2 x 64 KB ring + 8 x 4 KB), so nothing in the
// epilogue aliases a DMA target and no barrier is needed inside it.  Kinds without residuals (plain, row vector, GEGLU)
// apply bias / row vector / activation in the accumulator (fragment) layout, convert to fp16 and transpose 32 x 64 fp16
// through the scratch (ds_write_b64 / ds_read_b128, both conflict-free: tools/lds_bank_sim.py) -- half the LDS write
// traffic of an fp32 transpose, which is what bounds the epilogue (ds_write ~ 80 B/clk/CU).  Kinds with residuals keep
// fp32 through the transpose (32 x 32 fp32 per pass) so that the sum is rounded once.
// Eligibility (checked by the launcher): 16-byte aligned rows everywhere ("wide" path of igemm.hip), N % 8 == 0.
#include "igemm_common.h"

namespace {

constexpr int WM = 2, WN = 4, MI = 4, NJ = 2;                  // waves, accumulator tiles per wave
constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32;          // 256 x 256
constexpr int BKS = 64, RB = 128;                              // K per tile, LDS row bytes
constexpr int SXB = TBM * RB, STB = SXB + TBN * RB;            // bytes: X part, one K tile of the ring
constexpr int SCRATCH0 = 2 * STB, SCRATCH_WAVE = 4096;
constexpr int LDS_BYTES = SCRATCH0 + 8 * SCRATCH_WAVE;         // 163840 = the CU's whole LDS
constexpr int LOOKAHEAD_OPS = 8;                               // DMA instructions of the last 4 phases may be in flight

// epilogue scratch images (tools/lds_bank_sim.py)
__device__ __forceinline__ int h16_off(int r, int c8) {        // 32 rows x 64 fp16; c8 = 8-byte chunk (4 columns)
    return r * 128 + (((c8 >> 1) ^ ((r >> 1) & 7)) << 4) + (((c8 & 1) ^ (r & 1)) << 3);
}
__device__ __forceinline__ int f32_off(int r, int c) {         // 32 rows x 32 fp32; c = 16-byte chunk (4 columns)
    return r * 128 + ((c ^ (((r >> 1) & 3) | ((r & 1) << 2))) << 4);
}

// division by a launch-invariant divisor without v_rcp sequences (whose loop-invariant parts hipcc hoists out of the
// persistent loop and then spills): q = umulhi(n, mul) >> shr, exact for 0 <= n < 2^31 (host side: fastdiv_make)
struct FastDiv { unsigned mul, shr; };
__device__ __forceinline__ int fdiv(int n, const FastDiv d) { return d.mul ? (int)(__umulhi((unsigned)n, d.mul) >> d.shr) : n; }

struct Aux {                        // launch-invariant scalars computed by the launcher
    FastDiv tiles_n, hw, wout, t3hw, t3t;
    int ldx16;                      // activation row stride in 16-byte units
    const f16* zero;                // the zero page (as an argument: addressing the symbol costs an s_load through the
};                                  // GOT at every use, and its lgkmcnt(0) wait also drains the fragment reads in flight)

// One DMA row group = ONE packed register (Cursor::gx), decoded at every tap switch:
//   plain    m                                         conv     img << 20 | oy << 10 | ox
//   convT3   m | (frame > 0 or unclipped) << 29 | (frame < T - 1 or unclipped) << 30
//   -1       row beyond M
constexpr int XO_INVALID = (int)0x80000000;

struct Cursor {                     // one half-tile pair (X half h, W half h) of the persistent K-tile stream
    int local;                      // walk position of the output tile it is in
    int ikc, ksw, ky, kx;           // K tile within the tap / overall, tap coordinates (convT3: ky = tap)
    bool live;                      // false past the end of the walk: every source is the zero page
    int gx[2];                      // packed row geometry
    int xo[2];                      // source row of the current tap, 16-byte units from a.x (XO_INVALID = zero page)
    unsigned wo[2];                 // weight row + this lane's swizzled chunk, bytes from a.w
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void igemm8_f16_kernel(const mofa_igemm_args a, const int tilesN, const int ntiles,
                                                            const Aux aux) {
    constexpr bool GEGLU = (EPI & EPI_GEGLU) != 0, R1 = (EPI & EPI_R1) != 0, R2 = (EPI & EPI_R2) != 0,
                   RV = (EPI & EPI_RV) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY shared object
    TileWalk walk;
    walk.init(ntiles);
    if (walk.local >= walk.count) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                     // waves w and w + 4 share a SIMD
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = igemm_taps(a);
    const int kpt = a.Cin / BKS, nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;

    // ---- producer side ----------------------------------------------------------------------------------------------
    // piece q (8 rows x 128 B, one DMA instruction) of half h: half-local row hr = 16 wave + 8 q + lane / 8;
    // X tile row = (hr / 64) * 128 + h * 64 + hr % 64, W tile row = (hr / 32) * 64 + h * 32 + hr % 32; the 16-byte slot
    // lane % 8 of a row holds source chunk slot ^ ((row >> 1) & 7) = slot ^ (4 q + lane / 16)
    const int hr_l = 16 * wave + (lane >> 3);
    int swz_e[2];                                                  // element offset of this lane's chunk, per piece
    swz_e[0] = ((lane & 7) ^ (lane >> 4)) * 8;
    swz_e[1] = ((lane & 7) ^ (4 + (lane >> 4))) * 8;
    auto x_row = [&](int h, int q, int hr0) { const int hr = hr0 + 8 * q; return (hr >> 6) * 128 + h * 64 + (hr & 63); };
    auto w_row = [&](int h, int q, int hr0) { const int hr = hr0 + 8 * q; return (hr >> 5) * 64 + h * 32 + (hr & 31); };
    const int ks_ = a.ksize > 0 ? a.ksize : 3, dil_ = a.dil > 0 ? a.dil : 1;
    const int org_ = a.pad == MOFA_PAD_TRAILING ? 0 : (ks_ >> 1);
    auto pack_geo = [&](int m) -> int {
        if (m >= a.M) return -1;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int img = fdiv(m, aux.hw), rem = m - img * (a.Hout * a.Wout);
            const int oy = fdiv(rem, aux.wout);
            return (img << 20) | (oy << 10) | (rem - oy * a.Wout);
        }
        if (a.mode == MOFA_MODE_CONVT3) {
            int lo = 1, hi = 1;
            if (a.T > 0) {
                const int fr = fdiv(m, aux.t3hw);                  // frame index; its position within the clip of T
                const int f = fr - fdiv(fr, aux.t3t) * a.T;
                lo = f > 0; hi = f < a.T - 1;
            }
            return m | (lo << 29) | (hi << 30);
        }
        return m;
    };
    auto tap_src = [&](int g, int ky, int kx) -> int {            // 16-byte units from a.x, or XO_INVALID
        if (g < 0) return XO_INVALID;
        if (a.mode == MOFA_MODE_PLAIN) return g * aux.ldx16;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int vy = ((g >> 10) & 1023) * a.stride + (ky - org_) * dil_;
            const int vx = (g & 1023) * a.stride + (kx - org_) * dil_;
            if (vy < 0 || vx < 0 || vy >= a.Hin * a.up || vx >= a.Win * a.up) return XO_INVALID;
            const int iy = (a.up == 2) ? (vy >> 1) : vy, ix = (a.up == 2) ? (vx >> 1) : vx;
            return (((g >> 20) * a.Hin + iy) * a.Win + ix) * aux.ldx16;
        }
        const int m = g & 0x1fffffff;                              // convT3: tap ky - 1 frames away
        if ((ky == 0 && !((g >> 29) & 1)) || (ky == 2 && !((g >> 30) & 1))) return XO_INVALID;
        return (m + (ky - 1) * a.HW) * aux.ldx16;
    };
    auto cur_setup = [&](Cursor& c, const int h) {
        c.live = c.local < walk.count;
        if (c.live) {
            const int tile = walk.start + c.local;
            const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                c.gx[q] = pack_geo(tm * TBM + x_row(h, q, hr_l));
                int n = tn * TBN + w_row(h, q, hr_l);
                n = n < a.N ? n : a.N - 1;
                c.wo[q] = (unsigned)n * (unsigned)(Ktot * 2) + swz_e[q] * 2;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) { c.gx[q] = -1; c.wo[q] = swz_e[q] * 2; }
        }
        c.ikc = 0; c.ksw = 0; c.ky = 0; c.kx = 0;
    };
    auto issue_x = [&](Cursor& c, const int h, const int bo) {
        if (c.ikc == 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) c.xo[q] = tap_src(c.gx[q], c.ky, c.kx);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const char* s = (const char*)a.x + ((long long)(c.xo[q] + c.ikc * 8 + (swz_e[q] >> 3)) << 4);
            if (c.xo[q] == XO_INVALID) s = (const char*)aux.zero;
            glds16((const f16*)s, smem + bo + x_row(h, q, 16 * wave) * RB);
        }
    };
    auto issue_w = [&](Cursor& c, const int h, const int bo) {
        const char* base = c.live ? (const char*)a.w + (size_t)c.ksw * (BKS * 2) : (const char*)aux.zero;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16((const f16*)(base + c.wo[q]), smem + bo + SXB + w_row(h, q, 16 * wave) * RB);
    };
    auto advance = [&](Cursor& c, const int h) {
        ++c.ksw;
        if (++c.ikc == kpt) {
            c.ikc = 0;
            if (a.mode == MOFA_MODE_CONV3X3) { if (++c.kx == ks_) { c.kx = 0; ++c.ky; } } else ++c.ky;
        }
        if (c.ksw == nk) { c.local += walk.stride; cur_setup(c, h); }
    };

    // ---- consumer side ----------------------------------------------------------------------------------------------
    const int fsw = (l31 >> 1) & 7;
    const int xfrag = (wm * MI * 32 + l31) * RB;                   // byte offsets of this lane's fragment rows
    const int wfrag = SXB + (wn * NJ * 32 + l31) * RB;
    int slot[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) slot[kk] = ((kk * 2 + lh) ^ fsw) * 16;

    Cursor ca, cb;                                                 // (X0, W0) stream, (X1, W1) stream
    ca.local = cb.local = walk.local;
    cur_setup(ca, 0);
    cur_setup(cb, 1);
    // prologue: K tile 0 complete, (X0, W0) of K tile 1
    issue_x(ca, 0, 0); issue_w(ca, 0, 0); advance(ca, 0);
    issue_w(cb, 1, 0); issue_x(cb, 1, 0); advance(cb, 1);
    issue_x(ca, 0, STB); issue_w(ca, 0, STB); advance(ca, 0);
    wait_vmcnt_only<LOOKAHEAD_OPS>();                              // X0, W0 of K tile 0 have landed
    __builtin_amdgcn_s_barrier();

    int gt = 0;                                                    // K tiles consumed so far (ring slot = gt & 1)
    for (int cl = walk.local; cl < walk.count; cl += walk.stride) {
        const int tile = walk.start + cl;
        const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        if (grp == 1) __builtin_amdgcn_s_barrier();                // second wave group runs one barrier behind
        for (int kt = 0; kt < nk; ++kt, ++gt) {
            const int bo = (gt & 1) * STB, bn = STB - bo;          // ring slot of this K tile / of the next one
            const char* sb = smem + bo;
            f16x8 xf[2][4], w0[4], w1[4];
            // ---------------- phase 1: X0 x W0 ----------------
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) xf[il][kk] = *(const f16x8*)(sb + xfrag + il * 32 * RB + slot[kk]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w0[kk] = *(const f16x8*)(sb + wfrag + slot[kk]);
            issue_w(cb, 1, bn);
            wait_vmcnt_only<LOOKAHEAD_OPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int il = 0; il < 2; ++il)
                    acc[il][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kk], xf[il][kk], acc[il][0], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- phase 2: X0 x W1 ----------------
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w1[kk] = *(const f16x8*)(sb + wfrag + 32 * RB + slot[kk]);
            issue_x(cb, 1, bn);
            advance(cb, 1);
            wait_vmcnt_only<LOOKAHEAD_OPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int il = 0; il < 2; ++il)
                    acc[il][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kk], xf[il][kk], acc[il][1], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- phase 3: X1 x W1 ----------------
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) xf[il][kk] = *(const f16x8*)(sb + xfrag + (2 + il) * 32 * RB + slot[kk]);
            issue_x(ca, 0, bo);
            wait_vmcnt_only<LOOKAHEAD_OPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int il = 0; il < 2; ++il)
                    acc[2 + il][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[kk], xf[il][kk], acc[2 + il][1], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- phase 4: X1 x W0 ----------------
            issue_w(ca, 0, bo);
            advance(ca, 0);
            wait_vmcnt_only<LOOKAHEAD_OPS>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int il = 0; il < 2; ++il)
                    acc[2 + il][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[kk], xf[il][kk], acc[2 + il][0], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();                // both groups meet again: equal barrier counts per tile

        // ---- epilogue (no barriers; private scratch) ------------------------------------------------------------------
        char* scr = smem + SCRATCH0 + wave * SCRATCH_WAVE;
        const int mw = tm * TBM + wm * MI * 32;                    // first output row / (pre-GEGLU) column of this wave
        const int nw = tn * TBN + wn * NJ * 32;
        f16* out = (f16*)a.out;
        float s1v = a.s1, s2v = a.s2, saccv = a.s_acc;             // VGPR operands on purpose (see igemm.hip's epilogue)
        asm volatile("" : "+v"(s1v), "+v"(s2v), "+v"(saccv));
        int rv_div = a.rv_div, rv_mod_in = a.rv_mod_in, rv_mod_out = a.rv_mod_out;
        asm volatile("" : "+s"(rv_div), "+s"(rv_mod_in), "+s"(rv_mod_out));   // reciprocals are set up here, not hoisted
        if constexpr (!R1 && !R2) {
            // ---- fragment-layout math, fp16 transpose.  register r of accumulator tile (i, j): row 32 i + l31, column
            //      32 j + 8 (r >> 2) + 4 lh + (r & 3)
            f32x4 bv[NJ][4];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int n = nw + 32 * j + 8 * g + 4 * lh;
                    n = n + 4 <= a.N ? n : 0;                       // columns beyond N are never stored
                    bv[j][g] = a.bias ? *(const f32x4*)(a.bias + n) : (f32x4){0, 0, 0, 0};
                }
            const int nout = GEGLU ? a.N / 2 : a.N;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float* rvp = nullptr;
                if (RV) {
                    int m = mw + 32 * i + l31;
                    m = m < a.M ? m : a.M - 1;
                    const int idx = ((m / rv_div) * a.rv_mul + (m % rv_mod_in)) % rv_mod_out;
                    rvp = a.rowvec + (size_t)idx * a.N;
                }
                if constexpr (GEGLU) {
                    // value tile j = 0, gate tile j = 1 (weight rows interleaved in blocks of 32 at load time); the
                    // outputs of accumulator rows i (even) and i + 1 share one scratch image: columns 0..31 / 32..63
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[e] = (f16)(saccv * (acc[i][0][4 * g + e] + bv[0][g][e]) *
                                         gelu_erf_f(saccv * (acc[i][1][4 * g + e] + bv[1][g][e])));
                        *(f16x4*)(scr + h16_off(l31, 8 * (i & 1) + 2 * g + lh)) = o;
                    }
                    if (i & 1) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const int row = 8 * p + (lane >> 3), blk = lane & 7;
                            const f16x8 v = *(const f16x8*)(scr + row * 128 + ((blk ^ ((row >> 1) & 7)) << 4));
                            const int m = mw + 32 * (i - 1 + (blk >> 2)) + row;
                            const int n = nw / 2 + 8 * (blk & 3);
                            f16x8 o = v;
                            if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                            if (m < a.M && n + 8 <= nout) *(f16x8*)(out + (size_t)m * a.ldo + n) = o;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 rv = {0, 0, 0, 0};
                            if (RV) {
                                int n = nw + 32 * j + 8 * g + 4 * lh;
                                n = n + 4 <= a.N ? n : 0;
                                rv = *(const f32x4*)(rvp + n);
                            }
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = saccv * (acc[i][j][4 * g + e] + bv[j][g][e] + rv[e]);
                            if (a.act == MOFA_ACT_SILU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                            } else if (a.act == MOFA_ACT_RELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                            } else if (a.act == MOFA_ACT_GELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                            }
                            const f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                            *(f16x4*)(scr + h16_off(l31, 8 * j + 2 * g + lh)) = o;
                        }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int row = 8 * p + (lane >> 3), blk = lane & 7;
                        const f16x8 v = *(const f16x8*)(scr + row * 128 + ((blk ^ ((row >> 1) & 7)) << 4));
                        const int m = mw + 32 * i + row, n = nw + 8 * blk;
                        f16x8 o = v;
                        if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                        if (m < a.M && n + 8 <= nout) *(f16x8*)(out + (size_t)m * a.ldo + n) = o;
                    }
                }
            }
        } else {
            // ---- residual kinds: fp32 transpose, 32 x 32 per pass; a lane owns 8 consecutive columns of one row ----
            const int piece = lane & 3;
            const f16* r1 = (const f16*)a.r1;
            const f16* r2 = (const f16*)a.r2;
            f32x4 bv[NJ][2];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int n = nw + 32 * j + 8 * piece;
                n = n + 8 <= a.N ? n : 0;
                bv[j][0] = a.bias ? *(const f32x4*)(a.bias + n) : (f32x4){0, 0, 0, 0};
                bv[j][1] = a.bias ? *(const f32x4*)(a.bias + n + 4) : (f32x4){0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int n = nw + 32 * j + 8 * piece;
                    const int nc = n + 8 <= a.N ? n : 0;
                    // residual / row-vector loads of both passes go out before the transpose
                    f16x8 t1[2], t2[2];
                    f32x4 rv0[2], rv1[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        int m = mw + 32 * i + 16 * p + (lane >> 2);
                        m = m < a.M ? m : a.M - 1;
                        if (R1) t1[p] = *(const f16x8*)(r1 + (size_t)m * a.ldr1 + nc);
                        if (R2) t2[p] = *(const f16x8*)(r2 + (size_t)m * a.ldr2 + nc);
                        if (RV) {
                            const int idx = ((m / rv_div) * a.rv_mul + (m % rv_mod_in)) % rv_mod_out;
                            const float* q = a.rowvec + (size_t)idx * a.N + nc;
                            rv0[p] = *(const f32x4*)q;
                            rv1[p] = *(const f32x4*)(q + 4);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                        *(f32x4*)(scr + f32_off(l31, 2 * g + lh)) = v;
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int row = 16 * p + (lane >> 2);
                        const f32x4 v0 = *(const f32x4*)(scr + f32_off(row, 2 * piece));
                        const f32x4 v1 = *(const f32x4*)(scr + f32_off(row, 2 * piece + 1));
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x0 = v0[e] + bv[j][0][e], x1 = v1[e] + bv[j][1][e];
                            if (RV) { x0 += rv0[p][e]; x1 += rv1[p][e]; }
                            x0 *= saccv; x1 *= saccv;
                            if (R1) { x0 += s1v * (float)t1[p][e]; x1 += s1v * (float)t1[p][4 + e]; }
                            if (R2) { x0 += s2v * (float)t2[p][e]; x1 += s2v * (float)t2[p][4 + e]; }
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
                        const int m = mw + 32 * i + row;
                        if (m < a.M && n + 8 <= a.N) {
                            f16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
                            *(f16x8*)(out + (size_t)m * a.ldo + n) = o;
                        }
                    }
                }
        }
    }
    // a wave must not retire with an LDS DMA in flight (the cursors' run-ahead past the last tile)
    wait_vmcnt_only<0>();
}

}  // namespace

typedef void (*igemm8_kern_t)(const mofa_igemm_args, const int, const int, const Aux);
static const igemm8_kern_t k_igemm8[9] = {igemm8_f16_kernel<0>, igemm8_f16_kernel<1>, igemm8_f16_kernel<2>,
                                           igemm8_f16_kernel<3>, igemm8_f16_kernel<4>, igemm8_f16_kernel<5>,
                                           igemm8_f16_kernel<6>, igemm8_f16_kernel<7>, igemm8_f16_kernel<8>};

static const f16* s_zero_page = nullptr;

int igemm8_init() {
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess || !zp) return MOFA_ELAUNCH;
    s_zero_page = (const f16*)zp;
    for (igemm8_kern_t k : k_igemm8)
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
    return MOFA_OK;
}

// q = umulhi(n, mul) >> shr == n / d for 0 <= n < 2^31 (mul == 0: d == 1)
static FastDiv fastdiv_make(int d) {
    FastDiv f = {0, 0};
    if (d > 1) {
        unsigned lg = 0;
        while ((1u << lg) < (unsigned)d) ++lg;                     // ceil(log2 d)
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + (unsigned)d - 1) / (unsigned)d);
        f.shr = p - 32;
    }
    return f;
}

// 16-byte row alignment everywhere (the kernel has no narrow-store path); packed row geometry and 32-bit offsets in range
static bool igemm8_eligible(const mofa_igemm_args* a, int kind, long long Ktot) {
    const int nout = kind == 8 ? a->N / 2 : a->N;
    if ((a->ldo & 7) || (nout & 7) || (a->N & 7) || (((size_t)a->out) & 15) || (((size_t)a->x) & 15)) return false;
    if (a->r1 && ((a->ldr1 & 7) || (((size_t)a->r1) & 15))) return false;
    if (a->r2 && ((a->ldr2 & 7) || (((size_t)a->r2) & 15))) return false;
    if (a->bias && (((size_t)a->bias) & 15)) return false;
    if (a->rowvec && (((size_t)a->rowvec) & 15)) return false;
    if ((long long)a->N * Ktot * 2 >= (1ll << 32) || (((size_t)a->w) & 15)) return false;
    long long rows_in = a->M;
    if (a->mode == MOFA_MODE_CONV3X3) {
        const long long nimg = a->M / ((long long)a->Hout * a->Wout);
        if (a->Hout > 1024 || a->Wout > 1024 || nimg > 2047) return false;
        rows_in = nimg * a->Hin * a->Win;
    } else if (a->mode == MOFA_MODE_CONVT3) {
        if (a->M >= (1 << 29)) return false;
        rows_in = (long long)a->M + a->HW;
    }
    if (rows_in * (a->ldx / 8) >= (1ll << 31)) return false;
    return true;
}

int igemm8_launch(const mofa_igemm_args* a, int kind, int n_cu, hipStream_t stream) {
    const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
    if (!igemm8_eligible(a, kind, (long long)taps * a->Cin)) return 1;   // the caller falls back to a 4-wave tile
    const int tilesM = cdiv(a->M, TBM), tilesN = cdiv(a->N, TBN);
    const long long nt = (long long)tilesM * tilesN;
    if (nt > 0x7fffffffLL) return MOFA_EINVAL;
    Aux aux;
    aux.tiles_n = fastdiv_make(tilesN);
    aux.hw = fastdiv_make(a->mode == MOFA_MODE_CONV3X3 ? a->Hout * a->Wout : 1);
    aux.wout = fastdiv_make(a->mode == MOFA_MODE_CONV3X3 ? a->Wout : 1);
    aux.t3hw = fastdiv_make(a->mode == MOFA_MODE_CONVT3 ? a->HW : 1);
    aux.t3t = fastdiv_make(a->mode == MOFA_MODE_CONVT3 && a->T > 0 ? a->T : 1);
    aux.ldx16 = a->ldx / 8;
    aux.zero = s_zero_page;
    int grid = (int)(nt < n_cu ? ((nt + 7) / 8) * 8 : (n_cu / 8) * 8);
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(k_igemm8[kind], dim3(grid), dim3(512), LDS_BYTES, stream, *a, tilesN, (int)nt, aux);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
