// Implicit GEMM, 256x256 output tile, 8 waves, phase-pipelined K loop (the workhorse tile of the denoise loop).
//
// Same contract as igemm.hip (see mofa_hip.h: out = act(s_acc * (conv + bias + rowvec) + s1 * r1 + s2 * r2)) and the
// same operand roles (MFMA A = weight tile, B = activation tile: an accumulator lane owns one output row), LDS image
// (128-byte rows, XOR swizzle on the DMA source address and on the fragment read) and persistent XCD-aware tile walk.
// What differs is the K loop, built after the 8-phase schedule of the CDNA4 guide (cdna_hip_programming.md "The 256^2
// 8-phase template", T3+T4+T5):
//
//   * wave (wm, wn) of 2 x 4 owns 128 (M) x 64 (N) outputs = 4 x 2 accumulator tiles of 32x32.  A K tile (64 deep) is
//     consumed in TWO PHASES of 16 MFMAs (512 matrix-pipe cycles) each:
//         phase 1: X half 0 (accumulator rows 0-1) x W     reads X0 (8 ds_read_b128) + W (8)
//         phase 2: X half 1 (accumulator rows 2-3) x W     reads X1 (8)                       (W stays in registers)
//     A phase is  {fragment reads, LDS-DMA issue, s_waitcnt vmcnt(8) lgkmcnt(0)} s_barrier {16 MFMAs} s_barrier.
//     (Four phases of 8 MFMAs with quarter-tile DMA were measured in round 2: the same rate with twice the barriers.)
//   * the two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so on every
//     SIMD one wave is in its MFMA segment while its partner reads fragments and issues DMA; s_setprio(1) around the
//     MFMA cluster lets the matrix pipe win the issue arbitration.
//   * the LDS ring is 2 K tiles x (X 256 rows + W 256 rows) x 128 B = 2 x 64 KB.  A part is released as soon as its
//     fragments are in registers (X0 and W after phase 1, X1 after phase 2) and refilled in the NEXT phase:
//         phase 1: X1 of K tile t+1 (2 DMA instructions per thread)     phase 2: X0 + W of K tile t+2 (2 + 4)
//     i.e. 8 DMA instructions (1 KB each per wave) per thread and K tile, every one with about one K tile of flight
//     before the counted wait `vmcnt(8)` (= the DMA of the last two phases may stay outstanding) retires it; it is read
//     no earlier than the phase after that wait (the wait precedes the phase's first barrier; with the two wave groups
//     one barrier apart that is the barrier after which the other group's reads start -- guide: "read a staged buffer
//     one phase AFTER the wait that retires it").  vmcnt is never drained inside the loop, and the stream runs on ACROSS
//     output tiles: while a tile's epilogue runs, the first K tiles of the workgroup's next output tile are in flight.
//   * two producer cursors (X0 + W, X1) walk the same persistent tile sequence as the consumer, 1 and 2 K tiles ahead;
//     past the end of the walk their rows carry the out-of-range offset (zeros through the buffer descriptor's bounds
//     check) so that the instruction count stays static.
//
// Epilogue: private 4 KB LDS scratch per wave OUTSIDE the ring (160 KB = 2 x 64 KB ring + 8 x 4 KB), so nothing in the
// epilogue aliases a DMA target and no barrier is needed inside it.  Kinds without residuals (plain, row vector, GEGLU)
// apply bias / row vector / activation in the accumulator (fragment) layout, convert to fp16 and transpose 32 x 64 fp16
// through the scratch (ds_write_b64 / ds_read_b128, both conflict-free: tools/lds_bank_sim.py) -- half the LDS write
// traffic of an fp32 transpose, which is what bounds the epilogue (ds_write ~ 80 B/clk/CU).  Kinds with residuals keep
// fp32 through the transpose (32 x 32 fp32 per pass) so that the sum is rounded once.
// Eligibility (checked by the launcher): 16-byte aligned rows everywhere ("wide" path of igemm.hip), N % 8 == 0.
//
// Build variants: the K-loop timing probes and the per-segment cycle trace (VAR != 0, tools/igemm8_probe.py) are compiled
// only with -DMOFA_PROBE into tools/libmofa_hip_probe.so; the product library holds the nine VAR = 0 kernels alone.
#include "igemm_common.h"
#include "igemm_pipe.h"

namespace {

constexpr int WM = 2, WN = 4, MI = 4, NJ = 2;                  // waves, accumulator tiles per wave
constexpr int TBM = WM * MI * 32, TBN = WN * NJ * 32;          // 256 x 256
constexpr int BKS = 64, RB = 128;                              // K per tile, LDS row bytes
constexpr int SXB = TBM * RB, STB = SXB + TBN * RB;            // bytes: X part, one K tile of the ring
constexpr int SCRATCH0 = 2 * STB, SCRATCH_WAVE = 4096;
constexpr int LDS_BYTES = SCRATCH0 + 8 * SCRATCH_WAVE;         // 163840 = the CU's whole LDS
constexpr int LOOKAHEAD_OPS = 8;                               // DMA instructions of the last 2 phases may be in flight

// epilogue scratch images (tools/lds_bank_sim.py)
__device__ __forceinline__ int h16_off(int r, int c8) {        // 32 rows x 64 fp16; c8 = 8-byte chunk (4 columns)
    return r * 128 + (((c8 >> 1) ^ ((r >> 1) & 7)) << 4) + (((c8 & 1) ^ (r & 1)) << 3);
}
__device__ __forceinline__ int f32_off(int r, int c) {         // 32 rows x 32 fp32; c = 16-byte chunk (4 columns)
    return r * 128 + ((c ^ (((r >> 1) & 3) | ((r & 1) << 2))) << 4);
}

struct Cursor {                     // one half-tile pair (X half h, W half h) of the persistent K-tile stream
    int local;                      // walk position of the output tile it is in
    int ikc, ksw, ky, kx;           // K tile within the tap / overall, tap coordinates (convT3: ky = tap)
    // per DMA piece (named scalars, not arrays: hipcc otherwise keeps part of the struct in scratch memory)
    int gx0, gx1;                   // packed row geometry
    unsigned xo0, xo1;              // source row of the current tap + this lane's swizzled chunk, bytes from aux.xbase
    unsigned wo0, wo1, wo2, wo3;    // (W stream only) weight row + this lane's swizzled chunk, bytes from a.w
};

// VAR: diagnostic build variants of the K loop (tools/igemm8_probe.py; the library's launches use VAR = 0):
//   1 no stagger between the wave groups   2 no s_setprio   4 DMA issue before the fragment reads   8 no vmcnt wait
//   16 no DMA inside the loop   32 no fragment reads inside the loop   64 per-segment cycle trace into aux.trace
// (8, 16, 32 give wrong results: timing probes only)
template <int EPI, int VAR = 0>
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
    // (lane-derived values are recomputed where they are used -- per tap / per tile -- instead of living in VGPRs)
    auto lane_now = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto swz_bytes = [&](int l, int q) __attribute__((always_inline)) { return (((l & 7) ^ (4 * q + (l >> 4))) * 16); };
    auto x_row = [&](int h, int q, int hr0) __attribute__((always_inline)) { const int hr = hr0 + 8 * q; return (hr >> 6) * 128 + h * 64 + (hr & 63); };
    auto w_row = [&](int h, int q, int hr0) __attribute__((always_inline)) { const int hr = hr0 + 8 * q; return (hr >> 5) * 64 + h * 32 + (hr & 31); };
    const int ks_ = a.ksize > 0 ? a.ksize : 3, dil_ = a.dil > 0 ? a.dil : 1;
    const int org_ = a.pad == MOFA_PAD_TRAILING ? 0 : (ks_ >> 1);
    // (single-exit lambdas: with several return statements hipcc leaves the result slots in scratch memory)
    auto pack_geo = [&](int m) __attribute__((always_inline)) -> int {
        int g = m;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int img = fdiv(m, aux.hw), rem = m - img * (a.Hout * a.Wout);
            const int oy = fdiv(rem, aux.wout);
            g = (img << 20) | (oy << 10) | (rem - oy * a.Wout);
        } else if (a.mode == MOFA_MODE_CONVT3) {
            int lo = 1, hi = 1;
            if (a.T > 0) {
                const int fr = fdiv(m, aux.t3hw);                  // frame index; its position within the clip of T
                const int f = fr - fdiv(fr, aux.t3t) * a.T;
                lo = f > 0; hi = f < a.T - 1;
            }
            g = m | (lo << 29) | (hi << 30);
        }
        return m < a.M ? g : -1;
    };
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)aux.xbase, 0, aux.x_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, aux.w_bytes, 0x00020000);
    auto bglds16 = [&](const decltype(rsx)& rs, unsigned voff, int soff, char* lds_wave_base) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
    };
    auto tap_src = [&](int g, int ky, int kx, int swzb) __attribute__((always_inline)) -> unsigned {   // bytes from aux.xbase
        int row = g;                                               // plain: the output row itself
        bool ok = g >= 0;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int vy = ((g >> 10) & 1023) * a.stride + (ky - org_) * dil_;
            const int vx = (g & 1023) * a.stride + (kx - org_) * dil_;
            ok = ok && vy >= 0 && vx >= 0 && vy < a.Hin * a.up && vx < a.Win * a.up;
            const int iy = (a.up == 2) ? (vy >> 1) : vy, ix = (a.up == 2) ? (vx >> 1) : vx;
            row = ((g >> 20) * a.Hin + iy) * a.Win + ix;
        } else if (a.mode == MOFA_MODE_CONVT3) {                   // tap ky - 1 frames away
            ok = ok && !(ky == 0 && !((g >> 29) & 1)) && !(ky == 2 && !((g >> 30) & 1));
            row = (g & 0x1fffffff) + (ky - 1) * a.HW + aux.row_shift;
        }
        const unsigned off = (unsigned)row * (unsigned)aux.ldxb + (unsigned)swzb;
        return ok ? off : XO_INVALID;
    };
    auto cur_setup = [&](Cursor& c, const int h, const bool with_w) __attribute__((always_inline)) {
        // branch-free on purpose (selects on the uniform `live`): stores to the cursor's fields from two arms of an
        // if / else get merged into a store through a pointer phi, which pins those fields to scratch memory
        const bool live = c.local < walk.count;
        const int tile = walk.start + (live ? c.local : 0);
        const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
        const int l = lane_now(), hr_l = 16 * wave + (l >> 3);
        const int g0 = pack_geo(tm * TBM + x_row(h, 0, hr_l)), g1 = pack_geo(tm * TBM + x_row(h, 1, hr_l));
        c.gx0 = live ? g0 : -1;
        c.gx1 = live ? g1 : -1;
        if (with_w) {
            auto w_off = [&](int hq) __attribute__((always_inline)) {
                int n = tn * TBN + w_row(hq >> 1, hq & 1, hr_l);
                n = n < a.N ? n : a.N - 1;
                const unsigned o = (unsigned)n * (unsigned)(Ktot * 2) + swz_bytes(l, hq & 1);
                return live ? o : XO_INVALID;
            };
            c.wo0 = w_off(0); c.wo1 = w_off(1); c.wo2 = w_off(2); c.wo3 = w_off(3);
        }
        c.ikc = 0; c.ksw = 0; c.ky = 0; c.kx = 0;
    };
    auto issue_x = [&](Cursor& c, const int h, const int bo) __attribute__((always_inline)) {
        if (c.ikc == 0) {
            const int l = lane_now();
            c.xo0 = tap_src(c.gx0, c.ky, c.kx, swz_bytes(l, 0));
            c.xo1 = tap_src(c.gx1, c.ky, c.kx, swz_bytes(l, 1));
        }
        bglds16(rsx, c.xo0, c.ikc * (BKS * 2), smem + bo + x_row(h, 0, 16 * wave) * RB);
        bglds16(rsx, c.xo1, c.ikc * (BKS * 2), smem + bo + x_row(h, 1, 16 * wave) * RB);
    };
    auto issue_w = [&](Cursor& c, const int bo) __attribute__((always_inline)) {                  // both W halves: 4 pieces
        bglds16(rsw, c.wo0, c.ksw * (BKS * 2), smem + bo + SXB + w_row(0, 0, 16 * wave) * RB);
        bglds16(rsw, c.wo1, c.ksw * (BKS * 2), smem + bo + SXB + w_row(0, 1, 16 * wave) * RB);
        bglds16(rsw, c.wo2, c.ksw * (BKS * 2), smem + bo + SXB + w_row(1, 0, 16 * wave) * RB);
        bglds16(rsw, c.wo3, c.ksw * (BKS * 2), smem + bo + SXB + w_row(1, 1, 16 * wave) * RB);
    };
    auto advance = [&](Cursor& c, const int h, const bool with_w) __attribute__((always_inline)) {
        ++c.ksw;
        if (++c.ikc == kpt) {
            c.ikc = 0;
            if (a.mode == MOFA_MODE_CONV3X3) { if (++c.kx == ks_) { c.kx = 0; ++c.ky; } } else ++c.ky;
        }
        if (c.ksw == nk) { c.local += walk.stride; cur_setup(c, h, with_w); }
    };

    // ---- consumer side ----------------------------------------------------------------------------------------------
    // fragment read addresses of ring slot 0 (one register per 16-deep K step; accumulator tile / ring slot 1 are
    // immediates / a flipped bit 16)
    int xa[4], wa[4];
    {
        const int fsw = (l31 >> 1) & 7;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int slot = ((kk * 2 + lh) ^ fsw) * 16;
            xa[kk] = (wm * MI * 32 + l31) * RB + slot;
            wa[kk] = SXB + (wn * NJ * 32 + l31) * RB + slot;
        }
    }

    // The wave's 64 bias values travel through its LDS scratch: one 4-byte-per-lane DMA at the START of the tile (the
    // scratch is idle during the K loop; columns beyond N read as zero through the descriptor's bounds check), so the
    // epilogue starts with LDS reads instead of a global-memory round trip.
    char* scr = smem + SCRATCH0 + wave * SCRATCH_WAVE;
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.N * 4u : 0u, 0x00020000);
    auto bias_prefetch = [&](int nw, int slot = 0) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(scr + slot * 256), 4,
                                                 (unsigned)(nw + lane_now()) * 4u, 0, 0, 0);
    };
    // GEGLU kind (BIAS_INIT): the bias is the INITIAL VALUE of the accumulators, as in igemm320.hip -- its epilogue is bound by
    // VALU issue (about 15 instructions per output; the two bias adds were 2 of them) and does not transpose through the
    // scratch, so the scratch can hold two 256-byte bias slots: the DMA issued at the start of tile t brings tile t + 1's
    // bias (same instruction count per tile as before: the counted waits do not change), retired by the second phase's
    // counted wait of tile t; the first tile's bias is the oldest operation of the prologue.
    constexpr bool BIAS_INIT = GEGLU;

    Cursor ca, cb;                                                 // (X0, W0, W1) stream, X1 stream
    ca.local = cb.local = walk.local;
    cur_setup(ca, 0, true);
    cur_setup(cb, 1, false);
    // prologue: (BIAS_INIT: the first tile's bias,) K tile 0 complete, (X0, W) of K tile 1: 6 + 2 + 6 DMA instructions
    if constexpr (BIAS_INIT) {
        const int tile0 = walk.start + walk.local;
        bias_prefetch((tile0 - fdiv(tile0, aux.tiles_n) * tilesN) * TBN + wn * NJ * 32, 0);
    }
    issue_x(ca, 0, 0); issue_w(ca, 0); advance(ca, 0, true);
    issue_x(cb, 1, 0); advance(cb, 1, false);
    issue_x(ca, 0, STB); issue_w(ca, STB); advance(ca, 0, true);
    wait_vmcnt_only<LOOKAHEAD_OPS>();                              // X0, W of K tile 0 have landed
    __builtin_amdgcn_s_barrier();

    // ---- one phase of a K tile: {fragment reads, DMA, counted wait} barrier {16 MFMAs on 4 accumulators} barrier
    f32x16 acc[MI][NJ];
    f16x8 xf[2][4], wf[2][4];
    unsigned long long tr[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0;
    if (VAR & 64) tr_t = __builtin_readcyclecounter();
    if (VAR & 32) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            xf[0][kk] = *(const f16x8*)(smem + xa[kk]);
            xf[1][kk] = *(const f16x8*)(smem + xa[kk] + 32 * RB);
            wf[0][kk] = *(const f16x8*)(smem + wa[kk]);
            wf[1][kk] = *(const f16x8*)(smem + wa[kk] + 32 * RB);
        }
    }
    auto stamp = [&](int k) __attribute__((always_inline)) {
        if (VAR & 64) { const unsigned long long t = __builtin_readcyclecounter(); tr[k] += t - tr_t; tr_t = t; }
    };
    auto phase = [&](auto pc, const int bo, const int bn) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::v;                         // 1: X half 0 x W, 2: X half 1 x W
        auto reads = [&]() __attribute__((always_inline)) {
            if (VAR & 32) return;
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    xf[il][kk] = *(const f16x8*)(smem + xa[kk] + ((P == 2 ? 2 : 0) + il) * 32 * RB);
            if constexpr (P == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) wf[j][kk] = *(const f16x8*)(smem + wa[kk] + j * 32 * RB);
            }
        };
        auto issues = [&]() __attribute__((always_inline)) {
            if (VAR & 16) return;
            if constexpr (P == 1) issue_x(cb, 1, bn);              // X1 of the next K tile
            if constexpr (P == 2) { issue_x(ca, 0, bo); issue_w(ca, bo); }   // X0, W0, W1 of the K tile after it
        };
        if (VAR & 4) { issues(); __builtin_amdgcn_sched_barrier(0); reads(); }
        else { reads(); __builtin_amdgcn_sched_barrier(0); issues(); }
        if constexpr (P == 1) advance(cb, 1, false);
        if constexpr (P == 2) advance(ca, 0, true);
        // DMA older than the last two phases has landed (read from the next phase on); this phase's fragment reads have
        // returned BEFORE the barrier, so their LDS rows may be refilled from the next phase on
        if (!(VAR & 8)) wait_vmcnt<LOOKAHEAD_OPS>(); else wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp(3 * (P - 1));
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    constexpr int I0 = (P == 2) ? 2 : 0;
                    acc[I0 + il][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][kk], xf[il][kk], acc[I0 + il][j], 0, 0, 0);
                }
        if (!(VAR & 2)) __builtin_amdgcn_s_setprio(0);
        stamp(3 * (P - 1) + 1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp(3 * (P - 1) + 2);
    };

    // ---- epilogue pieces -------------------------------------------------------------------------------------------------
    auto act_apply = [&](auto ac, float v) __attribute__((always_inline)) -> float {
        constexpr int ACT = decltype(ac)::v;
        if constexpr (ACT == MOFA_ACT_SILU) return silu_f(v);
        else if constexpr (ACT == MOFA_ACT_RELU) return fmaxf(v, 0.0f);
        else if constexpr (ACT == MOFA_ACT_GELU) return gelu_erf_f(v);
        else return v;
    };
    // kinds without residuals (plain, row vector, GEGLU): bias / row vector / activation in the accumulator (fragment)
    // layout -- register r of accumulator tile (i, j) is row 32 i + l31, column 32 j + 8 (r >> 2) + 4 lh + (r & 3) --
    // then fp16 and a 32 x 64 fp16 transpose through the scratch; a lane stores 8 consecutive columns of one row
    // UNI (row-vector kinds): every row of the wave's 128 x 64 block takes the SAME row of the row-vector table (time
    // embedding per frame / cross-attention vector per clip: true for every tile that does not straddle a frame), so its 64
    // values are loaded ONCE per tile.  Per-row loads in the fragment layout are 16 load instructions per 32-row block and
    // wave, 16 bytes per lane each: 8 waves x 4 blocks of them keep the CU's one texture path busy for >= 8 000 cycles per
    // tile, which is what the row-vector epilogue cost (15-19 k cycles against 5 k for the plain one).
    auto epilogue_light = [&](auto ac, auto un, f32x16 (&acc)[MI][NJ], const int mw, const int nw, const int idx_u) __attribute__((always_inline)) {
        constexpr bool UNI = decltype(un)::v != 0;
        constexpr bool SACC1 = GEGLU && UNI;                      // (GEGLU kind: `un` says s_acc == 1, the multiplies drop out)
        // lane-derived LDS / global offsets are recomputed per tile from a laundered lane id: hoisted out of the tile loop
        // they stay live across the K loop and get spilled to scratch (and reloaded here, latency-bound)
        // (row-vector kinds only, the ones that spilled: elsewhere the hoisted offsets fit and recomputing them costs 500-900
        // cycles per tile)
        const int lane_e = RV ? lane_now() : lane;
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        float saccv = a.s_acc;                                     // VGPR operand on purpose (see igemm.hip's epilogue)
        asm volatile("" : "+v"(saccv));
        f32x4 bv[NJ][4];                                           // bias (from the scratch); UNI: bias + row vector
        if (!BIAS_INIT && (!RV || UNI)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[j][g] = *(const f32x4*)(scr + (32 * j + 8 * g + 4 * lh) * 4);
        }
        const int nout = GEGLU ? a.N / 2 : a.N;
        int rv_div = a.rv_div, rv_mod_in = a.rv_mod_in, rv_mod_out = a.rv_mod_out;
        asm volatile("" : "+s"(rv_div), "+s"(rv_mod_in), "+s"(rv_mod_out));   // reciprocals are set up here, not hoisted
        // not UNI: row-vector values of accumulator row block i; block i + 1's are loaded as soon as block i's arithmetic
        // is done (same registers), so the load latency hides behind block i's transpose and stores
        f32x4 rv[NJ][4];
        auto rv_load = [&](int i, f32x4 (&rv)[NJ][4]) __attribute__((always_inline)) {
            int m = mw + 32 * i + l31;
            m = m < a.M ? m : a.M - 1;
            const int idx = UNI ? idx_u : ((m / rv_div) * a.rv_mul + (m % rv_mod_in)) % rv_mod_out;
            const float* rvp = a.rowvec + (size_t)idx * a.N;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int n = nw + 32 * j + 8 * g + 4 * lh;
                    n = n + 4 <= a.N ? n : 0;                       // columns beyond N are never stored
                    rv[j][g] = *(const f32x4*)(rvp + n);
                    if (!UNI && a.bias) rv[j][g] += *(const f32x4*)(a.bias + n);   // (per-row path: the bias rides along)
                }
        };
        if (RV) rv_load(0, rv);
        if (RV && UNI) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) bv[j][g] += rv[j][g];
        }
        // GEGLU kind (32 output columns per row and accumulator row block): stores without an LDS transpose.  After fp16 packing a lane holds columns 8 g + 4 lh .. + 3 (g = 0..3) of row l31 of
        // a 32 x 32 accumulator block; one v_permlane32_swap per dword between the registers of groups 2 k and 2 k + 1
        // (upper half of the first <-> lower half of the second) leaves lanes 0-31 with columns 16 k .. 16 k + 7 and lanes
        // 32-63 with 16 k + 8 .. 16 k + 15 of their row: ONE 16-byte store per lane and group pair, 32 bytes per row and
        // instruction, a row's 128-byte line completed by 4 consecutive instructions of the same wave.
        auto store_pair = [&](f16x4 g0, f16x4 g1, int m, int n0) __attribute__((always_inline)) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x2 a2 = __builtin_bit_cast(u32x2, g0), b2 = __builtin_bit_cast(u32x2, g1);
            const auto rx = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
            const u32x4 v = {rx[0], ry[0], rx[1], ry[1]};
            const int n = n0 + 8 * lh;
            if (m < a.M && n + 8 <= nout) *(u32x4*)(out + (size_t)m * a.ldo + n) = v;
        };
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = mw + 32 * i + l31;
            if constexpr (GEGLU) {
                // weight rows are interleaved in blocks of 16 at load time (weights.interleave_geglu): accumulator tile j holds
                // the value columns of outputs 16 j .. 16 j + 15 in registers 0-7 and their gate columns in registers 8-15 of
                // the same lane; output group g (columns 8 g + 4 lh + e of the wave's 32) = tile g >> 1, half g & 1
                f16x4 o[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = acc[i][g >> 1][4 * (g & 1) + e];          // (BIAS_INIT: the bias is already in)
                        const float gate = acc[i][g >> 1][8 + 4 * (g & 1) + e];
                        if constexpr (SACC1) o[g][e] = (f16)(val * (gate * gelu_phi_f(gate)));
                        else o[g][e] = (f16)(saccv * val * gelu_erf_f(saccv * gate));
                    }
                store_pair(o[0], o[1], m, nw / 2);
                store_pair(o[2], o[3], m, nw / 2 + 16);
            } else {
                // 64 output columns per row: through the scratch (32 x 64 fp16 transpose), so that a store instruction
                // covers 8 whole 128-byte lines.  (The register-only form above was measured here too: 16 instructions of
                // 32 partial lines each cost 7 500 cycles per tile against 5 300 -- the texture path pays per line.)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][j][4 * g + e] + (RV && !UNI ? rv[j][g][e] : bv[j][g][e]);
                            o[e] = (f16)act_apply(ac, saccv * v);
                        }
                        *(f16x4*)(scr + h16_off(l31, 8 * j + 2 * g + lh)) = o;
                    }
                if (RV && !UNI && i + 1 < MI) rv_load(i + 1, rv);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), blk = lane & 7;
                    const f16x8 v = *(const f16x8*)(scr + row * 128 + ((blk ^ ((row >> 1) & 7)) << 4));
                    const int mr = mw + 32 * i + row, n = nw + 8 * blk;
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    if (mr < a.M && n + 8 <= nout) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
                }
            }
        }
    };
    // kinds with residuals: fp32 transpose, 32 x 32 per step (8 steps: accumulator tiles (i, j)); a lane owns 8 consecutive
    // columns of one row.  The residual / row-vector loads of step s + D are issued before step s is processed (a ring of
    // D steps in registers; D from the registers a step's loads take), so their latency hides behind D steps of work.
    auto epilogue_residual = [&](auto un, f32x16 (&acc)[MI][NJ], const int mw, const int nw, const int idx_u) __attribute__((always_inline)) {
        constexpr bool UNI = decltype(un)::v != 0;                 // one row-vector row for the whole wave block (see above)
        constexpr int STEPS = MI * NJ;
        constexpr int REGS = (R1 ? 8 : 0) + (R2 ? 8 : 0) + (RV && !UNI ? 16 : 0);   // VGPRs of one step's loads
        constexpr int D0 = REGS <= 8 ? 8 : (REGS <= 16 ? 4 : (REGS <= 24 ? 2 : 1));
        constexpr int D = RV && UNI ? (REGS <= 8 ? 4 : 2) : (RV ? 1 : D0);   // (row-vector kinds: 16 more registers are taken)
        const int lane_e = RV ? lane_now() : lane;                 // (recomputed per tile, see epilogue_light)
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        const f16* r1 = (const f16*)a.r1;
        const f16* r2 = (const f16*)a.r2;
        float s1v = a.s1, s2v = a.s2, saccv = a.s_acc;
        asm volatile("" : "+v"(s1v), "+v"(s2v), "+v"(saccv));
        int rv_div = a.rv_div, rv_mod_in = a.rv_mod_in, rv_mod_out = a.rv_mod_out;
        asm volatile("" : "+s"(rv_div), "+s"(rv_mod_in), "+s"(rv_mod_out));
        const int piece = lane & 3;
        struct StepLoads { f16x8 t1[2], t2[2]; f32x4 rv0[2], rv1[2]; };
        StepLoads L[D];
        auto step_loads = [&](int st, StepLoads& l) __attribute__((always_inline)) {
            const int i = st / NJ, j = st % NJ;
            const int n = nw + 32 * j + 8 * piece;
            const int nc = n + 8 <= a.N ? n : 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                int m = mw + 32 * i + 16 * p + (lane >> 2);
                m = m < a.M ? m : a.M - 1;
                if (R1) l.t1[p] = *(const f16x8*)(r1 + (size_t)m * a.ldr1 + nc);
                if (R2) l.t2[p] = *(const f16x8*)(r2 + (size_t)m * a.ldr2 + nc);
                if (RV && !UNI) {
                    const int idx = ((m / rv_div) * a.rv_mul + (m % rv_mod_in)) % rv_mod_out;
                    const float* q = a.rowvec + (size_t)idx * a.N + nc;
                    l.rv0[p] = *(const f32x4*)q;
                    l.rv1[p] = *(const f32x4*)(q + 4);
                }
            }
        };
        f32x4 rvu[NJ][2];                                          // UNI: the row vector of this lane's 8 columns
        if (RV && UNI) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nw + 32 * j + 8 * piece;
                const float* q = a.rowvec + (size_t)idx_u * a.N + (n + 8 <= a.N ? n : 0);
                rvu[j][0] = *(const f32x4*)q;
                rvu[j][1] = *(const f32x4*)(q + 4);
            }
        }
#pragma unroll
        for (int st = 0; st < D && st < STEPS; ++st) step_loads(st, L[st]);
        f32x4 bv[NJ][2];                                           // bias of this lane's 8 columns, from the scratch
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bv[j][0] = *(const f32x4*)(scr + (32 * j + 8 * piece) * 4);
            bv[j][1] = *(const f32x4*)(scr + (32 * j + 8 * piece + 4) * 4);
            if (RV && UNI) { bv[j][0] += rvu[j][0]; bv[j][1] += rvu[j][1]; }
        }
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int i = st / NJ, j = st % NJ;
            StepLoads& l = L[st % D];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                *(f32x4*)(scr + f32_off(l31, 2 * g + lh)) = v;
            }
            const int n = nw + 32 * j + 8 * piece;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = 16 * p + (lane >> 2);
                const f32x4 v0 = *(const f32x4*)(scr + f32_off(row, 2 * piece));
                const f32x4 v1 = *(const f32x4*)(scr + f32_off(row, 2 * piece + 1));
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = v0[e] + bv[j][0][e], x1 = v1[e] + bv[j][1][e];
                    if (RV && !UNI) { x0 += l.rv0[p][e]; x1 += l.rv1[p][e]; }
                    x0 *= saccv; x1 *= saccv;
                    // (the layer's own output is rounded to fp16 BEFORE a residual add on every tile kernel, like the
                    //  reference's fp16 modules and igemm320's epilogue_res16: the tile choice never changes the bits)
                    if (R1 || R2) { x0 = (float)(f16)x0; x1 = (float)(f16)x1; }
                    if (R1) { x0 += s1v * (float)l.t1[p][e]; x1 += s1v * (float)l.t1[p][4 + e]; }
                    if (R2) { x0 += s2v * (float)l.t2[p][e]; x1 += s2v * (float)l.t2[p][4 + e]; }
                    o[e] = (f16)x0; o[4 + e] = (f16)x1;
                }
                const int m = mw + 32 * i + row;
                if (m < a.M && n + 8 <= a.N) *(f16x8*)(out + (size_t)m * a.ldo + n) = o;
            }
            if (st + D < STEPS) step_loads(st + D, L[st % D]);     // refill the ring slot just consumed
        }
    };

    int gt = 0;                                                    // K tiles consumed so far (ring slot = gt & 1)
    int bslot = 0;                                                 // (BIAS_INIT) scratch slot holding this tile's bias
    for (int cl = walk.local; cl < walk.count; cl += walk.stride, bslot ^= 1) {
        const int tile = walk.start + cl;
        const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
        if (VAR & 64) { stamp(9); tr[12] += nk; }          // tile set-up since the last stamp
        if constexpr (BIAS_INIT) {
            // the NEXT tile's bias into the other slot (no next tile: this tile's once more, never read -- the instruction
            // count per tile stays static), then the accumulators start at this tile's bias
            int tn2 = tn;
            if (cl + walk.stride < walk.count) {
                const int tile2 = walk.start + cl + walk.stride;
                tn2 = tile2 - fdiv(tile2, aux.tiles_n) * tilesN;
            }
            bias_prefetch(tn2 * TBN + wn * NJ * 32, bslot ^ 1);
            const int lh_e = lane_now() >> 5;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *(const f32x4*)(scr + bslot * 256 + (32 * j + 8 * g + 4 * lh_e) * 4);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = b[e];
                }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

            bias_prefetch(tn * TBN + wn * NJ * 32);                // (no bias: an empty descriptor, zeros arrive)
        }
        if (!(VAR & 1) && grp == 1) __builtin_amdgcn_s_barrier();  // second wave group runs one barrier behind
        for (int kt = 0; kt < nk; ++kt, ++gt) {
            const int bo = (gt & 1) * STB, bn = STB - bo;          // ring slot of this K tile / of the next one
            phase(IC<1>{}, bo, bn);
            phase(IC<2>{}, bo, bn);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { xa[kk] ^= STB; wa[kk] ^= STB; }   // the next K tile's ring slot
        }
        if (!(VAR & 1) && grp == 0) __builtin_amdgcn_s_barrier();  // both groups meet again: equal barrier counts per tile
        stamp(6);                                                  // (re-sync barrier of the first wave group)

        // ---- epilogue (no barriers; private scratch) ------------------------------------------------------------------
        const int mw = tm * TBM + wm * MI * 32;                    // first output row / (pre-GEGLU) column of this wave
        const int nw = tn * TBN + wn * NJ * 32;
        int idx_u = -1;                                            // >= 0: the one row-vector row of this wave's block
        if constexpr (RV) {
            if (a.rv_mod_in == 1) {                                // idx(m) = ((m / div) * mul) % mod_out: a step function
                int rv_div = a.rv_div;
                asm volatile("" : "+s"(rv_div));
                const int m0 = mw < a.M ? mw : a.M - 1, m1 = mw + 32 * MI - 1 < a.M ? mw + 32 * MI - 1 : a.M - 1;
                const int q0 = m0 / rv_div, q1 = m1 / rv_div;
                if (q0 == q1) idx_u = (q0 * a.rv_mul) % a.rv_mod_out;
            }
            idx_u = __builtin_amdgcn_readfirstlane(idx_u);
        }
        if constexpr (R1 || R2) {
            if (RV && idx_u >= 0) epilogue_residual(IC<1>{}, acc, mw, nw, idx_u);
            else epilogue_residual(IC<0>{}, acc, mw, nw, 0);
        } else if constexpr (GEGLU) {
            if (a.s_acc == 1.0f) epilogue_light(IC<MOFA_ACT_NONE>{}, IC<1>{}, acc, mw, nw, 0);
            else epilogue_light(IC<MOFA_ACT_NONE>{}, IC<0>{}, acc, mw, nw, 0);
        } else if constexpr (RV) {                                 // (row vector + activation launches run on the 4-wave tile)
            if (idx_u >= 0) epilogue_light(IC<MOFA_ACT_NONE>{}, IC<1>{}, acc, mw, nw, idx_u);
            else epilogue_light(IC<MOFA_ACT_NONE>{}, IC<0>{}, acc, mw, nw, 0);
        } else {
            if (a.act == MOFA_ACT_NONE) epilogue_light(IC<MOFA_ACT_NONE>{}, IC<0>{}, acc, mw, nw, 0);
            else if (a.act == MOFA_ACT_SILU) epilogue_light(IC<MOFA_ACT_SILU>{}, IC<0>{}, acc, mw, nw, 0);
            else if (a.act == MOFA_ACT_RELU) epilogue_light(IC<MOFA_ACT_RELU>{}, IC<0>{}, acc, mw, nw, 0);
            else epilogue_light(IC<MOFA_ACT_GELU>{}, IC<0>{}, acc, mw, nw, 0);
        }
        if (VAR & 64) { stamp(7); tr[8] += 1; }                   // epilogue (stores still in flight); tiles
    }
    // a wave must not retire with an LDS DMA in flight (the cursors' run-ahead past the last tile)
    wait_vmcnt_only<0>();
    if ((VAR & 64) && aux.trace && (tid & 255) == 0) {
#pragma unroll
        for (int k = 0; k < 13; ++k) aux.trace[((size_t)blockIdx.x * 2 + grp) * 16 + k] = tr[k];
    }
}

typedef void (*igemm8_kern_t)(const mofa_igemm_args, const int, const int, const Aux);

}  // namespace

static const igemm8_kern_t k_igemm8[9] = {igemm8_f16_kernel<0>, igemm8_f16_kernel<1>, igemm8_f16_kernel<2>,
                                           igemm8_f16_kernel<3>, igemm8_f16_kernel<4>, igemm8_f16_kernel<5>,
                                           igemm8_f16_kernel<6>, igemm8_f16_kernel<7>, igemm8_f16_kernel<8>};


#ifdef MOFA_PROBE
// diagnostic hook of tools/igemm8_probe.py (tools/libmofa_hip_probe.so only; not part of the C ABI): plain-epilogue
// launches run K-loop build variant `var` (see the kernel's VAR) and, for var & 64, write per-segment cycle sums to
// `trace` ([grid][2][16] u64)
static int s_probe_var = 0;
static unsigned long long* s_probe_trace = nullptr;
static const int k_probe_vars[] = {1, 2, 4, 8, 16, 32, 48, 64};
static const igemm8_kern_t k_trace_kern[9] = {nullptr, igemm8_f16_kernel<1, 64>, nullptr, nullptr, igemm8_f16_kernel<4, 64>, nullptr, nullptr,
                                               nullptr, igemm8_f16_kernel<8, 64>};
static const igemm8_kern_t k_probe_kern[] = {
    igemm8_f16_kernel<0, 1>, igemm8_f16_kernel<0, 2>, igemm8_f16_kernel<0, 4>, igemm8_f16_kernel<0, 8>,
    igemm8_f16_kernel<0, 16>, igemm8_f16_kernel<0, 32>, igemm8_f16_kernel<0, 48>, igemm8_f16_kernel<0, 64>};
extern "C" int mofa_igemm8_set_probe(int var, void* trace) {
    s_probe_var = var;
    s_probe_trace = (unsigned long long*)trace;
    return 0;
}
#endif

int igemm8_init() {
    for (igemm8_kern_t k : k_igemm8)
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
#ifdef MOFA_PROBE
    for (igemm8_kern_t k : k_probe_kern)
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
    for (igemm8_kern_t k : k_trace_kern)
        if (k && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
#endif
    return MOFA_OK;
}

int igemm8_launch(const mofa_igemm_args* a, int kind, int n_cu, hipStream_t stream) {
    const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
    if (!igemm_pipe_eligible(a, kind, (long long)taps * a->Cin)) return 1;   // the caller falls back to a 4-wave tile
    const int tilesM = cdiv(a->M, TBM), tilesN = cdiv(a->N, TBN);
    const long long nt = (long long)tilesM * tilesN;
    if (nt > 0x7fffffffLL) return MOFA_EINVAL;
    Aux aux = igemm_pipe_aux(a, taps, tilesN);
    igemm8_kern_t kern = k_igemm8[kind];
#ifdef MOFA_PROBE
    aux.trace = s_probe_trace;
    if (kind == 0 && s_probe_var)
        for (size_t i = 0; i < sizeof(k_probe_vars) / sizeof(int); ++i)
            if (k_probe_vars[i] == s_probe_var) kern = k_probe_kern[i];
    if (s_probe_var == 64 && k_trace_kern[kind]) kern = k_trace_kern[kind];
#endif
    int grid = (int)(nt < n_cu ? ((nt + 7) / 8) * 8 : (n_cu / 8) * 8);
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS_BYTES, stream, *a, tilesN, (int)nt, aux);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
