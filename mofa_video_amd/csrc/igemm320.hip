// Implicit GEMM, 256 (M) x 320 (N) output tile, 8 waves, phase-pipelined K loop: the tile of the UNet's channel counts.
//
// Every N of the SVD UNet / ControlNet trunk is a multiple of 320 (320, 640, 960, 1280, 1920, 2560, 3840, ...): a
// 320-wide tile has NO padded columns on any of them (256-wide tiles waste 37.5 % of their MFMA issue at N = 320, 17 % at
// N = 640), reads the activation tile ONCE where N = 320 (the 256x256 and 192x128 kernels stream it 2 and 3 times), and
// moves 10 % fewer bytes through the LDS-DMA path per flop than the 256x256 tile ((256 + 320) / (256 * 320) rows per
// output).  Same contract (mofa_hip.h), operand roles (MFMA A = weight tile, B = activation tile: an accumulator lane owns
// one output row), buffer-descriptor DMA with bounds-check zero fill, persistent XCD-aware tile walk and two-wave-group
// stagger as igemm8.hip; what differs:
//
//   * wave (wm, wn) of 4 x 2 owns 64 (M) x 160 (N) outputs = 2 x 5 accumulator tiles of 32x32 (160 accumulator registers).
//     A K tile (64 deep) is consumed in TWO PHASES BY K HALVES: phase p multiplies K columns [32 p, 32 p + 32) of all the
//     wave's rows -- 4 X + 10 W fragment reads (56 registers), 20 MFMAs (640 matrix-pipe cycles) -- so both phases carry
//     the same reads and the same MFMA count (the 256x256 kernel splits by X rows: 16 + 8 reads).
//   * LDS: a ring of 2 K tiles, each as two K-HALF PLANES [X 256 rows | W 320 rows] x 64 B (36 KB per plane, 72 KB per K
//     tile, 144 KB ring + 8 x 2 KB wave scratch = the CU's 160 KB).  A plane is released when its phase's fragments are in
//     registers and refilled in the NEXT phase: phase 0 issues plane 1 of K tile t + 1, phase 1 issues plane 0 of K tile
//     t + 2 -- 4 or 5 DMA pieces per wave and phase (9 per K tile; waves 0-3 take the ring's 4 odd W pieces in plane 0,
//     waves 4-7 in plane 1), each with one K tile of flight before the counted wait `vmcnt(9)`.  A DMA piece is 16 rows x
//     64 B (tools/dmabench.hip, profiles/archive/r03_dmabench.log: up to 5 such half-line pieces per phase hide completely behind
//     20 MFMAs of the partner wave; 6 do not).  64-byte rows: the 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3)
//     (applied to the DMA source address and to the fragment read; tools/lds_bank_sim.py: conflict-free).
//   * bias (and the row vector when the wave's 64 rows share one row of it: every tile that does not straddle a frame) is
//     the INITIAL VALUE of the accumulators -- no bias handling in any epilogue.  The bias arrives through LDS one tile
//     ahead (LDS-DMA into a 768-byte slot per wave and tile parity), the uniform row-vector row by global loads.
//   * epilogues: no residual, uniform row vector -> activation in the fragment layout, fp16, 32 x 32 transposes, 16-byte
//     stores; residuals / per-row row vector -> 16 x 32 fp32 transposes (half the lanes write at a time), the sum rounded once.
// Kinds: plain (+ SiLU / ReLU / GELU), row vector, one or two residuals, GEGLU pair (value / gate rows interleaved by 16).
#include "igemm_common.h"
#include "igemm_pipe.h"

namespace {

#ifndef STATS_D
#define STATS_D 5
#endif
constexpr int WN3 = 2, MI3 = 2;                                // wave grid 4 x 2, accumulator row tiles per wave
constexpr int TBM3 = 256;
constexpr int RBH = 64;                                        // bytes of one K half of a row
constexpr int XPL = TBM3 * RBH;                                // X part of a plane (16 KB)
// NJ = accumulator column tiles per wave: 5 -> the 256x320 tile (shipped).  NJ = 4 (a 256x256 tile with the same K-half
// schedule, incl. the GEGLU pair epilogue below) was built and measured in round 3 and is NOT instantiated: it is 5-10 %
// SLOWER than igemm8.hip's X-row split on every shape (profiles/archive/r03_igemm_tiles_bench_with_ksplit256.log: GEGLU L0 / L1 / L2
// 750 / 920 / 1039 against 833 / 1017 / 1141 TF/s) -- the 256x320 tile's advantage is its shape (load segment of 14 reads
// + 4.5 DMA pieces under 640 MFMA cycles; 256x256: 12 + 4 under 512), not the K-half split.
template <int NJ> struct Geo {
    static constexpr int TBN = WN3 * NJ * 32;
    static constexpr int PLANE = (TBM3 + TBN) * RBH;           // 36 KB / 32 KB
    static constexpr int SLOT = 2 * PLANE;                     // one K tile
    static constexpr int BIAS0 = 2 * SLOT;                     // per wave 2 x 768 B: the bias of this tile and of the next one
    static constexpr int LDS_BYTES = BIAS0 + 8 * 2 * 768;     // 159744 (the epilogue transposes through a free ring plane)
    static constexpr int LOOKAHEAD = 4 + NJ;                   // DMA instructions of the last two phases may be in flight
};

struct Cursor3 {                    // one K-half plane of the persistent K-tile stream
    int local;                      // walk position of the output tile it is in
    int ikc, ksw, ky, kx;           // K tile within the tap / overall, tap coordinates (convT3: ky = tap)
    int gx0, gx1;                   // packed row geometry of this wave's two X pieces (16 rows each)
    unsigned xo0, xo1;              // source row of the current tap + this lane's swizzled chunk, bytes from aux.xbase
    int phase;                      // 0: whole tiles of the walk, 1: this workgroup's split-K item, 2: past the end
    int kend;                       // first K tile beyond the current item's K range
    unsigned wo0;                   // weight row of W piece `wave` + this lane's swizzled chunk, bytes from a.w; the
                                    // wave's other pieces are a uniform number of rows further (rows beyond N: out of the
                                    // descriptor's range = zeros, no clamp)
};

// SPLIT: after its whole tiles (tile ids [0, ntiles)) workgroup b < aux.nitems computes ONE split-K item: K slice b % nsplit of
// remainder tile aux.tile0 + b / nsplit (see igemm320_launch).  Its accumulators start at zero and leave as an fp32 partial
// tile in aux.ws; bias, row vector, residuals and activation are applied by igemm320_fixup_kernel.  The item follows the
// whole tiles in the same persistent stream (the cursors prefetch it during the last whole tile's epilogue).
// STATS: the epilogue also emits the GroupNorm pair sums of the outputs (mofa_igemm_args.stats); kinds 0 / 1 / 4 / 5, no
// activation, no per-row vector (igemm320_stats_ok)
template <int EPI, int NJ3, bool SPLIT = false, bool STATS = false>
__global__ __launch_bounds__(512, 2) void igemm320_f16_kernel(const mofa_igemm_args a, const int tilesN, const int ntiles,
                                                              const Aux aux) {
    constexpr bool R1 = (EPI & EPI_R1) != 0, R2 = (EPI & EPI_R2) != 0, RV = (EPI & EPI_RV) != 0, GEGLU = (EPI & EPI_GEGLU) != 0;
    constexpr int TBN3 = Geo<NJ3>::TBN, PLANE = Geo<NJ3>::PLANE, SLOT = Geo<NJ3>::SLOT, LOOKAHEAD3 = Geo<NJ3>::LOOKAHEAD;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY shared object
    TileWalk walk;
    walk.init(ntiles);
    const bool has_item = SPLIT && (int)blockIdx.x < aux.nitems;
    if (walk.local >= walk.count && !has_item) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                     // waves w and w + 4 share a SIMD
    const int wm = wave / WN3, wn = wave - wm * WN3;
    const int l31 = lane & 31, lh = lane >> 5;

    const int taps = igemm_taps(a);
    const int kpt = a.Cin / 64, nk = taps * kpt;
    const size_t Ktot = (size_t)taps * a.Cin;

    // ---- producer side ----------------------------------------------------------------------------------------------
    // plane p of a K tile = 16 X pieces + 20 W pieces of 16 rows x 64 B; wave w issues X pieces w, w + 8, W pieces w, w + 8
    // and, when p == grp, W piece 16 + (w & 3).  Lane l of a piece: row l / 4, slot l % 4 <- source chunk slot ^ ((row >> 2) & 3)
    // = slot ^ ((l >> 4) & 3) (piece rows start at multiples of 16)
    auto lane_now = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto swz_bytes = [&](int l, int p) __attribute__((always_inline)) { return (((l & 3) ^ ((l >> 4) & 3)) * 16 + p * RBH); };
    const int ks_ = a.ksize > 0 ? a.ksize : 3, dil_ = a.dil > 0 ? a.dil : 1;
    const int org_ = a.pad == MOFA_PAD_TRAILING ? 0 : (ks_ >> 1);
    auto pack_geo = [&](int m) __attribute__((always_inline)) -> int {
        int g = m;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int img = fdiv(m, aux.hw), rem = m - img * (a.Hout * a.Wout);
            const int oy = fdiv(rem, aux.wout);
            g = (img << 20) | (oy << 10) | (rem - oy * a.Wout);
        } else if (a.mode == MOFA_MODE_CONVT3) {
            int lo = 1, hi = 1;
            if (a.T > 0) {
                const int fr = fdiv(m, aux.t3hw);
                const int f = fr - fdiv(fr, aux.t3t) * a.T;
                lo = f > 0; hi = f < a.T - 1;
            }
            g = m | (lo << 29) | (hi << 30);
        }
        return m < a.M ? g : -1;
    };
    // W pieces w + 8 and 16 + (w & 3) start 128 and 256 + 16 (w & 3) - 16 w rows after piece w (uniform byte distances)
    const unsigned wd1 = 128u * (unsigned)(Ktot * 2), wd2 = (unsigned)(256 + 16 * (wave & 3) - 16 * wave) * (unsigned)(Ktot * 2);
    constexpr unsigned W_DEAD = 0x80000000u;                       // past the end of the walk: beyond any weight tensor (< 2 GB)
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)aux.xbase, 0, aux.x_bytes, 0x00020000);
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, aux.w_bytes, 0x00020000);
    auto bglds16 = [&](const decltype(rsx)& rs, unsigned voff, int soff, char* lds_wave_base) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
    };
    auto tap_src = [&](int g, int ky, int kx, int swzb) __attribute__((always_inline)) -> unsigned {   // bytes from aux.xbase
        int row = g;
        bool ok = g >= 0;
        if (a.mode == MOFA_MODE_CONV3X3) {
            const int vy = ((g >> 10) & 1023) * a.stride + (ky - org_) * dil_;
            const int vx = (g & 1023) * a.stride + (kx - org_) * dil_;
            ok = ok && vy >= 0 && vx >= 0 && vy < a.Hin * a.up && vx < a.Win * a.up;
            const int iy = (a.up == 2) ? (vy >> 1) : vy, ix = (a.up == 2) ? (vx >> 1) : vx;
            row = ((g >> 20) * a.Hin + iy) * a.Win + ix;
        } else if (a.mode == MOFA_MODE_CONVT3) {
            ok = ok && !(ky == 0 && !((g >> 29) & 1)) && !(ky == 2 && !((g >> 30) & 1));
            row = (g & 0x1fffffff) + (ky - 1) * a.HW + aux.row_shift;
        }
        const unsigned off = (unsigned)row * (unsigned)aux.ldxb + (unsigned)swzb;
        return ok ? off : XO_INVALID;
    };
    // position in the workgroup's stream -> output tile and K-tile range [kb, ke)
    auto item_decode = [&](int phase, int local, int& tile, int& kb, int& ke) __attribute__((always_inline)) {
        tile = walk.start + (phase == 0 ? local : 0); kb = 0; ke = nk;
        if constexpr (SPLIT) {
            const int q = fdiv((int)blockIdx.x, aux.nsplit_d), ks = (int)blockIdx.x - q * aux.nsplit;
            const int kb1 = fdiv(ks * nk, aux.nsplit_d), ke1 = fdiv((ks + 1) * nk, aux.nsplit_d);
            const bool it = phase == 1;
            tile = it ? aux.tile0 + q : tile;
            kb = it ? kb1 : 0;
            ke = it ? ke1 : nk;
        }
    };
    auto next_pos = [&](int& phase, int& local) __attribute__((always_inline)) {
        if (phase == 0) {
            local += walk.stride;
            if (local >= walk.count) phase = has_item ? 1 : 2;
        } else {
            phase = 2;
        }
    };
    auto cur_setup = [&](Cursor3& c, const int p) __attribute__((always_inline)) {
        // branch-free on purpose (selects on the uniform `live`; see igemm8.hip)
        const bool live = c.phase < 2;
        int tile, kb, ke;
        item_decode(c.phase, c.local, tile, kb, ke);
        const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
        const int l = lane_now(), r_l = l >> 2;
        const int g0 = pack_geo(tm * TBM3 + 16 * wave + r_l), g1 = pack_geo(tm * TBM3 + 16 * (wave + 8) + r_l);
        c.gx0 = live ? g0 : -1;
        c.gx1 = live ? g1 : -1;
        const unsigned wo = (unsigned)(tn * TBN3 + 16 * wave + r_l) * (unsigned)(Ktot * 2) + swz_bytes(l, p);
        c.wo0 = live ? wo : W_DEAD;
        c.ikc = 0; c.ksw = 0; c.ky = 0; c.kx = 0;
        c.kend = ke;
        if constexpr (SPLIT) {                                     // a slice may start in the middle of a tap
            int tap = fdiv(kb, aux.kpt_d);
            c.ikc = kb - tap * kpt;
            c.ksw = kb;
            if (a.mode == MOFA_MODE_CONV3X3) { c.ky = fdiv(tap, aux.ks_d); c.kx = tap - c.ky * ks_; } else c.ky = tap;
            c.xo0 = tap_src(c.gx0, c.ky, c.kx, swz_bytes(l, p));
            c.xo1 = tap_src(c.gx1, c.ky, c.kx, swz_bytes(l, p));
        }
    };
    auto issue = [&](Cursor3& c, const int p, const int slot) __attribute__((always_inline)) {
        if (c.ikc == 0) {
            const int l = lane_now();
            c.xo0 = tap_src(c.gx0, c.ky, c.kx, swz_bytes(l, p));
            c.xo1 = tap_src(c.gx1, c.ky, c.kx, swz_bytes(l, p));
        }
        char* pl = smem + slot + p * PLANE;
        const int wk = c.ksw * 128;
        bglds16(rsx, c.xo0, c.ikc * 128, pl + wave * 1024);
        bglds16(rsx, c.xo1, c.ikc * 128, pl + (wave + 8) * 1024);
        bglds16(rsw, c.wo0, wk, pl + XPL + wave * 1024);
        bglds16(rsw, c.wo0 + wd1, wk, pl + XPL + (wave + 8) * 1024);
        if (NJ3 == 5 && grp == p) bglds16(rsw, c.wo0 + wd2, wk, pl + XPL + (16 + (wave & 3)) * 1024);
    };
    auto advance = [&](Cursor3& c, const int p) __attribute__((always_inline)) {
        ++c.ksw;
        if (++c.ikc == kpt) {
            c.ikc = 0;
            if (a.mode == MOFA_MODE_CONV3X3) { if (++c.kx == ks_) { c.kx = 0; ++c.ky; } } else ++c.ky;
        }
        if (c.ksw == c.kend) { next_pos(c.phase, c.local); cur_setup(c, p); }
    };

    // ---- consumer side ----------------------------------------------------------------------------------------------
    // fragment read addresses of ring slot 0, plane 0 (one register per 16-deep K step of a plane; accumulator tile and
    // plane are immediates, the ring slot is added / subtracted per K tile)
    // (the second 16-deep K step of a plane is the chunk with bit 1 flipped: address ^ 32)
    int xa, wa;
    {
        const int sl = (lh ^ ((l31 >> 2) & 3)) * 16;
        xa = (wm * MI3 * 32 + l31) * RBH + sl;
        wa = XPL + (wn * NJ3 * 32 + l31) * RBH + sl;
    }

    // The wave's 160 bias values travel through LDS one tile AHEAD: three 4-byte-per-lane LDS-DMA instructions issued at the
    // start of tile t bring tile t + 1's bias (columns beyond N read as zero through the descriptor's bounds check; no bias: an
    // empty descriptor, zeros arrive), landed long before tile t + 1 starts (every phase's counted wait covers them), so the
    // accumulator initialisation is 20 LDS reads with no VMEM operation queued behind the previous epilogue's stores.  (The
    // first version loaded the bias straight from global memory at tile start; tools/bias_probe.py shows no measurable
    // difference between the two, nor between bias and no bias -- its first reading, "bias costs 15 %", was the slower first
    // measurement after fresh allocations.)
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? (unsigned)a.N * 4u : 0u, 0x00020000);
    char* bias_lds = smem + Geo<NJ3>::BIAS0 + wave * 1536;
    auto bias_prefetch = [&](int phase_, int local_, int slot) __attribute__((always_inline)) {
        if (phase_ != 0) return;                                   // (a split-K item starts from zero; past the end: nothing)
        int tile_, kb_, ke_;
        item_decode(phase_, local_, tile_, kb_, ke_);
        const int tm_ = fdiv(tile_, aux.tiles_n), tn_ = tile_ - tm_ * tilesN;
        const unsigned n0 = (unsigned)(tn_ * TBN3 + wn * NJ3 * 32 + lane_now());
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (__attribute__((address_space(3))) void*)(bias_lds + slot * 768 + k * 256), 4,
                                                     (n0 + 64u * k) * 4u, 0, 0, 0);
    };

    Cursor3 ca, cb;                                                // plane 0 stream, plane 1 stream
    ca.local = cb.local = walk.local;
    ca.phase = cb.phase = walk.local < walk.count ? 0 : 1;     // (no whole tile: the workgroup was kept for its split-K item)
    cur_setup(ca, 0);
    cur_setup(cb, 1);
    // prologue: the first tile's bias, K tile 0 complete, plane 0 of K tile 1
    bias_prefetch(walk.local < walk.count ? 0 : 1, walk.local, 0);
    issue(ca, 0, 0); advance(ca, 0);
    issue(cb, 1, 0); advance(cb, 1);
    issue(ca, 0, SLOT); advance(ca, 0);
    wait_vmcnt_only<LOOKAHEAD3>();                                 // plane 0 of K tile 0 has landed
    __builtin_amdgcn_s_barrier();

    f32x16 acc[MI3][NJ3];
    f16x8 xf[MI3][2], wf[NJ3][2];
    auto phase = [&](auto pc, const int bo, const int bn) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::v;                         // K half
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int xk = k2 ? (xa ^ 32) : xa, wk = k2 ? (wa ^ 32) : wa;
#pragma unroll
            for (int i = 0; i < MI3; ++i) xf[i][k2] = *(const f16x8*)(smem + xk + P * PLANE + i * 32 * RBH);
#pragma unroll
            for (int j = 0; j < NJ3; ++j) wf[j][k2] = *(const f16x8*)(smem + wk + P * PLANE + j * 32 * RBH);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P == 0) { issue(cb, 1, bn); advance(cb, 1); }   // plane 1 of the next K tile
        else { issue(ca, 0, bo); advance(ca, 0); }                    // plane 0 of the K tile after it (this slot)
        // DMA older than the last two phases has landed (read from the next phase on); this phase's fragment reads have
        // returned BEFORE the barrier, so their plane may be refilled from the next phase on
        wait_vmcnt<LOOKAHEAD3>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int i = 0; i < MI3; ++i)
#pragma unroll
                for (int j = 0; j < NJ3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j][k2], xf[i][k2], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- epilogue scratch: the ring plane that is FREE while an epilogue runs ---------------------------------------------
    // When a tile's K loop ends, the ring holds the next tile's K tile 0 (other slot, both planes) and plane 0 of its K tile 1
    // (this slot); plane 1 of the last K tile's slot was read for the last time in the final phase and is refilled only by
    // phase 0 of the NEXT tile -- by every wave into its own pieces.  Each wave therefore transposes through the four 1 KB
    // blocks it will DMA into itself (X pieces w, w + 8 and W pieces w, w + 8 of that plane: 8 KB apart): 32 rows x 128 B,
    // rows 8 b .. 8 b + 7 in block b.  No other wave touches those blocks before its fragment reads two barriers later, the
    // wave's own DMA is issued after its epilogue (program order, lgkmcnt(0) at the end), and no barrier is needed.
    auto srow = [&](int r) __attribute__((always_inline)) { return (r >> 3) * 8192 + (r & 7) * 128; };
    auto act_apply = [&](auto ac, float v) __attribute__((always_inline)) -> float {
        constexpr int ACT = decltype(ac)::v;
        if constexpr (ACT == MOFA_ACT_SILU) return silu_f(v);
        else if constexpr (ACT == MOFA_ACT_RELU) return fmaxf(v, 0.0f);
        else if constexpr (ACT == MOFA_ACT_GELU) return gelu_erf_f(v);
        else return v;
    };
    // register r of accumulator tile (i, j) is row 32 i + l31, column 32 j + 8 (r >> 2) + 4 lh + (r & 3)
    // ---- no residual, no per-row vector: activation in the fragment layout, fp16; two accumulator tiles (64 columns) per
    //      transpose so that a store instruction covers 8 whole 128-byte rows.  The wave's 160 columns start at byte 0 (wn = 0)
    //      or 320 (wn = 1) of a 640-byte-aligned row: the pairs are (0,1) (2,3) + tile 4, or tile 0 + (1,2) (3,4), so that
    //      every pair starts on a 128-byte boundary -----------------------------------------------------------------------------
    auto epilogue_light = [&](auto ac, auto wnc, f32x16 (&acc)[MI3][NJ3], const int mw, const int nw, char* eb) __attribute__((always_inline)) {
        constexpr int J0 = (NJ3 == 5 && decltype(wnc)::v) ? 1 : 0;   // first tile of the first pair
        constexpr int JS = decltype(wnc)::v ? 0 : 4;                 // NJ = 5: the tile without a partner
        const int lane_e = lane_now();                             // (lane-derived offsets are not kept live across the K loop)
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        float saccv = a.s_acc;                                     // VGPR operand on purpose (see igemm.hip's epilogue)
        asm volatile("" : "+v"(saccv));
        char* wr = eb + srow(l31);
        const int wsw = (l31 >> 1) & 7, wpar = l31 & 1;
#pragma unroll
        for (int i = 0; i < MI3; ++i) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                       // the two pairs
                const int j0 = J0 + 2 * pr;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (f16)act_apply(ac, saccv * acc[i][j0 + jj][4 * g + e]);
                        const int c8 = 8 * jj + 2 * g + lh;
                        *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), blk = lane & 7;
                    const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    const int mr = mw + 32 * i + row, n = nw + 32 * j0 + 8 * blk;
                    if (mr < a.M && n + 8 <= a.N) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
                }
            }
            if constexpr (NJ3 == 5) {                              // the single tile: 32 columns per row
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)act_apply(ac, saccv * acc[i][JS][4 * g + e]);
                    const int c8 = 2 * g + lh;
                    *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int row = 16 * p + (lane >> 2), blk = lane & 3;
                    const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    const int mr = mw + 32 * i + row, n = nw + 32 * JS + 8 * blk;
                    if (mr < a.M && n + 8 <= a.N) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
                }
            }
        }
        wait_lds();
    };
    // ---- residual(s) without a per-row vector (r04): the light epilogue's fp16 pair transposes -- whole 128-byte lines per store,
    //      half the LDS traffic of the fp32 32 x 32 transposes of epilogue_rows below -- and the residual(s) added in the ROW
    //      layout afterwards: a lane holds 8 consecutive columns of one row, i.e. exactly one 16-byte residual load.  The
    //      accumulator is rounded to fp16 BEFORE the add, which is what the reference's fp16 modules do (the layer's output in
    //      fp16, then `+`); with one residual and s1 == 1 (every residual add of the networks except the AlphaBlender mixes) the
    //      add is 4 v_pk_add_f16 per 8 outputs.  A ring of D residual pieces is in flight ahead of the transposes. --------------
    auto epilogue_res16 = [&](auto wnc, auto unit, f32x16 (&acc)[MI3][NJ3], const int mw, const int nw, char* eb) __attribute__((always_inline)) {
        constexpr int J0 = (NJ3 == 5 && decltype(wnc)::v) ? 1 : 0;   // first tile of the first pair
        constexpr int JS = decltype(wnc)::v ? 0 : 4;                 // NJ = 5: the tile without a partner
        constexpr bool UNIT = decltype(unit)::v != 0;                // one residual, s1 == 1: packed fp16 adds
        constexpr int PPI = NJ3 == 5 ? 10 : 8, NP = MI3 * PPI;       // store pieces (16 bytes per lane) per row tile / per tile
        const int lane_e = lane_now();
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        const f16* r1 = (const f16*)a.r1;
        const f16* r2 = (const f16*)a.r2;
        float s1v = a.s1, s2v = a.s2, saccv = a.s_acc;
        asm volatile("" : "+v"(s1v), "+v"(s2v), "+v"(saccv));
        // piece q of row tile i: q < 8 -> pair q / 4, rows 8 (q % 4) + lane / 8, columns 8 (lane % 8) of the pair's 64;
        //                       q >= 8 -> the single tile, rows 16 (q - 8) + lane / 4, columns 8 (lane % 4) of its 32
        auto piece_pos = [&](int st, int& mr, int& n) __attribute__((always_inline)) {
            const int i = st / PPI, q = st % PPI;
            if (q < 8) {
                mr = mw + 32 * i + 8 * (q & 3) + (lane >> 3);
                n = nw + 32 * (J0 + 2 * (q >> 2)) + 8 * (lane & 7);
            } else {
                mr = mw + 32 * i + 16 * (q - 8) + (lane >> 2);
                n = nw + 32 * JS + 8 * (lane & 3);
            }
        };
        constexpr int D = (R1 && R2) ? 2 : 5;
        f16x8 L1[D], L2[D];
        auto piece_loads = [&](int st, int slot) __attribute__((always_inline)) {
            int mr, n;
            piece_pos(st, mr, n);
            mr = mr < a.M ? mr : a.M - 1;
            n = n + 8 <= a.N ? n : 0;
            if (R1) L1[slot] = *(const f16x8*)(r1 + (size_t)mr * a.ldr1 + n);
            if (R2) L2[slot] = *(const f16x8*)(r2 + (size_t)mr * a.ldr2 + n);
        };
#pragma unroll
        for (int st = 0; st < D; ++st) piece_loads(st, st);
        auto finish = [&](int st, f16x8 o) __attribute__((always_inline)) {
            int mr, n;
            piece_pos(st, mr, n);
            if constexpr (UNIT) {
                o = o + L1[st % D];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = (float)o[e];
                    if (R1) x += s1v * (float)L1[st % D][e];
                    if (R2) x += s2v * (float)L2[st % D][e];
                    o[e] = (f16)x;
                }
            }
            if (mr < a.M && n + 8 <= a.N) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
            if (st + D < NP) piece_loads(st + D, st % D);           // refill the ring slot just consumed
        };
        char* wr = eb + srow(l31);
        const int wsw = (l31 >> 1) & 7, wpar = l31 & 1;
#pragma unroll
        for (int i = 0; i < MI3; ++i) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                       // the two pairs
                const int j0 = J0 + 2 * pr;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (f16)(saccv * acc[i][j0 + jj][4 * g + e]);
                        const int c8 = 8 * jj + 2 * g + lh;
                        *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), blk = lane & 7;
                    const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    finish(i * PPI + pr * 4 + p, o);
                }
            }
            if constexpr (NJ3 == 5) {                              // the single tile: 32 columns per row
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (f16)(saccv * acc[i][JS][4 * g + e]);
                    const int c8 = 2 * g + lh;
                    *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int row = 16 * p + (lane >> 2), blk = lane & 3;
                    const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    finish(i * PPI + 8 + p, o);
                }
            }
        }
        wait_lds();
    };
    // ---- STATS kernels: epilogue_res16 (or its residual-free form) with the store pieces walked COLUMN SET by column set -- the
    //      tile pairs (J0, J0 + 1), (J0 + 2, J0 + 3), then the single tile, each over both row tiles -- so that only one set's
    //      accumulators (4 column pairs x {sum, sum of squares}) are live at a time (a residual is added in fp16: s1 == 1 only).  A finished piece is 8 consecutive fp16
    //      outputs of one row: 2 x 4 v_dot2_f32_f16 add the pairs and their squares; after a set, xor-shuffles over the lanes that
    //      hold the same columns (8 rows each -> the wave's 64 rows), and lanes 0-7 (0-3) write 32 bytes each:
    //      stats[(m / 64) * N + n .. n + 7] = {s, q} x 4 pairs.  Fixed order: bit-identical run to run. -------------------------
    auto epilogue_stats = [&](auto wnc, f32x16 (&acc)[MI3][NJ3], const int mw, const int nw, char* eb) __attribute__((always_inline)) {
        constexpr int J0 = decltype(wnc)::v ? 1 : 0, JS = decltype(wnc)::v ? 0 : 4;
        constexpr int NP = 20;                                       // pieces: 2 sets x 2 row tiles x 4, then 2 row tiles x 2
        const int lane_e = lane_now();
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        const f16* r1 = (const f16*)a.r1;
        float saccv = a.s_acc;
        asm volatile("" : "+v"(saccv));
        auto piece_pos = [&](int st, int& mr, int& n) __attribute__((always_inline)) {
            if (st < 16) {                                           // set st / 8, row tile (st / 4) % 2, piece st % 4
                mr = mw + 32 * ((st >> 2) & 1) + 8 * (st & 3) + (lane >> 3);
                n = nw + 32 * (J0 + 2 * (st >> 3)) + 8 * (lane & 7);
            } else {                                                 // the single tile: row tile (st - 16) / 2, piece (st - 16) % 2
                mr = mw + 32 * ((st - 16) >> 1) + 16 * ((st - 16) & 1) + (lane >> 2);
                n = nw + 32 * JS + 8 * (lane & 3);
            }
        };
        constexpr int D = STATS_D;
        f16x8 L1[D];
        auto piece_loads = [&](int st, int slot) __attribute__((always_inline)) {
            int mr, n;
            piece_pos(st, mr, n);
            mr = mr < a.M ? mr : a.M - 1;
            L1[slot] = *(const f16x8*)(r1 + (size_t)mr * a.ldr1 + n);
        };
        if constexpr (R1) {
#pragma unroll
            for (int st = 0; st < D; ++st) piece_loads(st, st);
        }
        float S[4], Q[4];
        const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
        auto finish = [&](int st, f16x8 o) __attribute__((always_inline)) {
            int mr, n;
            piece_pos(st, mr, n);
            if constexpr (R1) o = o + L1[st % D];                   // (s1 == 1: the residual add in fp16, as the reference's modules)
            if (mr < a.M) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f16x2 h = {o[2 * k], o[2 * k + 1]};
                S[k] = __builtin_amdgcn_fdot2(h, one2, S[k], false);
                Q[k] = __builtin_amdgcn_fdot2(h, h, Q[k], false);
            }
            // (pins the eight v_dot2 HERE: left alone the compiler sinks them to the end of the column set and keeps all eight
            // pieces' outputs -- 32 registers -- live until then, which spills)
            asm volatile("" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(Q[0]), "+v"(Q[1]), "+v"(Q[2]), "+v"(Q[3]));
            if constexpr (R1) {
                if (st + D < NP) piece_loads(st + D, st % D);       // refill the ring slot just consumed
            }
        };
        auto reduce_store = [&](const int from, const int n0, const bool writer) __attribute__((always_inline)) {
            // (ds_bpermute addressed from the laundered lane id: __shfl_xor's own lane id is loop invariant, gets hoisted out of the
            // tile loop and then lives -- spilled -- across the K loop)
#pragma unroll
            for (int o = from; o < 64; o <<= 1) {
                const int src = (lane ^ o) << 2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    S[k] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, S[k])));
                    Q[k] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, Q[k])));
                }
            }
            if (writer && mw < a.M) {                                // (M % 64 == 0: the wave's 64 rows are all inside or all outside)
                float* sp = a.stats + (size_t)(mw >> 6) * a.N + n0;
                *(f32x4*)sp = (f32x4){S[0], Q[0], S[1], Q[1]};
                *(f32x4*)(sp + 4) = (f32x4){S[2], Q[2], S[3], Q[3]};
            }
        };
        char* wr = eb + srow(l31);
        const int wsw = (l31 >> 1) & 7, wpar = l31 & 1;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {                           // the two pairs
            const int j0 = J0 + 2 * pr;
#pragma unroll
            for (int k = 0; k < 4; ++k) { S[k] = 0.f; Q[k] = 0.f; }
#pragma unroll
            for (int i = 0; i < MI3; ++i) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (f16)(saccv * acc[i][j0 + jj][4 * g + e]);
                        const int c8 = 8 * jj + 2 * g + lh;
                        *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int row = 8 * p + (lane >> 3), blk = lane & 7;
                    const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                    f16x8 o = v;
                    if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                    finish(pr * 8 + i * 4 + p, o);
                }
            }
            reduce_store(8, nw + 32 * j0 + 8 * (lane & 7), lane < 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { S[k] = 0.f; Q[k] = 0.f; }
#pragma unroll
        for (int i = 0; i < MI3; ++i) {                            // the single tile: 32 columns per row
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)(saccv * acc[i][JS][4 * g + e]);
                const int c8 = 2 * g + lh;
                *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = o;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = 16 * p + (lane >> 2), blk = lane & 3;
                const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                f16x8 o = v;
                if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                finish(16 + i * 2 + p, o);
            }
        }
        reduce_store(4, nw + 32 * JS + 8 * (lane & 3), lane < 4);
        wait_lds();
    };
    // ---- GEGLU pair kind: the weight rows are interleaved in blocks of 16 (weights.interleave_geglu), so accumulator tile j
    //      holds the value columns of outputs 16 j .. 16 j + 15 in registers 0-7 and the matching gate columns in registers
    //      8-15 of the SAME lane: the product is lane-local and a wave's NJ tiles give 16 NJ output columns (80 of the tile's
    //      160).  out = s_acc * val * gelu(s_acc * gate), erf GELU as x * Phi(x) (common.h).  Tiles 0-3 (64 output columns)
    //      go through one 32 x 64 fp16 transpose and whole 128-byte row pieces, tile 4 (16 columns) through a second one ------
    auto epilogue_geglu = [&](auto s1c, f32x16 (&acc)[MI3][NJ3], const int mw, const int nw, char* eb) __attribute__((always_inline)) {
        constexpr bool SACC1 = decltype(s1c)::v != 0;              // s_acc == 1: the multiplies drop out
        const int lane_e = lane_now();
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        float saccv = a.s_acc;
        asm volatile("" : "+v"(saccv));
        const int nout = a.N / 2, no0 = nw / 2;                    // first output column of this wave
        char* wr = eb + srow(l31);
        const int wsw = (l31 >> 1) & 7, wpar = l31 & 1;
        auto product = [&](int i, int j, int g) __attribute__((always_inline)) -> f16x4 {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float val = acc[i][j][4 * g + e], gate = acc[i][j][8 + 4 * g + e];
                if constexpr (SACC1) o[e] = (f16)(val * (gate * gelu_phi_f(gate)));
                else o[e] = (f16)(saccv * val * gelu_erf_f(saccv * gate));
            }
            return o;
        };
#pragma unroll
        for (int i = 0; i < MI3; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c8 = 4 * j + 2 * g + lh;             // 8-byte chunk (4 columns) of the 64-column row
                    *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = product(i, j, g);
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = 8 * p + (lane >> 3), blk = lane & 7;
                const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                f16x8 o = v;
                if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                const int mr = mw + 32 * i + row, n = no0 + 8 * blk;
                if (mr < a.M && n + 8 <= nout) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
            }
            if constexpr (NJ3 == 5) {                              // tile 4: 16 columns per row
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c8 = 2 * g + lh;
                    *(f16x4*)(wr + (((c8 >> 1) ^ wsw) << 4) + (((c8 & 1) ^ wpar) << 3)) = product(i, 4, g);
                }
                const int row = lane >> 1, blk = lane & 1;
                const f16x8 v = *(const f16x8*)(eb + srow(row) + ((blk ^ ((row >> 1) & 7)) << 4));
                f16x8 o = v;
                if (row & 1) o = (f16x8){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
                const int mr = mw + 32 * i + row, n = no0 + 64 + 8 * blk;
                if (mr < a.M && n + 8 <= nout) *(f16x8*)(out + (size_t)mr * a.ldo + n) = o;
            }
        }
        wait_lds();
    };
    // ---- residuals and / or a per-row vector: 32 x 32 fp32 transposes; a lane owns 8 consecutive
    //      columns of one row.  The loads of step s + D are issued before step s is processed (ring of D steps in registers) ----
    auto epilogue_rows = [&](auto un, f32x16 (&acc)[MI3][NJ3], const int mw, const int nw, char* eb) __attribute__((always_inline)) {
        constexpr bool RVROW = RV && decltype(un)::v == 0;         // the row vector was NOT folded into the accumulators
        constexpr int STEPS = MI3 * NJ3;                           // (i, j): one accumulator tile per step
        const int lane_e = lane_now();
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        f16* out = (f16*)a.out;
        const f16* r1 = (const f16*)a.r1;
        const f16* r2 = (const f16*)a.r2;
        float s1v = a.s1, s2v = a.s2, saccv = a.s_acc;
        asm volatile("" : "+v"(s1v), "+v"(s2v), "+v"(saccv));
        int rv_div = a.rv_div, rv_mod_in = a.rv_mod_in, rv_mod_out = a.rv_mod_out;
        asm volatile("" : "+s"(rv_div), "+s"(rv_mod_in), "+s"(rv_mod_out));
        const int piece = lane & 3, rrow = lane >> 2;
        // ring of D steps of loads in registers (8 VGPRs per residual, 16 for a per-row vector): the epilogue of a shallow-K
        // launch is bound by the latency of exactly these loads
        constexpr int REGS = (R1 ? 8 : 0) + (R2 ? 8 : 0) + (RVROW ? 16 : 0);
        constexpr int D = RVROW ? 1 : (REGS <= 8 ? 6 : 3);       // (the per-row vector path is rare: tiles that straddle a frame)
        struct StepLoads { f16x8 t1[2], t2[2]; f32x4 rv0[2], rv1[2]; };
        StepLoads L[D];
        auto step_loads = [&](int st, StepLoads& l) __attribute__((always_inline)) {
            const int i = st / NJ3, j = st % NJ3;
            const int n = nw + 32 * j + 8 * piece;
            const int nc = n + 8 <= a.N ? n : 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                int m = mw + 32 * i + 16 * p + rrow;
                m = m < a.M ? m : a.M - 1;
                if (R1) l.t1[p] = *(const f16x8*)(r1 + (size_t)m * a.ldr1 + nc);
                if (R2) l.t2[p] = *(const f16x8*)(r2 + (size_t)m * a.ldr2 + nc);
                if (RVROW) {
                    const int idx = ((m / rv_div) * a.rv_mul + (m % rv_mod_in)) % rv_mod_out;
                    const float* q = a.rowvec + (size_t)idx * a.N + nc;
                    l.rv0[p] = *(const f32x4*)q;
                    l.rv1[p] = *(const f32x4*)(q + 4);
                }
            }
        };
#pragma unroll
        for (int st = 0; st < D; ++st) step_loads(st, L[st]);
        char* wr = eb + srow(l31);
        const int wsw = ((l31 >> 1) & 3) | ((l31 & 1) << 2);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int i = st / NJ3, j = st % NJ3;
            StepLoads& l = L[st % D];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                *(f32x4*)(wr + (((2 * g + lh) ^ wsw) << 4)) = v;
            }
            const int n = nw + 32 * j + 8 * piece;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = 16 * p + rrow;
                const int rsw = ((row >> 1) & 3) | ((row & 1) << 2);
                const f32x4 v0 = *(const f32x4*)(eb + srow(row) + (((2 * piece) ^ rsw) << 4));
                const f32x4 v1 = *(const f32x4*)(eb + srow(row) + (((2 * piece + 1) ^ rsw) << 4));
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = v0[e], x1 = v1[e];
                    if (RVROW) { x0 += l.rv0[p][e]; x1 += l.rv1[p][e]; }
                    x0 *= saccv; x1 *= saccv;
                    // (with residuals the layer's own output is rounded to fp16 BEFORE the add, like epilogue_res16 and the
                    //  reference's fp16 modules: every tile of a launch rounds the same way -- round-4 advice)
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
        wait_lds();
    };

    // ---- split-K: the fp32 partial tile, row-major [256][TBN], through the same 32 x 32 fp32 transposes ------------------------
    auto epilogue_dump = [&](f32x16 (&acc)[MI3][NJ3], float* wst, char* eb) __attribute__((always_inline)) {
        const int lane_e = lane_now();
        const int lane = lane_e, l31 = lane & 31, lh = lane >> 5;
        const int piece = lane & 3, rrow = lane >> 2;
        char* wr = eb + srow(l31);
        const int wsw = ((l31 >> 1) & 3) | ((l31 & 1) << 2);
#pragma unroll
        for (int st = 0; st < MI3 * NJ3; ++st) {
            const int i = st / NJ3, j = st % NJ3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                *(f32x4*)(wr + (((2 * g + lh) ^ wsw) << 4)) = v;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int row = 16 * p + rrow;
                const int rsw = ((row >> 1) & 3) | ((row & 1) << 2);
                const f32x4 v0 = *(const f32x4*)(eb + srow(row) + (((2 * piece) ^ rsw) << 4));
                const f32x4 v1 = *(const f32x4*)(eb + srow(row) + (((2 * piece + 1) ^ rsw) << 4));
                float* d = wst + (size_t)(wm * MI3 * 32 + 32 * i + row) * TBN3 + wn * NJ3 * 32 + 32 * j + 8 * piece;
                *(f32x4*)d = v0;
                *(f32x4*)(d + 4) = v1;
            }
        }
        wait_lds();
    };

    int gt = 0;                                                    // K tiles consumed so far (ring slot = gt & 1)
    int cph = walk.local < walk.count ? 0 : 1, cl = walk.local, bslot = 0;
    for (; cph < 2; next_pos(cph, cl), bslot ^= 1) {
        int tile, kb, ke;
        item_decode(cph, cl, tile, kb, ke);
        {                                                          // the NEXT tile's bias into the other slot
            int nph = cph, nl = cl;
            next_pos(nph, nl);
            bias_prefetch(nph, nl, bslot ^ 1);
        }
        const int tm = fdiv(tile, aux.tiles_n), tn = tile - tm * tilesN;
        const int mw = tm * TBM3 + wm * MI3 * 32;                  // first output row / column of this wave
        const int nw = tn * TBN3 + wn * NJ3 * 32;
        int idx_u = -1;                                            // >= 0: the one row-vector row of this wave's 64 rows
        if constexpr (RV) {
            if (a.rv_mod_in == 1) {                                // idx(m) = ((m / div) * mul) % mod_out: a step function
                int rv_div = a.rv_div;
                asm volatile("" : "+s"(rv_div));
                const int m0 = mw < a.M ? mw : a.M - 1, m1 = mw + 32 * MI3 - 1 < a.M ? mw + 32 * MI3 - 1 : a.M - 1;
                const int q0 = m0 / rv_div, q1 = m1 / rv_div;
                if (q0 == q1) idx_u = (q0 * a.rv_mul) % a.rv_mod_out;
            }
            idx_u = __builtin_amdgcn_readfirstlane(idx_u);
        }
        // accumulators start at bias (from its LDS slot) + the wave's row of the row vector
        if (SPLIT && cph == 1) {
#pragma unroll
            for (int i = 0; i < MI3; ++i)
#pragma unroll
                for (int j = 0; j < NJ3; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        } else {
            const int lane_e = lane_now();
            const int lh_e = lane_e >> 5;
#pragma unroll
            for (int j = 0; j < NJ3; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 b = *(const f32x4*)(bias_lds + bslot * 768 + (32 * j + 8 * g + 4 * lh_e) * 4);
                    if (RV && idx_u >= 0) {
                        int n = nw + 32 * j + 8 * g + 4 * lh_e;
                        n = n + 4 <= a.N ? n : 0;                   // columns beyond N are never stored
                        b += *(const f32x4*)(a.rowvec + (size_t)idx_u * a.N + n);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[0][j][4 * g + e] = b[e]; acc[1][j][4 * g + e] = b[e]; }
                }
        }
        if (grp == 1) __builtin_amdgcn_s_barrier();                // second wave group runs one barrier behind
        for (int kt = kb; kt < ke; ++kt, ++gt) {
            const int bo = (gt & 1) * SLOT, bn = SLOT - bo;        // ring slot of this K tile / of the next one
            phase(IC<0>{}, bo, bn);
            phase(IC<1>{}, bo, bn);
            const int d = (gt & 1) ? -SLOT : SLOT;                 // the next K tile's ring slot
            xa += d;
            wa += d;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();                // both groups meet again: equal barrier counts per tile

        // ---- epilogue (no barriers; scratch = this wave's blocks of the free ring plane) --------------------------------
        char* eb = smem + ((gt - 1) & 1) * SLOT + PLANE + wave * 1024;
        if (SPLIT && cph == 1) {
            epilogue_dump(acc, aux.ws + (size_t)blockIdx.x * (TBM3 * TBN3), eb);
        } else if constexpr (STATS) {                              // (kinds 0 / 1 / 4 / 5; the launcher excluded a per-row vector and s1 != 1)
            if (wn == 0) epilogue_stats(IC<0>{}, acc, mw, nw, eb);
            else epilogue_stats(IC<1>{}, acc, mw, nw, eb);
        } else if constexpr (GEGLU) {
            if (a.s_acc == 1.0f) epilogue_geglu(IC<1>{}, acc, mw, nw, eb);
            else epilogue_geglu(IC<0>{}, acc, mw, nw, eb);
        } else if constexpr (R1 || R2) {
            if (RV && idx_u < 0) {
                epilogue_rows(IC<0>{}, acc, mw, nw, eb);           // a per-row vector (tile straddles a frame): fp32 row path
            } else {                                               // (a uniform row-vector row is already in the accumulators)
                const bool unit = R1 && !R2 && a.s1 == 1.0f;
                if (NJ3 == 4 || wn == 0) {
                    if (unit) epilogue_res16(IC<0>{}, IC<1>{}, acc, mw, nw, eb);
                    else epilogue_res16(IC<0>{}, IC<0>{}, acc, mw, nw, eb);
                } else {
                    if (unit) epilogue_res16(IC<1>{}, IC<1>{}, acc, mw, nw, eb);
                    else epilogue_res16(IC<1>{}, IC<0>{}, acc, mw, nw, eb);
                }
            }
        } else if (RV && idx_u < 0) {
            epilogue_rows(IC<0>{}, acc, mw, nw, eb);
        } else if (NJ3 == 4 || wn == 0) {
            if (RV || a.act == MOFA_ACT_NONE) epilogue_light(IC<MOFA_ACT_NONE>{}, IC<0>{}, acc, mw, nw, eb);
            else if (a.act == MOFA_ACT_SILU) epilogue_light(IC<MOFA_ACT_SILU>{}, IC<0>{}, acc, mw, nw, eb);
            else if (a.act == MOFA_ACT_RELU) epilogue_light(IC<MOFA_ACT_RELU>{}, IC<0>{}, acc, mw, nw, eb);
            else epilogue_light(IC<MOFA_ACT_GELU>{}, IC<0>{}, acc, mw, nw, eb);
        } else {
            if (RV || a.act == MOFA_ACT_NONE) epilogue_light(IC<MOFA_ACT_NONE>{}, IC<1>{}, acc, mw, nw, eb);
            else if (a.act == MOFA_ACT_SILU) epilogue_light(IC<MOFA_ACT_SILU>{}, IC<1>{}, acc, mw, nw, eb);
            else if (a.act == MOFA_ACT_RELU) epilogue_light(IC<MOFA_ACT_RELU>{}, IC<1>{}, acc, mw, nw, eb);
            else epilogue_light(IC<MOFA_ACT_GELU>{}, IC<1>{}, acc, mw, nw, eb);
        }
    }
    // a wave must not retire with an LDS DMA in flight (the cursors' run-ahead past the last tile)
    wait_vmcnt_only<0>();
}

typedef void (*igemm320_kern_t)(const mofa_igemm_args, const int, const int, const Aux);

// split-K fix-up: out = act(s_acc * (sum over the K slices + bias + rowvec[idx(m)]) + s1 r1 + s2 r2) for the R remainder tiles,
// one thread per 8 output columns of a row; the slices are added in slice order (deterministic)
__global__ __launch_bounds__(256) void igemm320_fixup_kernel(const mofa_igemm_args a, const float* __restrict__ ws, const int tile0,
                                                             const int nsplit, const int tilesN, const int R) {
    constexpr int TBN = Geo<5>::TBN, PPR = TBN / 8, PPT = TBM3 * PPR;   // pieces per row / per tile
    const long long total = (long long)R * PPT;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / PPT), rem = (int)(idx - (long long)r * PPT);
        const int rr = rem / PPR, pc = rem - rr * PPR;
        const int tile = tile0 + r, tm = tile / tilesN, tn = tile - tm * tilesN;
        const int m = tm * TBM3 + rr, n = tn * TBN + pc * 8;
        if (m >= a.M || n + 8 > a.N) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int sl = 0; sl < nsplit; ++sl) {
            const float* q = ws + ((size_t)(r * nsplit + sl) * TBM3 + rr) * TBN + pc * 8;
            const f32x4 p0 = *(const f32x4*)q, p1 = *(const f32x4*)(q + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += p0[e]; v[4 + e] += p1[e]; }
        }
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.bias[n + e];
        }
        if (a.rowvec) {
            const int idx_rv = ((m / a.rv_div) * a.rv_mul + (m % a.rv_mod_in)) % a.rv_mod_out;
            const float* q = a.rowvec + (size_t)idx_rv * a.N + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += q[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a.s_acc;
        if (a.r1 || a.r2) {                                        // rounded to fp16 before the residual add, as in the whole tiles
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)(f16)v[e];
        }
        if (a.r1) {
            const f16x8 t = *(const f16x8*)((const f16*)a.r1 + (size_t)m * a.ldr1 + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.s1 * (float)t[e];
        }
        if (a.r2) {
            const f16x8 t = *(const f16x8*)((const f16*)a.r2 + (size_t)m * a.ldr2 + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a.s2 * (float)t[e];
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (a.act == MOFA_ACT_SILU) x = silu_f(x);
            else if (a.act == MOFA_ACT_RELU) x = fmaxf(x, 0.0f);
            else if (a.act == MOFA_ACT_GELU) x = gelu_erf_f(x);
            o[e] = (f16)x;
        }
        *(f16x8*)((f16*)a.out + (size_t)m * a.ldo + n) = o;
    }
}

// STATS launches with a split-K last round: the remainder tiles get their outputs from the fix-up kernel, so their pair sums are
// taken from the finished outputs (R tiles x 4 row blocks; a few MB, L2 / MALL resident): one workgroup per (tile, 64-row block),
// one thread per column pair, rows in order
__global__ __launch_bounds__(192) void igemm320_tile_stats_kernel(const f16* __restrict__ out, const int ldo, float* __restrict__ stats,
                                                                  const int M, const int N, const int tile0, const int tilesN) {
    constexpr int TBN = Geo<5>::TBN;
    const int tile = tile0 + (int)(blockIdx.x >> 2), tm = tile / tilesN, tn = tile - tm * tilesN;
    const int m0 = tm * TBM3 + 64 * (int)(blockIdx.x & 3), n = tn * TBN + 2 * (int)threadIdx.x;
    if (m0 >= M || (int)threadIdx.x >= TBN / 2 || n + 2 > N) return;
    const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
    float sm = 0.f, q = 0.f;
    const f16* p = out + (size_t)m0 * ldo + n;
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
        const f16x2 h = *(const f16x2*)(p + (size_t)r * ldo);
        sm = __builtin_amdgcn_fdot2(h, one2, sm, false);
        q = __builtin_amdgcn_fdot2(h, h, q, false);
    }
    float* sp = stats + (size_t)(m0 >> 6) * N + n;
    sp[0] = sm;
    sp[1] = q;
}

}  // namespace

// (kind 7 = row vector + two residuals does not fit the register file beside 160 accumulators and occurs nowhere in the
// model graph: it runs on the 4-wave tiles)
static const igemm320_kern_t k_igemm320[9] = {igemm320_f16_kernel<0, 5>, igemm320_f16_kernel<1, 5>, igemm320_f16_kernel<2, 5>,
                                               igemm320_f16_kernel<3, 5>, igemm320_f16_kernel<4, 5>, igemm320_f16_kernel<5, 5>,
                                               igemm320_f16_kernel<6, 5>, nullptr, igemm320_f16_kernel<8, 5>};
static const igemm320_kern_t k_igemm320_split[7] = {
    igemm320_f16_kernel<0, 5, true>, igemm320_f16_kernel<1, 5, true>, igemm320_f16_kernel<2, 5, true>, igemm320_f16_kernel<3, 5, true>,
    igemm320_f16_kernel<4, 5, true>, igemm320_f16_kernel<5, 5, true>, igemm320_f16_kernel<6, 5, true>};

// STATS instantiations (GroupNorm pair sums from the epilogue): the kinds the networks' GroupNorm producers use -- plain (down-sampling
// conv), one residual (conv2 / the temporal conv2 / proj_out), uniform row vector (conv1 + time embedding), both
static const igemm320_kern_t k_igemm320_stats[7] = {igemm320_f16_kernel<0, 5, false, true>, igemm320_f16_kernel<1, 5, false, true>, nullptr,
                                                     nullptr, igemm320_f16_kernel<4, 5, false, true>, igemm320_f16_kernel<5, 5, false, true>, nullptr};
static const igemm320_kern_t k_igemm320_split_stats[7] = {igemm320_f16_kernel<0, 5, true, true>, igemm320_f16_kernel<1, 5, true, true>, nullptr,
                                                           nullptr, igemm320_f16_kernel<4, 5, true, true>, igemm320_f16_kernel<5, 5, true, true>, nullptr};

int igemm320_init() {
    for (igemm320_kern_t k : k_igemm320)
        if (k && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<5>::LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
    for (igemm320_kern_t k : k_igemm320_split)
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<5>::LDS_BYTES) != hipSuccess)
            return MOFA_ELAUNCH;
    for (int i = 0; i < 7; ++i)
        for (igemm320_kern_t k : {k_igemm320_stats[i], k_igemm320_split_stats[i]})
            if (k && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<5>::LDS_BYTES) != hipSuccess)
                return MOFA_ELAUNCH;
    return MOFA_OK;
}

// can a launch with these arguments emit mofa_igemm_args.stats?  (a->stats itself is not looked at)
bool igemm320_stats_ok(const mofa_igemm_args* a) {
    if (!a || !a->x || !a->w || !a->out || a->M <= 0 || a->N <= 0 || a->Cin <= 0 || a->Cin % 64 != 0) return false;
    if (a->act != MOFA_ACT_NONE || a->r2 || (a->r1 && a->s1 != 1.0f)) return false;
    if (a->M % 64 != 0 || a->N % Geo<5>::TBN != 0) return false;
    if (a->rowvec && (a->rv_mod_in != 1 || a->rv_div <= 0 || a->rv_div % 64 != 0)) return false;   // constant over a wave's 64 rows
    if (a->tile != 0 && a->tile != MOFA_TILE_256X320) return false;
    const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
    const int kind = (a->r1 ? 1 : 0) | (a->rowvec ? 4 : 0);
    if (!igemm_pipe_eligible(a, kind, (long long)taps * a->Cin)) return false;
    if ((long long)(a->N + Geo<5>::TBN) * taps * a->Cin * 2 >= 0x7ff00000LL) return false;
    return true;
}

// Split-K for the tiles of a partial last round.  T tiles on n_cu persistent workgroups take ceil(T / n_cu) tile times although
// the last round may hold only a few tiles (M = 460800, N = 320: 1800 tiles = 7 rounds + 8 tiles; M = 7200: 116 tiles on 256
// CUs).  When a workspace is supplied, the R = T mod n_cu remainder tiles are cut into S K-slices (R * S <= n_cu work items of
// nk / S K tiles each, fp32 partial tiles in the workspace) and a small fix-up kernel adds the slices in slice order and
// applies the epilogue: the round shrinks to about 1 / S of a tile time.  Returns S (1 = no split).
int igemm320_split(long long T, int nk, int n_cu, long long ws_bytes) {
    const long long R = T % n_cu;
    if (ws_bytes <= 0 || R == 0) return 1;
    if (T > n_cu && R * 10 > (long long)n_cu * 6) return 1;   // a last round that is more than 60 % full stays whole
    long long s = n_cu / R;
    if (s > 8) s = 8;
    if (s > nk / 8) s = nk / 8;                                // at least eight K tiles per slice, and ... (r04: up to 16 slices
                                                               // of >= 4 K tiles changes nothing on the per-rank shapes of an
                                                               // 8-GPU run: mix 691 against 693 TF/s, profiles/r04_shard_shapes_tiles.log)
    // ... only where it pays: a slice saves (1 - 1 / S) of a tile's K loop (about 2 us per K tile) but costs the partial-tile
    // dump and the fix-up launch (about 25 us; profiles/archive/r03b_kernel_stats_bench.md: splitting the shallow-K launches of the
    // clip -- 5 100 of 12 012 -- made it 4 % SLOWER)
    if (s >= 2 && 2.0 * nk * (1.0 - 1.0 / (double)s) < 50.0) s = 1;
    const long long per = (long long)TBM3 * Geo<5>::TBN * 4;
    while (s >= 2 && R * s * per > ws_bytes) --s;
    return s >= 2 ? (int)s : 1;
}

// returns 0 launched, < 0 error, 1 not eligible (the caller falls back to another tile)
int igemm320_launch(const mofa_igemm_args* a, int kind, int n_cu, hipStream_t stream) {
    const int taps = a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize * a->ksize : 9) : (a->mode == MOFA_MODE_CONVT3 ? 3 : 1);
    if (kind == 7 || !igemm_pipe_eligible(a, kind, (long long)taps * a->Cin)) return 1;
    constexpr int tbn = Geo<5>::TBN;
    if ((long long)(a->N + tbn) * taps * a->Cin * 2 >= 0x7ff00000LL) return 1;   // unclamped W row offsets stay below W_DEAD
    const int tilesM = cdiv(a->M, TBM3), tilesN = cdiv(a->N, tbn);
    const long long nt = (long long)tilesM * tilesN;
    if (nt > 0x7fffffffLL) return MOFA_EINVAL;
    Aux aux = igemm_pipe_aux(a, taps, tilesN);
    const int nk = taps * (a->Cin / 64);
    const bool ws_ok = a->workspace && (((size_t)a->workspace) & 15) == 0;
    const int S = kind == 8 ? 1 : igemm320_split(nt, nk, n_cu, ws_ok ? a->workspace_bytes : 0);   // (the fix-up has no GEGLU form)
    if (a->stats && (kind > 6 || !k_igemm320_stats[kind] || (((size_t)a->stats) & 15) || !igemm320_stats_ok(a))) return MOFA_EINVAL;
    if (S == 1) {
        int grid = (int)(nt < n_cu ? ((nt + 7) / 8) * 8 : (n_cu / 8) * 8);
        if (grid < 8) grid = 8;
        hipLaunchKernelGGL(a->stats ? k_igemm320_stats[kind] : k_igemm320[kind], dim3(grid), dim3(512), Geo<5>::LDS_BYTES, stream, *a, tilesN,
                           (int)nt, aux);
        MOFA_CHECK_LAUNCH();
        return MOFA_OK;
    }
    // one launch: the whole tiles of the full rounds, then one K slice of a remainder tile per workgroup; then the fix-up
    const long long full = nt - nt % n_cu;
    const int R = (int)(nt - full), items = R * S;
    aux.nsplit = S;
    aux.nsplit_d = fastdiv_make(S);
    aux.tile0 = (int)full;
    aux.nitems = items;
    aux.ws = (float*)a->workspace;
    int grid = full > 0 ? (n_cu / 8) * 8 : ((items + 7) / 8) * 8;
    if (grid < items) return MOFA_EINVAL;                       // (R * S <= n_cu by construction)
    hipLaunchKernelGGL(a->stats ? k_igemm320_split_stats[kind] : k_igemm320_split[kind], dim3(grid), dim3(512), Geo<5>::LDS_BYTES, stream, *a,
                       tilesN, (int)full, aux);
    MOFA_CHECK_LAUNCH();
    const long long pieces = (long long)R * TBM3 * (tbn / 8);
    hipLaunchKernelGGL(igemm320_fixup_kernel, dim3((int)((pieces + 255) / 256)), dim3(256), 0, stream, *a,
                       (const float*)a->workspace, (int)full, S, tilesN, R);
    MOFA_CHECK_LAUNCH();
    if (a->stats) {                                              // the remainder tiles' pair sums, from their finished outputs
        hipLaunchKernelGGL(igemm320_tile_stats_kernel, dim3(R * 4), dim3(192), 0, stream, (const f16*)a->out, a->ldo, a->stats, a->M,
                           a->N, (int)full, tilesN);
        MOFA_CHECK_LAUNCH();
    }
    return MOFA_OK;
}
