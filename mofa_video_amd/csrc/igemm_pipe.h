// Pieces shared by the two 8-wave phase-pipelined implicit-GEMM kernels (igemm8.hip: 256x256 tile, igemm320.hip: 256x320
// tile): launch-invariant scalars, magic-number division, the packed row geometry of a DMA row group and its tap source.
#pragma once
#include "igemm_common.h"

namespace {

// division by a launch-invariant divisor without v_rcp sequences (whose loop-invariant parts hipcc hoists out of the
// persistent loop and then spills): q = umulhi(n, mul) >> shr, exact for 0 <= n < 2^31 (host side: fastdiv_make)
struct FastDiv { unsigned mul, shr; };
__device__ __forceinline__ int fdiv(int n, const FastDiv d) { return d.mul ? (int)(__umulhi((unsigned)n, d.mul) >> d.shr) : n; }

struct Aux {                        // launch-invariant scalars computed by the launcher
    FastDiv tiles_n, hw, wout, t3hw, t3t;
    int ldxb;                       // activation row stride in bytes
    int row_shift;                  // rows between xbase and a.x (convT3 with caller-supplied halo frames: HW, else 0)
    const void* xbase;              // base of the activation buffer descriptor
    unsigned x_bytes, w_bytes;      // extents of the two buffer descriptors
    unsigned long long* trace;      // VAR & 64: [workgroup][group][16] cycle sums
    // split-K launches of igemm320.hip (remainder tiles of a partial last round): work item = (tile0 + item / nsplit, K slice
    // item % nsplit); fp32 partial tiles [item][256][320] go to ws
    FastDiv nsplit_d, kpt_d, ks_d;
    int nsplit, tile0, nitems;
    float* ws;
};

// One DMA row group = ONE packed register (Cursor::gx), decoded at every tap switch:
//   plain    m                                         conv     img << 20 | oy << 10 | ox
//   convT3   m | (frame > 0 or unclipped) << 29 | (frame < T - 1 or unclipped) << 30
//   -1       row beyond M
// DMA sources are addressed through BUFFER descriptors (buffer_load_dwordx4 ... lds): a 32-bit byte offset per lane, the
// K offset in an SGPR, no 64-bit address arithmetic -- and an offset beyond the descriptor's extent reads as ZERO, which
// is how rows of an out-of-image tap / beyond M / past the end of the tile walk are sourced (no zero page, no select).
constexpr unsigned XO_INVALID = 0xffffffffu;

template <int V> struct IC { static constexpr int v = V; };


}  // namespace

// q = umulhi(n, mul) >> shr == n / d for 0 <= n < 2^31 (mul == 0: d == 1)
static inline FastDiv fastdiv_make(int d) {
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

// rows of the activation buffer the launch may address (convT3 without clipping: one halo frame on either side)
static inline long long igemm8_rows_in(const mofa_igemm_args* a) {
    if (a->mode == MOFA_MODE_CONV3X3) return a->M / ((long long)a->Hout * a->Wout) * a->Hin * a->Win;
    if (a->mode == MOFA_MODE_CONVT3 && a->T == 0) return (long long)a->M + 2ll * a->HW;
    return a->M;
}


// 16-byte row alignment everywhere (the kernel has no narrow-store path); packed row geometry and 32-bit offsets in range
static inline bool igemm_pipe_eligible(const mofa_igemm_args* a, int kind, long long Ktot) {
    const int nout = kind == 8 ? a->N / 2 : a->N;
    if ((a->ldo & 7) || (nout & 7) || (a->N & 7) || (((size_t)a->out) & 15) || (((size_t)a->x) & 15)) return false;
    if (a->r1 && ((a->ldr1 & 7) || (((size_t)a->r1) & 15))) return false;
    if (a->r2 && ((a->ldr2 & 7) || (((size_t)a->r2) & 15))) return false;
    if (a->bias && (((size_t)a->bias) & 15)) return false;
    if ((a->r1 || a->r2 || a->rowvec) && a->act != MOFA_ACT_NONE) return false;   // only the plain kind carries activation code here
    if (a->rowvec && (((size_t)a->rowvec) & 15)) return false;
    if ((long long)a->N * Ktot * 2 >= (1ll << 32) || (((size_t)a->w) & 15)) return false;
    if (a->mode == MOFA_MODE_CONV3X3) {
        const long long nimg = a->M / ((long long)a->Hout * a->Wout);
        if (a->Hout > 1024 || a->Wout > 1024 || nimg > 2047) return false;
    } else if (a->mode == MOFA_MODE_CONVT3) {
        if (a->M >= (1 << 29)) return false;
    }
    if (igemm8_rows_in(a) * a->ldx * 2 >= (1ll << 32) - 65536) return false;   // 32-bit buffer offsets
    return true;
}


// launch-invariant scalars of a launch (tilesN = output tile columns of the chosen tile)
static inline Aux igemm_pipe_aux(const mofa_igemm_args* a, int taps, int tilesN) {
    Aux aux;
    aux.tiles_n = fastdiv_make(tilesN);
    aux.hw = fastdiv_make(a->mode == MOFA_MODE_CONV3X3 ? a->Hout * a->Wout : 1);
    aux.wout = fastdiv_make(a->mode == MOFA_MODE_CONV3X3 ? a->Wout : 1);
    aux.t3hw = fastdiv_make(a->mode == MOFA_MODE_CONVT3 ? a->HW : 1);
    aux.t3t = fastdiv_make(a->mode == MOFA_MODE_CONVT3 && a->T > 0 ? a->T : 1);
    const bool halo = a->mode == MOFA_MODE_CONVT3 && a->T == 0;   // rows before a.x are read (tap -1 of the first frame)
    aux.ldxb = a->ldx * 2;
    aux.row_shift = halo ? a->HW : 0;
    aux.xbase = (const char*)a->x - (halo ? (size_t)a->HW * a->ldx * 2 : 0);
    aux.x_bytes = (unsigned)(igemm8_rows_in(a) * a->ldx * 2 - (a->ldx - a->Cin) * 2);
    aux.w_bytes = (unsigned)((long long)a->N * taps * a->Cin * 2);
    aux.trace = nullptr;
    aux.nsplit_d = fastdiv_make(1);
    aux.kpt_d = fastdiv_make(a->Cin / 64);
    aux.ks_d = fastdiv_make(a->mode == MOFA_MODE_CONV3X3 ? (a->ksize > 0 ? a->ksize : 3) : 1);
    aux.nsplit = 1;
    aux.tile0 = 0;
    aux.nitems = 0;
    aux.ws = nullptr;
    return aux;
}
