// Pieces shared by the two implicit-GEMM translation units (igemm.hip: 4-wave tiles + the launcher; igemm8.hip: the
// 8-wave 256x256 phase-pipelined kernel): implicit-GEMM row geometry, the XCD-aware persistent tile walk, the LDS-DMA
// primitive and the counted waits.
#pragma once
#include "common.h"

struct RowGeo {   // two registers per DMA row group
    int a;        // plain / convT3: global output row m; conv: image index; -1 when the row is beyond M
    int b;        // conv: (oy << 16) | ox of the output pixel; convT3: frame index within its clip
};

// source of out-of-image taps / rows beyond M (one copy per translation unit; never written)
static __device__ __attribute__((aligned(128))) f16 g_zero_page[128];

__device__ __forceinline__ RowGeo make_geo(const mofa_igemm_args& a, int m) {
    RowGeo g;
    g.a = (m < a.M) ? m : -1;
    g.b = 0;
    // the divisors pass through an empty asm so that the reciprocal sequences of these integer divisions are set up HERE,
    // per call (once per output tile): hoisted out of the persistent tile loop hipcc spills them to scratch and reloads
    // them with an s_waitcnt vmcnt(0) that also drains the next tile's DMA
    if (a.mode == MOFA_MODE_CONV3X3) {
        int hw = a.Hout * a.Wout, wout = a.Wout;
        asm volatile("" : "+s"(hw), "+s"(wout));
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / wout;
        g.a = (m < a.M) ? img : -1;
        g.b = (oy << 16) | (rem - oy * wout);
    } else if (a.mode == MOFA_MODE_CONVT3) {
        int HW = a.HW, T = a.T;
        asm volatile("" : "+s"(HW), "+s"(T));
        g.b = T > 0 ? (m / HW) % T : 0;
    }
    return g;
}

__device__ __forceinline__ const f16* x_src(const mofa_igemm_args& a, const RowGeo& g, int tap) {
    if (g.a < 0) return nullptr;
    const f16* x = (const f16*)a.x;
    if (a.mode == MOFA_MODE_PLAIN) {
        return x + (size_t)g.a * a.ldx;
    } else if (a.mode == MOFA_MODE_CONV3X3) {
        const int ks = a.ksize > 0 ? a.ksize : 3;
        const int dil = a.dil > 0 ? a.dil : 1;
        const int ky = tap / ks, kx = tap - ky * ks;
        const int org = a.pad == MOFA_PAD_TRAILING ? 0 : (ks >> 1);
        const int vy = (g.b >> 16) * a.stride + (ky - org) * dil;
        const int vx = (g.b & 0xffff) * a.stride + (kx - org) * dil;
        if (vy < 0 || vx < 0 || vy >= a.Hin * a.up || vx >= a.Win * a.up) return nullptr;
        const int iy = (a.up == 2) ? (vy >> 1) : vy;
        const int ix = (a.up == 2) ? (vx >> 1) : vx;
        return x + ((size_t)(g.a * a.Hin + iy) * a.Win + ix) * a.ldx;
    } else {  // MOFA_MODE_CONVT3
        const int tt = g.b + tap - 1;
        if (a.T > 0 && (tt < 0 || tt >= a.T)) return nullptr;   // T == 0: unclipped, caller supplies halo frames
        return x + ((size_t)g.a + (size_t)(tap - 1) * a.HW) * a.ldx;
    }
}

__device__ __forceinline__ int igemm_taps(const mofa_igemm_args& a) {
    return (a.mode == MOFA_MODE_CONV3X3) ? (a.ksize > 0 ? a.ksize * a.ksize : 9) : (a.mode == MOFA_MODE_CONVT3 ? 3 : 1);
}

// XCD-aware persistent tile walk: the hardware deals workgroup ids round-robin over the 8 XCDs, so workgroup b lives on
// XCD b & 7.  Each XCD gets a CONTIGUOUS range of tile ids (bijective split of ntiles into 8 ranges) and its resident
// workgroups walk that range with stride (workgroups per XCD): at any time one XCD works on neighbouring tiles (same
// activation row block, successive weight column blocks) and the activation tile is served from that XCD's L2.
struct TileWalk {
    int start, count, local, stride;
    __device__ __forceinline__ void init(int ntiles) {
        const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
        start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        count = q + (xcd < r ? 1 : 0);
        local = blockIdx.x >> 3;
        stride = gridDim.x >> 3;
    }
};

#define EPI_R1 1
#define EPI_R2 2
#define EPI_RV 4
#define EPI_GEGLU 8

__device__ __forceinline__ void glds16(const f16* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// s_waitcnt vmcnt(N) lgkmcnt(0): see the K loops for why the LDS counter is drained with it
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_only() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// host side: igemm8.hip's launcher, called by mofa_igemm_f16 when the 8-wave tile is chosen / forced
typedef void (*igemm_kern_t)(const mofa_igemm_args, const int, const int);
int igemm8_launch(const mofa_igemm_args* a, int kind, int n_cu, hipStream_t stream);
int igemm8_init();
// igemm320.hip: the 256x320 tile (same return convention: 0 launched, < 0 error, 1 not eligible)
int igemm320_launch(const mofa_igemm_args* a, int kind, int n_cu, hipStream_t stream);
int igemm320_init();
bool igemm320_stats_ok(const mofa_igemm_args* a);             // mofa_igemm_args.stats can be emitted for these arguments
int igemm320_split(long long tiles, int nk, int n_cu, long long ws_bytes);   // K slices for the remainder tiles (1 = none)
