// 320-input-channel linear layers for gfx950, TRANSPOSED form: out^T[n, token] = W[n, :] . X^T  (+ LayerNorm in front, bias, a
// per-row-group vector, a residual), N a multiple of 64.
//
// Replaces mofa_layernorm_f16 + mofa_igemm_f16 for the level-0 (C = 320) projections of diffusers' BasicTransformerBlock /
// TemporalBasicTransformerBlock / TransformerSpatioTemporalModel as the reference builds them
// (MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:169-232, models/controlnet_sdv.py:259-309): norm1 + to_q|k|v
// (N = 960), to_out.0 + residual + cross-attention vector (N = 320), proj_in (N = 320).  At M = 460 800 tokens these are the
// launches the 256x320 implicit-GEMM tile runs at 430-660 TF/s: five K tiles per output tile do not cover its pipeline fill, and
// the LayerNorm in front is a read + write of the whole tensor.
//
// Same building blocks as csrc/ff320.hip (the fused feed-forward): a wave owns 32 token rows; X^T -- lane (token l31, k half lh)
// holds X[token][16 s + 8 lh .. + 7] for the 20 k-steps -- is loaded once per 256-row tile straight from global memory into 80
// registers and is the B operand of every MFMA; the LayerNorm (optional; gain / bias folded into W / bias at load,
// weights.pack_lin320) is 160 values per lane plus one cross-half exchange.  W is walked in chunks of 64 output columns (two
// 32x32 tiles x 20 k-steps = 40 MFMAs, 40 KB) that arrive by LDS-DMA as lane-linear 1 KB blocks (the packed global image IS the
// LDS image; conflict-free ds_read_b128 at base + immediate) into a ring of three, two steps ahead.  Unlike the feed-forward
// there is no VALU-heavy stage and a chunk's two accumulator tiles are FINAL after its 40 MFMAs: they are stored at once (fp16
// rounding of s_acc * acc before the residual add like every implicit-GEMM tile, v_permlane32_swap for 16-byte pieces), so a
// wave needs ~150 registers and the kernel runs TWO waves per SIMD (8 waves, 256 rows per workgroup): one wave's epilogue and
// fragment reads under the other's MFMAs.
#include "common.h"

namespace {

constexpr int LN_C = 320, LN_KS = 20, LN_CHUNK = 2 * LN_KS * 1024;       // 40 KB per chunk of 64 output columns
constexpr int LN_OFF_T = 3 * LN_CHUNK;                                   // per wave 4 KB: a chunk's 32 rows x 64 columns, for the store transpose
constexpr int LN_OFF_B = LN_OFF_T + 8 * 4096;                            // ring of three 1 KB bias pieces (a chunk's 64 floats first)
constexpr int LN_LDS_BYTES = LN_OFF_B + 3 * 1024;                        // 158 720
#ifndef LN_LOOK_D
#define LN_LOOK_D 6
#endif
constexpr int LN_LOOK = LN_LOOK_D;                                        // fragment reads in flight ahead of the MFMA that consumes them

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IC { static constexpr int v = V; };
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}
#define LN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// Step barrier with a COUNTED wait.  vmcnt counts loads, LDS-DMA pieces and stores alike, in issue order, so a wait for anything also
// waits for every store issued before it -- and a store's round trip to L2 under load is longer than a step.  The first form of this
// kernel fetched a step's bias by global loads: hipcc's own s_waitcnt in front of the step's first MFMA then also waited for the
// previous step's stores, every step.  Now nothing a step needs is YOUNGER than the previous step's stores: the bias arrives by
// LDS-DMA with its chunk, two steps ahead, and the only wait of a step is this one, for chunk g's pieces, which lets the K
// youngest operations stay in flight:
//   [step g - 2, after its DMA]  4 stores   [step g - 1]  (4 residual loads)  6 DMA pieces of chunk g + 1, 4 stores   = 14 (+ 4)
// (Measured and dropped, profiles/r06_lin320.log: the two waves of a SIMD half a step apart -- an MFMA phase and an epilogue phase,
// two barriers per step -- is 5-20 % SLOWER than both waves in the same phase with one barrier.)
// The counts are exact because every step issues exactly that: the stores are buffer stores (rows beyond M are dropped by the
// descriptor's bounds check instead of being branched around), the pieces are always issued (beyond the last chunk they re-fetch).
// (LIN_T_*: timing-only builds for tools/lin320_anatomy.sh -- wrong results, the data flow X -> MFMAs -> outputs stays alive)
template <int K>
__device__ __forceinline__ void lin_barrier() {
#ifdef LIN_T_NOBAR
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(K) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(K) : "memory");
#endif
}

template <bool NORM, bool RV, bool R1>
__global__ __launch_bounds__(512, 2) void lin320_kernel(const mofa_lin320_args a, const int ntiles, const int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_ln[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, (unsigned)nchunk * LN_CHUNK, 0x00020000);
    // no bias: a descriptor of zero records -- the bounds check then fills the LDS piece with zeros
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : a.wp), 0, a.bias ? (unsigned)a.N * 4u : 0u, 0x00020000);
    const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((size_t)a.M * a.ldo * 2), 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    // a wave fetches the contiguous blocks 5 wave .. 5 wave + 4 of a chunk image (4 + 1 pieces on two LDS bases: the instruction's
    // immediate offset advances both addresses, tools/lds_dma_range.hip)
    auto dma_chunk = [&](int chunk, int slot) __attribute__((always_inline)) {
        char* dst = smem_ln + slot * LN_CHUNK + wave * 5120;
        const int src = chunk * LN_CHUNK + wave * 5120;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LN_LDS_PTR(dst), 16, voff, src, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LN_LDS_PTR(dst), 16, voff, src, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LN_LDS_PTR(dst), 16, voff, src, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LN_LDS_PTR(dst), 16, voff, src, 3072, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LN_LDS_PTR(dst + 4096), 16, voff, src + 4096, 0, 0);
        // sixth piece: the chunk's 64 bias floats = the first 256 bytes of a 1 KB piece (the same lane offsets as the weight pieces;
        // every wave fetches the same bytes to the same place: one count for all; beyond N the bounds check fills zeros)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, LN_LDS_PTR(smem_ln + LN_OFF_B + slot * 1024), 16, voff, chunk * 256, 0, 0);
    };
    const char* wl = smem_ln + lane * 16;

    f16x8 xf[LN_KS];
    const f16* xg = (const f16*)a.x;
    const f16* rg = (const f16*)a.r1;
    const float s_acc = a.s_acc, s1 = a.s1;

    // the ring runs on a counter that does not restart at tile boundaries: chunk g (g = tile-local chunk + nchunk * tiles done)
    // lives in slot g % 3 and is fetched two steps ahead
    int g = 0;
#ifdef LIN_T_NOEPI
    float tkeep = 0.f;
#endif
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nchunk;
    dma_chunk(0, 0);
    if (total > 1) dma_chunk(1 % nchunk, 1);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces of chunks 0 and 1 (the step barrier makes them everyone's)
    // token rows of the NEXT tile are fetched during the current tile's last chunk (xn): with both waves of a SIMD in lockstep nothing
    // else would cover the load latency at a tile boundary (five chunks per tile at N = 320)
    f16x8 xn[LN_KS];
    {
        const int m0 = (int)blockIdx.x * 256 + wave * 32 + l31;
        const f16* xrow0 = xg + (size_t)(m0 < a.M ? m0 : 0) * a.ldx;
#pragma unroll
        for (int s = 0; s < LN_KS; ++s) xn[s] = *(const f16x8*)(xrow0 + 16 * s + 8 * lh);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 256 + wave * 32 + l31;
        const bool rok = m < a.M;
        const int mr = rok ? m : 0;
#pragma unroll
        for (int s = 0; s < LN_KS; ++s) xf[s] = xn[s];
        if constexpr (NORM) {
            // LayerNorm in fp32, three passes over the fp16 fragments (mean, centred squares, normalise); gain / bias live in W / bias
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < LN_KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)xf[s][e];
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / LN_C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < LN_KS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xf[s][e] - mean; sq = fmaf(d, d, sq); }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / LN_C) + a.eps);
#pragma unroll
            for (int s = 0; s < LN_KS; ++s) {
                f16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (f16)(((float)xf[s][e] - mean) * rstd);
                xf[s] = y;
            }
        }
        const float* rvrow = nullptr;
        if constexpr (RV) rvrow = a.rowvec + (size_t)(((mr / a.rv_div) * a.rv_mul + (mr % a.rv_mod_in)) % a.rv_mod_out) * a.N;
        // row-contiguous layout of the stores / residual loads: lane j <-> row (j >> 3) + 8 i of the wave's 32, columns 8 (j & 7) ..
        const int mw = tile * 256 + wave * 32;
        const int mj = mw + (lane >> 3) < a.M ? mw + (lane >> 3) : 0;       // (rows beyond M: loads from row 0, stores masked)
        const f16* rbase = R1 ? rg + (size_t)mj * a.ldr1 + 8 * (lane & 7) : nullptr;
        const unsigned ldo2 = (unsigned)a.ldo * 2u;
        const unsigned obyte = (unsigned)(mw + (lane >> 3)) * ldo2 + 16u * (unsigned)(lane & 7);   // (unclamped: see the stores)

        for (int c = 0; c < nchunk; ++c, ++g) {
            lin_barrier<R1 ? 18 : 14>();                           // chunk g (+ its bias piece) has landed for everyone; slot (g + 2) % 3 is free
            // the residual pieces of this chunk (in the layout the stores use, below: lane j holds 16 bytes = segment j % 8 of row
            // j / 8 + 8 i of the wave's 32 rows -- whole 128-byte lines per instruction) and, in a tile's last chunk, the NEXT tile's
            // token rows are issued BEFORE the step's DMA pieces: what the epilogue waits for is then older than the pieces, which
            // stay in flight across the step
            f16x8 rp[4];
            if constexpr (R1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rp[i] = *(const f16x8*)(rbase + (size_t)(8 * i) * a.ldr1 + 64 * c);
            }
            if (c == nchunk - 1 && tile + (int)gridDim.x < ntiles) {
                const int m2 = (tile + (int)gridDim.x) * 256 + wave * 32 + l31;
                const f16* xrow2 = xg + (size_t)(m2 < a.M ? m2 : 0) * a.ldx;
#pragma unroll
                for (int s = 0; s < LN_KS; ++s) xn[s] = *(const f16x8*)(xrow2 + 16 * s + 8 * lh);
            }
            {                                                      // (always six pieces, so that the barrier's count holds: beyond the
                int cn = c + 2;                                    //  last chunk they re-fetch a chunk nobody reads)
                cn = cn >= nchunk ? cn - nchunk : cn;
                cn = cn >= nchunk ? cn - nchunk : cn;             // (nchunk >= 1: c + 2 wraps at most twice)
#ifndef LIN_T_NODMA
                dma_chunk(cn, (g + 2) % 3);
#endif
            }
            // bias (+ row vector) = the accumulators' first value; accumulator register r of tile t <-> column
            // 64 c + 32 t + 8 (r >> 2) + 4 lh + (r & 3): the bias piece is read from LDS (both k halves broadcast their 16 bytes)
            const int n0 = 64 * c + 4 * lh;
            const char* bl = smem_ln + LN_OFF_B + (g % 3) * 1024 + 16 * lh;
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 b = *(const f32x4*)(bl + 128 * t + 32 * q);
                    if constexpr (RV) b += *(const f32x4*)(rvrow + n0 + 32 * t + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = b[e];
                }
            const char* w = wl + (g % 3) * LN_CHUNK;
            auto frag = [&](int i) __attribute__((always_inline)) -> f16x8 {   // slot i: k-step i / 2, tile i % 2
                return *(const f16x8*)(w + ((i & 1) * LN_KS + (i >> 1)) * 1024);
            };
            constexpr int RING = LN_LOOK + 2;
            f16x8 ring[RING];
#pragma unroll
            for (int i = 0; i < LN_LOOK; ++i) ring[i] = frag(i);
            static_for<0, 2 * LN_KS>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::v;
                // reads two at a time on even slots, the later fragment first: one s_waitcnt per two MFMAs (LDS returns in order)
                if constexpr ((i & 1) == 0) {
                    if constexpr (i + LN_LOOK + 1 < 2 * LN_KS) ring[(i + LN_LOOK + 1) % RING] = frag(i + LN_LOOK + 1);
                    if constexpr (i + LN_LOOK < 2 * LN_KS) ring[(i + LN_LOOK) % RING] = frag(i + LN_LOOK);
                }
                acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[i % RING], xf[i >> 1], acc[i & 1], 0, 0, 0);
            });
            // epilogue of the chunk.  After the v_permlane32_swap a lane holds 4 pieces of 8 consecutive columns of ITS row (l31): stored
            // like that, an instruction would write 32 bytes into each of 32 rows -- quarter lines, which is what bounded the first
            // form of this kernel (564 TF/s at N = 960 whatever the waits).  So the chunk's 32 x 64 outputs go through 4 KB of LDS
            // private to the wave ([row][8 segments of 16 bytes], segment index XOR (row & 7): conflict-free both ways) and leave as
            // whole 128-byte lines: lane j stores segment j % 8 of rows j / 8 + 8 i.
            char* ts = smem_ln + LN_OFF_T + wave * 4096;
#ifdef LIN_T_NOEPI
            tkeep += acc[0][0] + acc[1][0];                         // (the chunk's MFMAs stay alive; one store at the end of the kernel)
            continue;
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f16x4 gq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) gq[q][e] = (f16)(s_acc * acc[t][4 * q + e]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x2 a2 = __builtin_bit_cast(u32x2, gq[2 * k]), b2 = __builtin_bit_cast(u32x2, gq[2 * k + 1]);
                    const auto rx = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
                    const auto ry = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
                    const u32x4 vv = {rx[0], ry[0], rx[1], ry[1]};
                    const int seg = 4 * t + 2 * k + lh;            // columns 8 seg .. 8 seg + 7 of the chunk, row l31
                    *(u32x4*)(ts + l31 * 128 + ((seg ^ (l31 & 7)) * 16)) = vv;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the wave's own writes; nobody else touches this scratch)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (lane >> 3) + 8 * i;
                f16x8 o = *(const f16x8*)(ts + r * 128 + (((lane & 7) ^ (r & 7)) * 16));
                if constexpr (R1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)o[e] + s1 * (float)rp[i][e]);
                }
                // (a row beyond M lies beyond the descriptor's records: the store is issued and dropped)
#ifdef LIN_T_NOSTORE
                if (o[0] == (f16)12345.f)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, obyte + (unsigned)(8 * i) * ldo2 + 128u * (unsigned)c, 0, 0);
            }
        }
    }
#ifdef LIN_T_NOEPI
    if (tkeep == 12345.f) ((float*)a.out)[lane] = tkeep;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef void (*lin320_kern_t)(const mofa_lin320_args, const int, const int);
template <bool NORM, bool RV, bool R1>
lin320_kern_t lin320_pick() { return lin320_kernel<NORM, RV, R1>; }

}  // namespace

extern "C" int mofa_lin320_f16(const mofa_lin320_args* a, mofa_stream_t stream) {
    if (!a || !a->x || !a->wp || !a->out || a->M <= 0 || a->N <= 0 || (a->N & 63)) return MOFA_EINVAL;
    if ((a->ldx & 7) || (a->ldo & 7) || a->ldx < LN_C || a->ldo < a->N || (((size_t)a->x) & 15) || (((size_t)a->out) & 15) ||
        (((size_t)a->wp) & 15) || (a->bias && (((size_t)a->bias) & 15)))
        return MOFA_EINVAL;
    if (a->rowvec && (a->rv_div <= 0 || a->rv_mod_in <= 0 || a->rv_mod_out <= 0 || (((size_t)a->rowvec) & 15))) return MOFA_EINVAL;
    if (a->r1 && ((a->ldr1 & 7) || a->ldr1 < a->N || (((size_t)a->r1) & 15))) return MOFA_EINVAL;
    if (((long long)a->M + 256) * a->ldo * 2 >= 0xffffffffLL) return MOFA_EINVAL;   // outputs are addressed through a 32-bit buffer descriptor
    static const lin320_kern_t kerns[8] = {
        lin320_pick<false, false, false>(), lin320_pick<true, false, false>(), lin320_pick<false, true, false>(), lin320_pick<true, true, false>(),
        lin320_pick<false, false, true>(),  lin320_pick<true, false, true>(),  lin320_pick<false, true, true>(),  lin320_pick<true, true, true>()};
    static const int n_cu = [] {
        int dev = 0, cus = 0;
        for (int i = 0; i < 8; ++i)
            (void)hipFuncSetAttribute((const void*)kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS_BYTES);
        return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                cus > 0) ? cus : 256;
    }();
    const int ntiles = (a->M + 255) / 256;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const lin320_kern_t k = kerns[(a->norm ? 1 : 0) | (a->rowvec ? 2 : 0) | (a->r1 ? 4 : 0)];
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), LN_LDS_BYTES, (hipStream_t)stream, *a, ntiles, a->N / 64);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
