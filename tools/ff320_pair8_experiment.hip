// Fused level-0 feed-forward for gfx950: LayerNorm -> GEGLU projection (320 -> 2 x 1280) -> GELU gate -> output projection
// (1280 -> 320) -> residual(s), ONE launch, the [M, 1280] hidden state never leaves the CU.
//
// Replaces, for C = 320 (the 21 level-0 feed-forwards of a denoise step: 7 transformer layers x {spatial ff, temporal ff_in,
// temporal ff}), the three launches  mofa_layernorm_f16 -> mofa_igemm_f16(MOFA_ACT_GEGLU_PAIR) -> mofa_igemm_f16(r1, r2)  of
// diffusers' BasicTransformerBlock / TemporalBasicTransformerBlock feed-forward legs (FeedForward(activation_fn="geglu"),
// built at MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:169-232, models/controlnet_sdv.py:259-309):
// per launch 1.18 GB of hidden state written and read back, 2 x 295 MB of normalised tokens, and a K = 320 GEMM whose five K
// tiles cannot amortise its GELU epilogue (653-745 TF/s, profiles/r05_geglu_anatomy.log).
//
// Shape of the computation -- everything TRANSPOSED, so that a token row never leaves its lanes:
//   * a workgroup = 8 waves = 128 token rows; waves w and w + 4 (the two waves of one SIMD) form a PAIR that owns 32 rows.
//   * X^T is the B operand of GEMM 1:  P^T[proj row, token] = W1[proj row, :] . X^T.  Lane (token l31, k half lh) holds
//     X[token][16 s + 8 lh .. + 7] for the 20 k-steps s: 80 registers loaded once per tile straight from global memory (by both
//     waves of the pair); a token row lives in two lanes -- the LayerNorm is 160 values per lane plus one cross-half exchange,
//     in registers (gain / bias of the norm are folded into W1 / b1 at load time: mofa_video_amd/weights.py::pack_ff320).
//   * the hidden axis is walked in 40 chunks of 32.  GEMM 1 of chunk k (one value + one gate tile, 2 x 20 MFMAs 32x32x16, bias
//     as the accumulators' initial value) is done by ONE wave of the pair, the owner of k (wave half == k & 1); the GELU of the
//     chunk, H^T = value * gelu(gate), by the same wave ONE STEP LATER -- while its partner issues the MFMAs of chunk k + 1 on the
//     same SIMD.  That is the point of the pairing: with one 512-register wave per SIMD (the first form of this kernel) the
//     wave's own ~240 VALU instructions per chunk do not hide under its MFMAs (a wave issues about one instruction per 4-5 cycles:
//     1413 us against 996 with the GELU and the operand traffic removed, profiles/r06_ff320_anatomy.log); with two waves the
//     matrix pipe takes MFMAs from one while the other runs VALU.
//   * H^T, rounded to fp16, is handed to both waves through LDS as lane-linear 1 KB blocks (2 KB per chunk and pair) and IS the B
//     operand of GEMM 2 (O^T[out col, token] += W2[out col, hidden chunk] . H^T): the k-slot -> hidden permutation of that MFMA is
//     absorbed in the packed order of W2 (same trick as the P operand of attention.hip).  Each wave of the pair accumulates HALF
//     of the output columns (5 tiles = 80 registers): 256 registers per wave suffice (X 80, O 80, P 32, fragments).
//   * weights: per chunk 40 KB of W1 + 20 KB of W2 arrive by LDS-DMA (buffer_load ... lds) as lane-linear 1 KB blocks -- the
//     packed global image IS the LDS image, every fragment read is a conflict-free ds_read_b128 at base + immediate -- double
//     buffered (120 KB); 16 KB of H^T blocks; one barrier per chunk.  Weights are 2.4 MB per layer: L2 resident.
//   * per tile 42 steps: step k = { GEMM 1 of chunk k (owner) | GELU of chunk k - 1 (the other wave) } + GEMM 2 of chunk k - 2
//     (both waves, 10 MFMAs each).
//   * epilogue: fp16 rounding of s_acc * (O + b2) before the residual add like every implicit-GEMM tile (include/mofa_hip.h),
//     v_permlane32_swap gives a lane 8 consecutive output columns of its row: 16-byte residual loads and stores; optional
//     second output LayerNorm(out) (the norm in front of the NEXT projection; row sums exchanged between the pair through LDS).
#include "common.h"

namespace {

constexpr int FF_C = 320, FF_H = 1280, FF_NCHUNK = 40, FF_KS = 20, FF_NJ = 10, FF_NJW = 5;
constexpr int W1_SLOT = 2 * FF_KS * 1024;      // 40 KB: [value | gate tile][k-step][64 lanes x 16 B]
constexpr int W2_SLOT = FF_NJ * 2 * 1024;      // 20 KB: [out tile][k-step u][64 lanes x 16 B]
constexpr int OFF_W1 = 0, OFF_W2 = 2 * W1_SLOT, OFF_HF = OFF_W2 + 2 * W2_SLOT;     // HF: [chunk parity][pair][k-step u][64 x 16 B]
constexpr int OFF_LN = OFF_HF + 2 * 4 * 2048;                                       // [half][pair][32 rows] floats
constexpr int FF_LDS_BYTES = OFF_LN + 2 * 4 * 32 * 4;                               // 140 288

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V> struct IC { static constexpr int v = V; };
// compile-time loop: f(IC<0>{}), f(IC<1>{}), ... -- the slot schedules below need every index as a constant expression
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

#define FF_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// step barrier: this wave's DMA pieces have landed, its LDS reads / writes are done, then everybody is here
__device__ __forceinline__ void ff_barrier() {
#ifdef FF_T_NOBAR                              // timing-only build (wrong results)
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <bool POS, bool R2, bool LNOUT>
__global__ __launch_bounds__(512, 2) void ff320_kernel(const mofa_ff320_args a, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_ff[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, half = wave >> 2;                    // waves w, w + 4 share a SIMD and 32 token rows
    const int l31 = lane & 31, lh = lane >> 5;

    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1p, 0, 2u * FF_H * FF_C * 2u, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2p, 0, (unsigned)FF_C * FF_H * 2u, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    // a wave fetches blocks wave, wave + 8, ... of a chunk image: 5 of W1's 40, 3 or 2 of W2's 20 (1 KB each)
    auto dma_w1 = [&](int chunk, int slot) __attribute__((always_inline)) {
#ifndef FF_T_NODMA
        char* dst = smem_ff + OFF_W1 + slot * W1_SLOT + wave * 1024;
        const int src = chunk * W1_SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, FF_LDS_PTR(dst + i * 8192), 16, voff, src + i * 8192, 0, 0);
#endif
    };
    auto dma_w2 = [&](int chunk, int slot) __attribute__((always_inline)) {
#ifndef FF_T_NODMA
        char* dst = smem_ff + OFF_W2 + slot * W2_SLOT + wave * 1024;
        const int src = chunk * W2_SLOT + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, FF_LDS_PTR(dst + i * 8192), 16, voff, src + i * 8192, 0, 0);
        if (wave < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, FF_LDS_PTR(dst + 16384), 16, voff, src + 16384, 0, 0);
#endif
    };

    const char* wl = smem_ff + lane * 16;                          // this lane's 16 bytes of every 1 KB block
    const float* b1l = a.b1 + 4 * lh;

    f16x8 xf[FF_KS];                                               // X^T fragments (normalised) of the pair's 32 rows
    f32x16 O[FF_NJW];                                              // this wave's half of O^T: out tiles 5 half .. 5 half + 4
    f32x16 pv, pg;                                                 // P^T value / gate tile of the chunk this wave owns (VGPRs, see mfma_p)
    constexpr int FF_LOOK = 4;

    // A's MFMA with the accumulator in VGPRs, through asm: the GELU reads these tiles with VALU instructions, and wherever hipcc
    // parks an accumulator in the AGPR half every read is a v_accvgpr_read, which executes IN the matrix pipe behind the MFMAs.
    // Operands: av comes from a ds_read and c's first value from a global load (hipcc's own s_waitcnt cover both); xf was
    // written by VALU long before; the same accumulator chains with 0 wait states.
    // HAZARD the compiler cannot see (the producer is an asm statement): an MFMA result may be read by a VALU instruction only
    // 12+ wait states after the MFMA.  The tiles are read by the GELU of the NEXT step, behind a barrier and two MFMA slots.
    // (Found the hard way in the one-wave form: 1 % error when the first read came right behind the barrier.)
    auto mfma_p = [](f32x16& c, const f16x8& av, const f16x8& bv) __attribute__((always_inline)) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    };

    // ---- the stages of a step.  k & 1 = PAR: W1 slot of A(k), W2 slot + HF block of C(k - 2); B(k - 1) writes HF block PAR ^ 1 ----
    // C(k - 2): O^T += W2 . H^T for this wave's 5 out tiles: 10 MFMAs, u-major (dependent MFMAs five slots apart)
    auto c_frag = [&](auto par, int i) __attribute__((always_inline)) -> f16x8 {
        constexpr int P = decltype(par)::v;
        return *(const f16x8*)(wl + OFF_W2 + P * W2_SLOT + half * (2 * FF_NJW * 1024) + (2 * (i % FF_NJW) + i / FF_NJW) * 1024);
    };
    auto hf_read = [&](auto par, f16x8 (&hb)[2]) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::v;
        const char* h = wl + OFF_HF + (P * 4 + pair) * 2048;
        hb[0] = *(const f16x8*)h;
        hb[1] = *(const f16x8*)(h + 1024);
    };
    // owner of chunk k: [C(k - 2)'s 10 MFMAs,] then A(k)'s 40; every fragment read FF_LOOK MFMAs ahead of its use
    auto path_owner = [&](auto par, auto do_c, const int chunk_a) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::v;
        constexpr int NC = decltype(do_c)::v ? 2 * FF_NJW : 0, NS = NC + 2 * FF_KS;
        f16x8 hb[2];
        if constexpr (NC > 0) hf_read(par, hb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bv = *(const f32x4*)(b1l + 32 * chunk_a + 8 * q);
            const f32x4 bg = *(const f32x4*)(b1l + FF_H + 32 * chunk_a + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) { pv[4 * q + e] = bv[e]; pg[4 * q + e] = bg[e]; }
        }
        const char* w1 = wl + OFF_W1 + P * W1_SLOT;
        auto frag = [&](int i) __attribute__((always_inline)) -> f16x8 {
            if (i < NC) return c_frag(par, i);
            const int t = i - NC;                                  // k-step t / 2, value / gate tile t % 2
            return *(const f16x8*)(w1 + ((t & 1) * FF_KS + (t >> 1)) * 1024);
        };
        f16x8 ring[FF_LOOK];
#pragma unroll
        for (int i = 0; i < FF_LOOK; ++i) ring[i] = frag(i);
        static_for<0, NS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::v;
            const f16x8 af = ring[i % FF_LOOK];
#ifndef FF_T_NOREAD
            if constexpr (i + FF_LOOK < NS) ring[i % FF_LOOK] = frag(i + FF_LOOK);
#endif
            if constexpr (i < NC) {
                O[i % FF_NJW] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, hb[i / FF_NJW], O[i % FF_NJW], 0, 0, 0);
            } else {
                constexpr int t = i - NC;
                if constexpr (t & 1) mfma_p(pg, af, xf[t >> 1]);
                else mfma_p(pv, af, xf[t >> 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // the other wave: GELU of the chunk it owned one step ago (x * Phi(x), Phi as in gelu_phi_f of common.h, on float2),
    // H^T -> LDS, [and C(k - 2)'s 10 MFMAs in between]
    auto path_helper = [&](auto par, auto do_b, auto do_c) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::v;
        constexpr bool B = decltype(do_b)::v != 0;
        constexpr int NC = decltype(do_c)::v ? 2 * FF_NJW : 0;
        f16x8 hb[2], hfo[2];
        if constexpr (NC > 0) hf_read(par, hb);
        f32x2 G[8], W[8], U[8], Q[8];
        auto gelu_pair = [&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::v;
            constexpr float K[9] = {5.626766414e-11f, -5.371867839e-09f, 2.268295702e-07f, -5.646214049e-06f, 9.359061369e-05f,
                                    -1.109400182e-03f, 9.818118997e-03f, -6.634692103e-02f, 3.989031613e-01f};
            G[d] = f32x2{pg[2 * d], pg[2 * d + 1]};
#ifdef FF_T_CHEAPGELU                          // timing-only: value * gate
            Q[d] = G[d];
#else
            W[d] = f32x2{__builtin_amdgcn_fmed3f(G[d][0], -4.2426405f, 4.2426405f), __builtin_amdgcn_fmed3f(G[d][1], -4.2426405f, 4.2426405f)};
            U[d] = W[d] * W[d];
            Q[d] = __builtin_elementwise_fma(f32x2{K[0], K[0]}, U[d], f32x2{K[1], K[1]});
#pragma unroll
            for (int t = 2; t < 9; ++t) Q[d] = __builtin_elementwise_fma(Q[d], U[d], f32x2{K[t], K[t]});
            Q[d] = __builtin_elementwise_fma(W[d], Q[d], f32x2{0.5f, 0.5f});   // Phi
            Q[d] = G[d] * Q[d];                                                 // gelu(gate)
#endif
            Q[d] = f32x2{pv[2 * d], pv[2 * d + 1]} * Q[d];
            hfo[d >> 2][2 * (d & 3)] = (f16)Q[d][0];
            hfo[d >> 2][2 * (d & 3) + 1] = (f16)Q[d][1];
        };
        if constexpr (NC > 0) {
            f16x8 ring[FF_LOOK];
#pragma unroll
            for (int i = 0; i < FF_LOOK; ++i) ring[i] = c_frag(par, i);
            static_for<0, NC>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::v;
                const f16x8 af = ring[i % FF_LOOK];
#ifndef FF_T_NOREAD
                if constexpr (i + FF_LOOK < NC) ring[i % FF_LOOK] = c_frag(par, i + FF_LOOK);
#endif
                O[i % FF_NJW] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, hb[i / FF_NJW], O[i % FF_NJW], 0, 0, 0);
                if constexpr (B && i >= 2) gelu_pair(IC<i - 2>{});             // 8 pairs behind MFMAs 2 .. 9
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if constexpr (B) {
            static_for<0, 8>([&](auto dc) __attribute__((always_inline)) { gelu_pair(dc); });
        }
        if constexpr (B) {
            char* h = smem_ff + OFF_HF + ((P ^ 1) * 4 + pair) * 2048 + lane * 16;
            *(f16x8*)h = hfo[0];
            *(f16x8*)(h + 1024) = hfo[1];
        }
    };
    // step k: the owner of chunk k (half == k & 1) runs A(k) [+ C]; the owner of chunk k - 1 (the other wave) runs B(k - 1) [+ C]
    auto step = [&](auto do_a, auto do_b, auto do_c, auto par, const int chunk_a) __attribute__((always_inline)) {
        constexpr bool A = decltype(do_a)::v != 0;
        constexpr int P = decltype(par)::v;
        if (half == P) {
            if constexpr (A) path_owner(par, do_c, chunk_a);
            else path_helper(par, IC<0>{}, do_c);                  // (drain steps: no chunk left to own)
        } else {
            path_helper(par, do_b, do_c);
        }
    };

    const f16* xg = (const f16*)a.x;
    const f16* r2g = (const f16*)a.r2;
    f16* og = (f16*)a.out;
    const float s_acc = a.s_acc, s1 = a.s1, s2 = a.s2;

    dma_w1(0, 0);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + pair * 32 + l31;
        const bool rok = m < a.M;
        const int mr = rok ? m : 0;
        const f16* xrow = xg + (size_t)mr * a.ldx;
        const float* prow = nullptr;
        if constexpr (POS) prow = a.pos + (size_t)((mr / a.HW) % a.T) * FF_C;
        // ---- token row -> registers, LayerNorm in fp32: three passes over the fp16 fragments (mean, centred squares, normalise),
        //      x' = x + pos re-formed in each (pos rows are L1 hits) -- no fp32 copy of the row: the 256 registers of a wave are
        //      spoken for (X 80, O 80, P 32) and a 160-register temporary here made hipcc spill X fragments INSIDE the chunk loop.
        //      Gain / bias of the norm live in W1 / b1. ----
        {
#pragma unroll
            for (int s = 0; s < FF_KS; ++s) xf[s] = *(const f16x8*)(xrow + 16 * s + 8 * lh);
            auto xval = [&](int s, int e, const f32x4& p0, const f32x4& p1) __attribute__((always_inline)) -> float {
                return (float)xf[s][e] + (POS ? (e < 4 ? p0[e & 3] : p1[e & 3]) : 0.f);
            };
            auto pos_of = [&](int s, f32x4& p0, f32x4& p1) __attribute__((always_inline)) {
                if constexpr (POS) {
                    p0 = *(const f32x4*)(prow + 16 * s + 8 * lh);
                    p1 = *(const f32x4*)(prow + 16 * s + 8 * lh + 4);
                }
            };
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < FF_KS; ++s) {
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                pos_of(s, p0, p1);
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += xval(s, e, p0, p1);
            }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / FF_C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < FF_KS; ++s) {
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                pos_of(s, p0, p1);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xval(s, e, p0, p1) - mean; sq = fmaf(d, d, sq); }
            }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / FF_C) + a.eps);
#pragma unroll
            for (int s = 0; s < FF_KS; ++s) {
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                pos_of(s, p0, p1);
                f16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (f16)((xval(s, e, p0, p1) - mean) * rstd);
                xf[s] = y;
            }
        }
        // ---- O^T starts at b2: register r of tile jj <-> out col 32 (5 half + jj) + 8 (r >> 2) + 4 lh + (r & 3) ----
#pragma unroll
        for (int jj = 0; jj < FF_NJW; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *(const f32x4*)(a.b2 + 32 * (FF_NJW * half + jj) + 8 * q + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) O[jj][4 * q + e] = b[e];
            }

        // ---- 42 steps, one barrier each.  W1(k) lives in W1 slot k & 1, W2(k) in W2 slot k & 1, H^T(k) in HF block k & 1.  Step k
        //      (after its barrier: everybody has finished step k - 1 and its DMA has landed) fetches W1(k + 1) -> slot (k + 1) & 1
        //      (A(k - 1) read it last) and W2(k - 1) -> slot (k + 1) & 1 (C(k - 3) read it last; C(k - 1) reads it in step k + 1).
        //      Step 39 fetches the NEXT tile's W1(0) into slot 0, where its step 0 reads it. ----
        ff_barrier();                                              // step 0: A(0)
        dma_w1(1, 1);
        step(IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}, 0);
        ff_barrier();                                              // step 1: A(1) | B(0)
        dma_w1(2, 0);
        dma_w2(0, 0);
        step(IC<1>{}, IC<1>{}, IC<0>{}, IC<1>{}, 1);
        for (int k = 2; k < FF_NCHUNK; k += 2) {
            ff_barrier();                                          // step k (even): A(k) | B(k - 1), C(k - 2)
            dma_w1(k + 1, 1);
            dma_w2(k - 1, 1);
            step(IC<1>{}, IC<1>{}, IC<1>{}, IC<0>{}, k);
            ff_barrier();                                          // step k + 1 (odd)
            dma_w1(k + 2 < FF_NCHUNK ? k + 2 : 0, 0);
            dma_w2(k, 0);
            step(IC<1>{}, IC<1>{}, IC<1>{}, IC<1>{}, k + 1);
        }
        // this wave's 10 fragments of the raw token row once more (the registers held the normalised row until A(39)): the residual
        // of the epilogue, landing under the two drain steps; the AlphaBlender's second residual likewise
        f16x8 xr[2 * FF_NJW], rr[R2 ? 2 * FF_NJW : 1];
        const f16* r2row = R2 ? r2g + (size_t)mr * a.ldr2 : nullptr;
        const int ncol0 = 32 * FF_NJW * half + 8 * lh;             // this lane's first output column
#pragma unroll
        for (int p = 0; p < 2 * FF_NJW; ++p) {
            xr[p] = *(const f16x8*)(xrow + ncol0 + 16 * p);
            if constexpr (R2) rr[p] = *(const f16x8*)(r2row + ncol0 + 16 * p);
        }
        ff_barrier();                                              // step 40: B(39), C(38)
        dma_w2(FF_NCHUNK - 1, 1);
        step(IC<0>{}, IC<1>{}, IC<1>{}, IC<0>{}, 0);
        ff_barrier();                                              // step 41: C(39)
        step(IC<0>{}, IC<0>{}, IC<1>{}, IC<1>{}, 0);

        // ---- epilogue: out = f16( f16(s_acc * O) + s1 * x' + s2 * r2 ), x' = x (+ pos).  The 10 pieces (8 columns each) of this
        //      wave are formed first and stored under ONE row test: a branch per piece would serialise loads behind vmcnt(0) ----
        f16* orow = og + (size_t)mr * a.ldo + ncol0;
        float lsum = 0.f;
        f16x8 op[2 * FF_NJW];
#pragma unroll
        for (int jj = 0; jj < FF_NJW; ++jj) {
            f16x4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[q][e] = (f16)(s_acc * O[jj][4 * q + e]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // groups 2 k and 2 k + 1 (columns 16 k + 4 lh + e and 16 k + 8 + 4 lh + e of the tile): after the swap lanes 0-31 hold
                // columns 16 k .. 16 k + 7, lanes 32-63 columns 16 k + 8 .. 16 k + 15 of their row: piece 2 jj + k of this wave
                const u32x2 a2 = __builtin_bit_cast(u32x2, g[2 * k]), b2 = __builtin_bit_cast(u32x2, g[2 * k + 1]);
                const auto rx = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
                const u32x4 vv = {rx[0], ry[0], rx[1], ry[1]};
                const f16x8 v16 = __builtin_bit_cast(f16x8, vv);
                const int p = 2 * jj + k;
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                if constexpr (POS) { p0 = *(const f32x4*)(prow + ncol0 + 16 * p); p1 = *(const f32x4*)(prow + ncol0 + 16 * p + 4); }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)v16[e] + s1 * ((float)xr[p][e] + (e < 4 ? p0[e & 3] : p1[e & 3]));
                    if constexpr (R2) v += s2 * (float)rr[p][e];
                    o[e] = (f16)v;
                    if constexpr (LNOUT) { const float vf = (float)o[e]; O[jj][8 * k + e] = vf; lsum += vf; }
                }
                op[p] = o;
            }
        }
        if (rok) {
#pragma unroll
            for (int p = 0; p < 2 * FF_NJW; ++p) *(f16x8*)(orow + 16 * p) = op[p];
        }
        if constexpr (LNOUT) {
            // LayerNorm of the row just written (the norm in front of the next projection): O[jj][8 k + e] = this lane's outputs.
            // A row is spread over two lanes of each wave of the pair: lane halves by shuffle, the two waves through LDS (two
            // passes, like the stand-alone kernel: mean, then centred squares)
            float* lx = (float*)(smem_ff + OFF_LN);
            lsum += __shfl_xor(lsum, 32, 64);
            if (lh == 0) lx[(half * 4 + pair) * 32 + l31] = lsum;
            ff_barrier();
            const float mean = (lsum + lx[((half ^ 1) * 4 + pair) * 32 + l31]) * (1.0f / FF_C);
            float sq = 0.f;
#pragma unroll
            for (int jj = 0; jj < FF_NJW; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = O[jj][r] - mean; sq = fmaf(d, d, sq); }
            sq += __shfl_xor(sq, 32, 64);
            ff_barrier();
            if (lh == 0) lx[(half * 4 + pair) * 32 + l31] = sq;
            ff_barrier();
            const float rstd = rsqrtf((sq + lx[((half ^ 1) * 4 + pair) * 32 + l31]) * (1.0f / FF_C) + a.ln_eps);
            f16* lrow = (f16*)a.out_ln + (size_t)mr * a.ldoln + ncol0;
#pragma unroll
            for (int p = 0; p < 2 * FF_NJW; ++p) {
                const int n = ncol0 + 16 * p;
                const f32x4 g0 = *(const f32x4*)(a.ln_gamma + n), g1 = *(const f32x4*)(a.ln_gamma + n + 4);
                const f32x4 c0 = *(const f32x4*)(a.ln_beta + n), c1 = *(const f32x4*)(a.ln_beta + n + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    op[p][e] = (f16)fmaf((O[p >> 1][8 * (p & 1) + e] - mean) * rstd, e < 4 ? g0[e & 3] : g1[e & 3], e < 4 ? c0[e & 3] : c1[e & 3]);
            }
            if (rok) {
#pragma unroll
                for (int p = 0; p < 2 * FF_NJW; ++p) *(f16x8*)(lrow + 16 * p) = op[p];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the last step's look-ahead DMA must not outlive the wave
}

typedef void (*ff320_kern_t)(const mofa_ff320_args, const int);
template <bool POS, bool R2, bool LNOUT>
ff320_kern_t ff320_pick() { return ff320_kernel<POS, R2, LNOUT>; }

}  // namespace

extern "C" int mofa_ff320_f16(const mofa_ff320_args* a, mofa_stream_t stream) {
    if (!a || !a->x || !a->w1p || !a->b1 || !a->w2p || !a->b2 || !a->out || a->M <= 0) return MOFA_EINVAL;
    if ((a->ldx & 7) || (a->ldo & 7) || a->ldx < FF_C || a->ldo < FF_C || (((size_t)a->x) & 15) || (((size_t)a->out) & 15) ||
        (((size_t)a->w1p) & 15) || (((size_t)a->w2p) & 15) || (((size_t)a->b1) & 15) || (((size_t)a->b2) & 15))
        return MOFA_EINVAL;
    if (a->pos && (a->HW <= 0 || a->T <= 0 || (((size_t)a->pos) & 15))) return MOFA_EINVAL;
    if (a->r2 && ((a->ldr2 & 7) || a->ldr2 < FF_C || (((size_t)a->r2) & 15))) return MOFA_EINVAL;
    if (a->out_ln && (!a->ln_gamma || !a->ln_beta || (a->ldoln & 7) || a->ldoln < FF_C || (((size_t)a->out_ln) & 15) ||
                      (((size_t)a->ln_gamma) & 15) || (((size_t)a->ln_beta) & 15)))
        return MOFA_EINVAL;
    static const ff320_kern_t kerns[8] = {
        ff320_pick<false, false, false>(), ff320_pick<true, false, false>(), ff320_pick<false, true, false>(), ff320_pick<true, true, false>(),
        ff320_pick<false, false, true>(),  ff320_pick<true, false, true>(),  ff320_pick<false, true, true>(),  ff320_pick<true, true, true>()};
    static const int n_cu = [] {
        int dev = 0, cus = 0;
        for (int i = 0; i < 8; ++i)
            (void)hipFuncSetAttribute((const void*)kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_BYTES);
        return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
                cus > 0) ? cus : 256;
    }();
    const int ntiles = (a->M + 127) / 128;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const ff320_kern_t k = kerns[(a->pos ? 1 : 0) | (a->r2 ? 2 : 0) | (a->out_ln ? 4 : 0)];
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), FF_LDS_BYTES, (hipStream_t)stream, *a, ntiles);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
