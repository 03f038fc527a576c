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
// Shape of the computation -- everything TRANSPOSED, so that nothing is exchanged between lanes or through LDS except weights:
//   * a wave owns 32 token rows for the whole kernel tile; a workgroup = 4 waves (one per SIMD, 512 registers each) = 128 rows.
//   * X^T is the B operand of GEMM 1:  P^T[proj row, token] = W1[proj row, :] . X^T.  Lane (token l31, k half lh) holds
//     X[token][16 s + 8 lh .. + 7] for the 20 k-steps s: 80 registers loaded ONCE per tile straight from global memory, and a
//     token row lives in exactly two lanes -- the LayerNorm is 160 values per lane plus one cross-half exchange, done in
//     registers (gain / bias of the norm are folded into W1 / b1 at load time: mofa_video_amd/weights.py::pack_ff320).
//   * the hidden axis is walked in 40 chunks of 32: per chunk one value tile and one gate tile of P^T (2 x 20 MFMAs
//     32x32x16), bias as the accumulators' initial value; H^T = value * gelu(gate) is formed lane-locally and -- rounded to
//     fp16 -- IS the B operand of GEMM 2 (O^T[out col, token] += W2[out col, hidden chunk] . H^T, 10 tiles x 2 k-steps): the
//     k-slot -> hidden permutation of that MFMA is absorbed in the packed order of W2 (same trick as the P operand of
//     attention.hip).  O^T is 10 accumulator tiles = 160 registers.
//   * the ONLY LDS traffic is weights: per chunk 40 KB of W1 + 20 KB of W2 arrive by LDS-DMA (buffer_load ... lds) as
//     lane-linear 1 KB blocks -- the packed global image IS the LDS image, every fragment read is a conflict-free
//     ds_read_b128 at base + immediate.  The LDS is all ring: three W1 chunk images (fetched TWO steps ahead: an L2 -> LDS
//     piece needs longer to land than one step's 60 MFMAs take) + two W2 images = 160 KB; one barrier per chunk with a
//     counted vmcnt.  Weights are 2.4 MB per layer: L2 resident for every workgroup of the launch.
//   * software pipeline, three stages one chunk apart: GEMM 1 of chunk k, GELU of chunk k - 1, GEMM 2 of chunk k - 2 are
//     independent streams of one step (the GELU's VALU under 60 MFMAs); filled / drained at tile boundaries (42 steps per tile).
//   * epilogue: fp16 rounding of s_acc * (O + b2) before the residual add like every implicit-GEMM tile (include/mofa_hip.h),
//     v_permlane32_swap gives a lane 8 consecutive output columns of its row: 16-byte residual loads and stores; optional
//     second output LayerNorm(out) (the norm in front of the NEXT projection: the lane pair holds the whole output row).
#include "common.h"

namespace {

constexpr int FF_C = 320, FF_H = 1280, FF_NCHUNK = 40, FF_KS = 20, FF_NJ = 10;
constexpr int W1_SLOT = 2 * FF_KS * 1024;      // 40 KB: [value | gate tile][k-step][64 lanes x 16 B]
constexpr int W2_SLOT = FF_NJ * 2 * 1024;      // 20 KB: [out tile][k-step u][64 lanes x 16 B]
constexpr int OFF_W1 = 0, OFF_W2 = 3 * W1_SLOT;               // W1: ring of THREE chunk images (DMA two steps ahead), W2: two
constexpr int FF_LDS_BYTES = OFF_W2 + 2 * W2_SLOT;            // 163 840 = the whole LDS of a CU

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IC { static constexpr int v = V; };
#ifndef FF_ILP
#define FF_ILP 2                               // GELU pairs (2 hidden columns each) whose micro-operations alternate
#endif
#ifndef FF_LOOK_D
#define FF_LOOK_D 6                            // MFMA slots a weight fragment is read ahead of its use (even)
#endif
#ifndef FF_BDELAY_D
#define FF_BDELAY_D 3                          // MFMA slots into a step before the GELU reads the previous step's tiles
#endif
// compile-time loop: f(IC<0>{}), f(IC<1>{}), ... -- the slot schedule below needs every index as a constant expression
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}

#define FF_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// step barrier: this wave's LDS-DMA pieces up to the N youngest have landed (vmcnt counts in issue order), its fragment reads are
// back, then everybody is here.  N = the W1 pieces of the chunk two steps ahead, which may stay in flight across the barrier.
#ifdef FF_T_NOB1
constexpr int FF_PRE = 0;
#else
constexpr int FF_PRE = 8;                      // b1 loads of the next step, issued at the end of a step (see step())
#endif
template <int N>
__device__ __forceinline__ void ff_barrier() {
#if defined(FF_DBG_VM0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#elif defined(FF_T_NOBAR)                     // timing-only build (wrong results): no step barrier at all
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

template <bool POS, bool R2, bool LNOUT>
__global__ __launch_bounds__(256, 1) void ff320_kernel(const mofa_ff320_args a, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_ff[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1p, 0, 2u * FF_H * FF_C * 2u, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2p, 0, (unsigned)FF_C * FF_H * 2u, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    // a wave fetches blocks wave, wave + 4, ... of a chunk image: 10 of W1's 40, 5 of W2's 20 (1 KB each)
    auto dma_w1 = [&](int chunk, int slot) __attribute__((always_inline)) {
        char* dst = smem_ff + OFF_W1 + slot * W1_SLOT + wave * 10240;
        const int src = chunk * W1_SLOT + wave * 10240;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, FF_LDS_PTR(dst + i * 1024), 16, voff, src + i * 1024, 0, 0);
    };

    const char* wl = smem_ff + lane * 16;                          // this lane's 16 bytes of every 1 KB block
    const float* b1l = a.b1 + 4 * lh;                             // (global memory, L1 / L2 hits: the LDS is all weight ring)

    // Three pipeline stages per hidden chunk c, one chunk apart, so that every step k of a tile is
    //     C(k - 2): O^T += W2(k - 2) . H^T(k - 2)     20 MFMAs, operands hf[k & 1]
    //     A(k)    : P^T(k) = b1 + W1(k) . X^T          40 MFMAs into p[k & 1]
    //     B(k - 1): H^T(k - 1) = value * gelu(gate)    VALU on p[(k - 1) & 1] -> hf[(k - 1) & 1]
    // three INDEPENDENT instruction streams.  With ONE wave per SIMD nobody else covers a latency, so the step is laid out by
    // hand as 60 (40, 20) MFMA slots with sched_barrier(0) between them: slot i issues the weight-fragment read of slot
    // i + FF_LOOK, its MFMA, its share of the GELU micro-operations (two hidden columns at a time on float2 -- 14 per pair, 112 per
    // chunk, i.e. about two per MFMA gap, which hides about five single-issue instructions: MI355X_MICROARCH.md) and, every
    // fourth slot, one 1 KB piece of the next chunk's weight DMA.
    // Register r of p lane (l31, lh) <-> hidden 32 c + 8 (r >> 2) + 4 lh + (r & 3).
    f16x8 xf[FF_KS];
    f32x16 O[FF_NJ];
    f32x16 pv[2], pg[2];
    f16x8 hf[2][2];
    constexpr int FF_LOOK = FF_LOOK_D, FF_BDELAY = FF_BDELAY_D, FF_GOPS = 15;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto step = [&](auto do_a, auto do_b, auto do_c, auto par, const int chunk_a, const int slot_a, auto d1, const int chunk_w1,
                    const int slot_w1, auto d2, const int chunk_w2, auto own_b1, auto pre_b1) __attribute__((always_inline)) {
        constexpr bool A = decltype(do_a)::v != 0, B = decltype(do_b)::v != 0, Cc = decltype(do_c)::v != 0;
        // DMA W1(chunk_w1) -> W1 slot slot_w1, W2(chunk_w2) -> W2 slot P ^ 1; A reads W1 slot slot_a
        constexpr bool D1 = decltype(d1)::v != 0, D2 = decltype(d2)::v != 0;
        constexpr int P = decltype(par)::v;                        // k & 1: A -> p[P], B: p[P ^ 1] -> hf[P ^ 1], C: hf[P], W2 slot P
        constexpr int NC = Cc ? 2 * FF_NJ : 0, NA = A ? 2 * FF_KS : 0, NS = NC + NA;
#ifdef FF_T_NODMA                              // timing-only builds (wrong results): no weight DMA / no GELU arithmetic / no fragment reads
        constexpr int NPIECE = 0;
#else
        constexpr int NPIECE = (D1 ? 10 : 0) + (D2 ? 5 : 0);
#endif
#ifdef FF_T_NOGELU_INVALID_DEAD_MFMA
        constexpr int NOPS = 0;
#else
        constexpr int NOPS = B ? 8 * FF_GOPS : 0;
#endif
        static_assert(NS >= 20, "every step has MFMA slots");
        const char* w1 = wl + OFF_W1 + slot_a * W1_SLOT;
        const char* w2 = wl + OFF_W2 + P * W2_SLOT;
        // this wave's pieces: a CONTIGUOUS run of blocks (W1: 10 wave .. + 9, W2: 5 wave .. + 4), four pieces per LDS base / scalar
        // offset, the piece inside a group picked by the instruction's immediate offset (it advances both addresses:
        // tools/lds_dma_range.hip) -- a third of the s_mov m0 / s_add per piece that one base per piece costs
        char* dst1 = smem_ff + OFF_W1 + slot_w1 * W1_SLOT + wave * 10240;
        char* dst2 = smem_ff + OFF_W2 + (P ^ 1) * W2_SLOT + wave * 5120;
        const int src1 = chunk_w1 * W1_SLOT + wave * 10240, src2 = chunk_w2 * W2_SLOT + wave * 5120;
        // b1 = the first value of A's accumulators, by global loads (the LDS is all weight ring).  A step loads the b1 of the NEXT
        // step's chunk at its END (pre_b1: the registers p[P ^ 1] are free once its GELU has read them), in front of the barrier:
        // loaded at the start of their own step they had 20 MFMA slots to arrive and hipcc's wait in front of A's first MFMA stalled
        // on them (- 3.5...4.8 % of the launch without the loads, profiles/r06_ff320_anatomy.log); only a tile's step 0 loads its own.
        auto load_b1 = [&](auto pc, const int chunk) __attribute__((always_inline)) {
            constexpr int PP = decltype(pc)::v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#ifdef FF_T_NOB1                               // timing-only: no bias loads at all
                const f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = {1.f, 1.f, 1.f, 1.f};
#else
                const f32x4 bv = *(const f32x4*)(b1l + 32 * chunk + 8 * q);
                const f32x4 bg = *(const f32x4*)(b1l + FF_H + 32 * chunk + 8 * q);
#endif
#pragma unroll
                for (int e = 0; e < 4; ++e) { pv[PP][4 * q + e] = bv[e]; pg[PP][4 * q + e] = bg[e]; }
            }
        };
        if constexpr (A && decltype(own_b1)::v != 0) load_b1(IC<P>{}, chunk_a);
        // slot i < NC: C's MFMA on O[i % 10] with k-step u = i / 10 (dependent MFMAs ten slots apart), block 2 j + u of the W2
        // slot; slot NC + t: A's MFMA (k-step t / 2, value / gate tile t % 2), block (t % 2) * 20 + t / 2 of the W1 slot
        auto frag = [&](int i) __attribute__((always_inline)) -> f16x8 {
            if (i < NC) return *(const f16x8*)(w2 + (2 * (i % FF_NJ) + i / FF_NJ) * 1024);
            const int t = i - NC;
            return *(const f16x8*)(w1 + ((t & 1) * FF_KS + (t >> 1)) * 1024);
        };
        constexpr int FF_RING = FF_LOOK + 2;                       // fragment j sits in ring[j % FF_RING]
        f16x8 ring[FF_RING];
#pragma unroll
        for (int i = 0; i < FF_LOOK; ++i) ring[i] = frag(i);
        // GELU micro-operations of pair d (hidden registers 2 d, 2 d + 1 of p[P ^ 1]): x * Phi(x), Phi as in gelu_phi_f (common.h);
        // FF_GOPS = 15 per pair on float2 (hipcc emits scalar v_fma_f32 pairs for them, which is what is wanted: packed fp32 measured
        // SLOWER beside the MFMAs, profiles/r06_ff320_anatomy.log), FF_ILP pairs alternating.
        // The P^T tiles live in VGPRs, written by MFMAs issued through asm (below): hipcc keeps every accumulator of a
        // 512-register kernel in the AGPR half, and v_accvgpr_read executes IN the matrix pipe, in order with the MFMAs -- 32 reads
        // per step cost 56 % of the kernel (1413 -> 599 us with the GELU fed from ordinary registers, same log).
        // HAZARD the compiler cannot see (the producer is an asm statement): an MFMA result may be read by a VALU instruction only
        // 12+ wait states after the MFMA.  The tiles read here were finished by the last MFMAs of the PREVIOUS step; the GELU starts
        // FF_BDELAY slots (MFMA issues) into the step.
        f32x2 G[8], W[8], U[8], Q[8];
        auto gelu_op = [&](auto dc, auto oc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::v, op = decltype(oc)::v;
            constexpr float K[9] = {5.626766414e-11f, -5.371867839e-09f, 2.268295702e-07f, -5.646214049e-06f, 9.359061369e-05f,
                                    -1.109400182e-03f, 9.818118997e-03f, -6.634692103e-02f, 3.989031613e-01f};
#ifdef FF_T_CHEAPGELU                          // timing-only: value * gate instead of value * gelu(gate): 3 of the 15 micro-operations, data flow intact
            if constexpr (op == 0) Q[d] = f32x2{pg[P ^ 1][2 * d], pg[P ^ 1][2 * d + 1]} * f32x2{pv[P ^ 1][2 * d], pv[P ^ 1][2 * d + 1]};
            else if constexpr (op == 13) hf[P ^ 1][d >> 2][2 * (d & 3)] = (f16)Q[d][0];
            else if constexpr (op == 14) hf[P ^ 1][d >> 2][2 * (d & 3) + 1] = (f16)Q[d][1];
            return;
#endif
            if constexpr (op == 0) {
                G[d] = f32x2{pg[P ^ 1][2 * d], pg[P ^ 1][2 * d + 1]};
                W[d] = f32x2{__builtin_amdgcn_fmed3f(G[d][0], -4.2426405f, 4.2426405f), __builtin_amdgcn_fmed3f(G[d][1], -4.2426405f, 4.2426405f)};
            } else if constexpr (op == 1) {
                U[d] = W[d] * W[d];
            } else if constexpr (op == 2) {
                Q[d] = __builtin_elementwise_fma(f32x2{K[0], K[0]}, U[d], f32x2{K[1], K[1]});
            } else if constexpr (op <= 9) {
                Q[d] = __builtin_elementwise_fma(Q[d], U[d], f32x2{K[op - 1], K[op - 1]});
            } else if constexpr (op == 10) {
                Q[d] = __builtin_elementwise_fma(W[d], Q[d], f32x2{0.5f, 0.5f});   // Phi
            } else if constexpr (op == 11) {
                Q[d] = G[d] * Q[d];                                 // gelu(gate)
            } else if constexpr (op == 12) {
                Q[d] = f32x2{pv[P ^ 1][2 * d], pv[P ^ 1][2 * d + 1]} * Q[d];
            } else if constexpr (op == 13) {
                hf[P ^ 1][d >> 2][2 * (d & 3)] = (f16)Q[d][0];
            } else {
                hf[P ^ 1][d >> 2][2 * (d & 3) + 1] = (f16)Q[d][1];
            }
        };
        // A's MFMAs are issued through asm with the accumulators in VGPRs ("+v", see above).  Operands: the fragments come from
        // ds_reads and the accumulators' first values from global loads (hipcc's own s_waitcnt cover both); xf was written by VALU
        // long before; the same accumulator chains with 0 wait states.
        // this wave's pieces of the step: W2's five FIRST (needed at the next barrier: vmcnt counts in issue order), then W1's ten
        auto dma_piece = [&](auto pcc) __attribute__((always_inline)) {
            constexpr int pc = decltype(pcc)::v, n2 = D2 ? 5 : 0;
            if constexpr (pc < n2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, FF_LDS_PTR(dst2 + (pc >> 2) * 4096), 16, voff, src2 + (pc >> 2) * 4096,
                                                         (pc & 3) * 1024, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, FF_LDS_PTR(dst1 + ((pc - n2) >> 2) * 4096), 16, voff,
                                                         src1 + ((pc - n2) >> 2) * 4096, ((pc - n2) & 3) * 1024, 0);
        };
        static_for<0, NS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::v;
            const f16x8 af = ring[i % FF_RING];
#ifndef FF_T_NOREAD
            // reads are issued two at a time on even slots, the LATER fragment first: LDS returns in order, so the s_waitcnt in
            // front of the MFMA that needs the earlier (younger) one covers the next MFMA's too -- one wait per two MFMAs
            if constexpr ((i & 1) == 0) {
                if constexpr (i + FF_LOOK + 1 < NS) ring[(i + FF_LOOK + 1) % FF_RING] = frag(i + FF_LOOK + 1);
                if constexpr (i + FF_LOOK < NS) ring[(i + FF_LOOK) % FF_RING] = frag(i + FF_LOOK);
            }
#endif
            if constexpr ((i & 1) == 0 && (i >> 1) < NPIECE) dma_piece(IC<(i >> 1)>{});   // one piece every other slot, from slot 0
            if constexpr (i < NC) {
                O[i % FF_NJ] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, hf[P][i / FF_NJ], O[i % FF_NJ], 0, 0, 0);
            } else {
                constexpr int t = i - NC;
                // the value and the gate MFMA of a k-step as ONE asm statement: hipcc puts a (conservative) s_waitcnt in front of
                // every asm statement that reads a freshly loaded fragment -- one per k-step instead of two (+ 1-2 % on the kernel)
                if constexpr ((t & 1) == 0)
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %4, %1"
                                 : "+v"(pv[P]), "+v"(pg[P]) : "v"(af), "v"(ring[(i + 1) % FF_RING]), "v"(xf[t >> 1]));
            }
            constexpr int ib = i < FF_BDELAY ? 0 : i - FF_BDELAY;
            constexpr int m0 = (ib * NOPS) / (NS - FF_BDELAY), m1 = i < FF_BDELAY ? 0 : ((ib + 1) * NOPS) / (NS - FF_BDELAY);
            static_for<m0, m1>([&](auto mc) __attribute__((always_inline)) {
                constexpr int mo = decltype(mc)::v;
                // two pairs in flight, their micro-operations alternating: consecutive instructions are independent (no VALU
                // dependency stall, and hipcc puts no s_nop between an asm statement and a following one that does not read it)
                gelu_op(IC<FF_ILP * (mo / (FF_ILP * FF_GOPS)) + (mo % FF_ILP)>{}, IC<(mo % (FF_ILP * FF_GOPS)) / FF_ILP>{});
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        // (a step with few slots cannot place all its DMA pieces: the rest go here)
        static_for<(NS + 1) / 2, NPIECE>([&](auto pcc) __attribute__((always_inline)) { dma_piece(pcc); });
        if constexpr (decltype(pre_b1)::v != 0) load_b1(IC<P ^ 1>{}, chunk_a + 1);
    };

    const f16* xg = (const f16*)a.x;
    const f16* r2g = (const f16*)a.r2;
    f16* og = (f16*)a.out;
    const float s_acc = a.s_acc, s1 = a.s1, s2 = a.s2;

    dma_w1(0, 0);
    dma_w1(1, 1);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m = tile * 128 + wave * 32 + l31;
        const bool rok = m < a.M;
        const int mr = rok ? m : 0;
        const f16* xrow = xg + (size_t)mr * a.ldx;
        const float* prow = nullptr;
        if constexpr (POS) prow = a.pos + (size_t)((mr / a.HW) % a.T) * FF_C;
        // ---- token row -> registers, LayerNorm in fp32: three passes over the fp16 fragments (mean, centred squares, normalise), x' =
        //      x + pos re-formed in each (pos rows are L1 hits) -- no fp32 copy of the row (160 registers that made hipcc spill).
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
        // ---- O^T starts at b2: register r of tile j <-> out col 32 j + 8 (r >> 2) + 4 lh + (r & 3) ----
#pragma unroll
        for (int j = 0; j < FF_NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *(const f32x4*)(a.b2 + 32 * j + 8 * q + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) O[j][4 * q + e] = b[e];
            }

        // ---- 42 steps, one barrier each.  Weight rings: W1(k) lives in W1 slot k % 3, W2(k) in W2 slot k & 1.  Step k (after its
        //      barrier: every wave has finished step k - 1) issues W2(k - 1) -> slot (k + 1) & 1 (C(k - 3) read it last; C(k - 1)
        //      reads it in step k + 1) and then W1(k + 2) -> slot (k + 2) % 3 (A(k - 1) read it last; A(k + 2) reads it two steps
        //      on), and reads W1 slot k % 3, W2 slot k & 1.  The barrier of step k + 1 waits for all but the ten youngest pieces
        //      (vmcnt(10)): W2(k - 1) and everything older has landed, W1(k + 2) stays in flight for another step -- an L2 -> LDS
        //      piece takes longer to land than the 60 MFMAs of a step leave.  42 = 0 mod 3 and mod 2: steps 40 / 41 fetch the NEXT
        //      tile's W1(0) / W1(1) into slots 0 / 1, where its steps 0 / 1 read them. ----
        // (barrier counts: + 8 for the b1 loads the step before issued last -- they stay in flight across the barrier)
        ff_barrier<0>();                                           // step 0: A(0); W1(2) -> slot 2
        step(IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}, 0, 0, IC<1>{}, 2, 2, IC<0>{}, 0, IC<1>{}, IC<1>{});
        ff_barrier<10 + FF_PRE>();                                 // step 1: A(1) || B(0); W2(0) -> slot 0, W1(3) -> slot 0
        step(IC<1>{}, IC<1>{}, IC<0>{}, IC<1>{}, 1, 1, IC<1>{}, 3, 0, IC<1>{}, 0, IC<0>{}, IC<1>{});
        int sa = 2;                                                // k % 3 of the even step below
        for (int k = 2; k < FF_NCHUNK - 2; k += 2) {
            const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
            ff_barrier<10 + FF_PRE>();                             // step k (even)
            step(IC<1>{}, IC<1>{}, IC<1>{}, IC<0>{}, k, sa, IC<1>{}, k + 2, sa1 == 2 ? 0 : sa1 + 1, IC<1>{}, k - 1, IC<0>{}, IC<1>{});
            ff_barrier<10 + FF_PRE>();                             // step k + 1 (odd)
            step(IC<1>{}, IC<1>{}, IC<1>{}, IC<1>{}, k + 1, sa1, IC<1>{}, k + 3 < FF_NCHUNK ? k + 3 : 0, sa, IC<1>{}, k, IC<0>{}, IC<1>{});
            sa = sa2;
        }
        // k = 38 (38 % 3 = 2), 39 (0): no chunk 40 / 41 exists, nothing for W1 to fetch
        ff_barrier<10 + FF_PRE>();
        step(IC<1>{}, IC<1>{}, IC<1>{}, IC<0>{}, 38, 2, IC<0>{}, 0, 0, IC<1>{}, 37, IC<0>{}, IC<1>{});
        ff_barrier<FF_PRE>();
        step(IC<1>{}, IC<1>{}, IC<1>{}, IC<1>{}, 39, 0, IC<0>{}, 0, 0, IC<1>{}, 38, IC<0>{}, IC<0>{});
        // the raw token row once more (its registers held the normalised row until A(39)): the residual of the epilogue, landing
        // under the two drain steps; the AlphaBlender's second residual likewise
        f16x8 rr[R2 ? FF_KS : 1];
        const f16* r2row = R2 ? r2g + (size_t)mr * a.ldr2 : nullptr;
#pragma unroll
        for (int s = 0; s < FF_KS; ++s) {
            xf[s] = *(const f16x8*)(xrow + 16 * s + 8 * lh);
            if constexpr (R2) rr[s] = *(const f16x8*)(r2row + 16 * s + 8 * lh);
        }
        ff_barrier<R2 ? 40 : 20>();                                // step 40: C(38) || B(39); W2(39) -> slot 1; W1(next tile's 0) -> slot 0
        step(IC<0>{}, IC<1>{}, IC<1>{}, IC<0>{}, 0, 0, IC<1>{}, 0, 0, IC<1>{}, FF_NCHUNK - 1, IC<0>{}, IC<0>{});
        ff_barrier<10>();                                          // step 41: C(39); W1(next tile's 1) -> slot 1
        step(IC<0>{}, IC<0>{}, IC<1>{}, IC<1>{}, 0, 0, IC<1>{}, 1, 1, IC<0>{}, 0, IC<0>{}, IC<0>{});
        // ---- epilogue: out = f16( f16(s_acc * O) + s1 * x' + s2 * r2 ), x' = x (+ pos).  All 20 pieces (8 columns each) are formed
        //      first and stored under ONE row test: a branch per piece would serialise the loads behind vmcnt(0) waits ----
        f16* orow = og + (size_t)mr * a.ldo;
        float lsum = 0.f;
        f16x8 op[2 * FF_NJ];
#pragma unroll
        for (int j = 0; j < FF_NJ; ++j) {
            f16x4 g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[q][e] = (f16)(s_acc * O[j][4 * q + e]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                // groups 2 k and 2 k + 1 (columns 16 k + 4 lh + e and 16 k + 8 + 4 lh + e of the tile): after the swap lanes 0-31 hold
                // columns 16 k .. 16 k + 7, lanes 32-63 columns 16 k + 8 .. 16 k + 15 of their row = 16 (2 j + k) + 8 lh + e: the
                // very columns of the token fragment xf[2 j + k]
                const u32x2 a2 = __builtin_bit_cast(u32x2, g[2 * k]), b2 = __builtin_bit_cast(u32x2, g[2 * k + 1]);
                const auto rx = __builtin_amdgcn_permlane32_swap(a2[0], b2[0], false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(a2[1], b2[1], false, false);
                const u32x4 vv = {rx[0], ry[0], rx[1], ry[1]};
                const f16x8 v16 = __builtin_bit_cast(f16x8, vv);
                const int n = 32 * j + 16 * k + 8 * lh;
                const f16x8 xr = xf[2 * j + k];
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                if constexpr (POS) { p0 = *(const f32x4*)(prow + n); p1 = *(const f32x4*)(prow + n + 4); }
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = (float)v16[e] + s1 * ((float)xr[e] + (e < 4 ? p0[e & 3] : p1[e & 3]));
                    if constexpr (R2) v += s2 * (float)rr[2 * j + k][e];
                    o[e] = (f16)v;
                    if constexpr (LNOUT) { const float vf = (float)o[e]; O[j][8 * k + e] = vf; lsum += vf; }
                }
                op[2 * j + k] = o;
            }
        }
        if (rok) {
#pragma unroll
            for (int p = 0; p < 2 * FF_NJ; ++p) *(f16x8*)(orow + 16 * p + 8 * lh) = op[p];
        }
        if constexpr (LNOUT) {
            // LayerNorm of the row just written (the norm in front of the next projection): O[j][8 k + e] = out[m][32 j + 16 k + 8 lh + e]
            lsum += __shfl_xor(lsum, 32, 64);
            const float mean = lsum * (1.0f / FF_C);
            float sq = 0.f;
#pragma unroll
            for (int j = 0; j < FF_NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = O[j][r] - mean; sq = fmaf(d, d, sq); }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / FF_C) + a.ln_eps);
            f16* lrow = (f16*)a.out_ln + (size_t)mr * a.ldoln;
#pragma unroll
            for (int p = 0; p < 2 * FF_NJ; ++p) {
                const int n = 16 * p + 8 * lh;
                const f32x4 g0 = *(const f32x4*)(a.ln_gamma + n), g1 = *(const f32x4*)(a.ln_gamma + n + 4);
                const f32x4 c0 = *(const f32x4*)(a.ln_beta + n), c1 = *(const f32x4*)(a.ln_beta + n + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    op[p][e] = (f16)fmaf((O[p >> 1][8 * (p & 1) + e] - mean) * rstd, e < 4 ? g0[e & 3] : g1[e & 3], e < 4 ? c0[e & 3] : c1[e & 3]);
            }
            if (rok) {
#pragma unroll
                for (int p = 0; p < 2 * FF_NJ; ++p) *(f16x8*)(lrow + 16 * p + 8 * lh) = op[p];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the last steps' look-ahead DMA must not outlive the wave
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
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), FF_LDS_BYTES, (hipStream_t)stream, *a, ntiles);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
