// Attention kernels for gfx950 (head_dim 64).
//
// mofa_attn_spatial_f16: flash-style self-attention over the S = h*w tokens of one frame.
//   One workgroup = 4 waves = 128 (or 256: two 32-query blocks per wave) query rows of one (frame, head); K and V
//   tiles of 64 keys are staged through double-buffered LDS (V row-major; its transposed fragments come from the LDS
//   transpose read ds_read_b64_tr_b16).  Scores are computed TRANSPOSED (S^T = K . Q^T, MFMA
//   32x32x16 f16) so every lane owns one query row: the softmax statistics are per-lane scalars
//   (one cross-half exchange per tile) and the probabilities are already laid out as the B operand of
//   O^T += V^T . P^T -- the k-slot -> key permutation of that MFMA is chosen to match the accumulator
//   layout of S^T, so P never leaves registers.  fp32 accumulation, online softmax in exp2 domain.
//
// mofa_attn_temporal_f16: self-attention over the T <= 32 frames of one (clip, pixel, head); HBM-bound, one wave per
//   sequence on the matrix cores (S^T = K . Q^T as one 32x32 tile, softmax per lane, O^T = V^T . P^T with P taken from the
//   S^T accumulators; V row-major in LDS, read through the LDS transpose read).

#include "common.h"

#define ATT_TILE 64
#define ATT_DEFER_SUM 16384.0f   // a tile whose row sum of exp2(score - reference) reaches this moves the reference (fp16 P < 65504)

// D = head dim (64 or 128).  K tile row stride D+8 halves (144 / 272 B: conflict-free ds_read_b128).
// QB = 32-query blocks per wave (1 or 2).  With QB = 2 every K / V^T fragment read from LDS feeds two MFMAs and the
// per-tile costs (tile loads, LDS store, barrier, 24 fragment reads) are paid once per 64 queries of a wave: used when
// a frame has enough query rows to fill the chip with 256-row workgroups.
// V stays ROW-major ([key][d], as the QKV projection writes it): the V^T fragments of O^T += V^T P^T are read with
// ds_read_b64_tr_b16, the LDS transpose read of gfx950.  Semantics (measured, tools note in DESIGN.md): within each group of
// 16 lanes, lane j supplies the address of an 8-byte chunk C_j (4 halves); lane i = 4 q + e of the group receives
// (C_q[e], C_{q+4}[e], C_{q+8}[e], C_{q+12}[e]).  With lane j = q + 4 r addressing V[k0 + r][d0 + 4 q .. + 3], lane i gets
// V[k0 .. k0 + 3][d0 + i]: four consecutive keys of ONE column d -- a V^T fragment piece -- from row-major data.  Row stride
// D + 32 halves: the 32 lanes of a half-wave (4 key rows x 8 chunks) then hit 32 distinct bank pairs.
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 lds_read_tr4(const f16* p) {
    const att_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)p);
    return __builtin_bit_cast(f16x4, v);
}

#ifndef ATT_NBUF_D
#define ATT_NBUF_D 2   // (3: K / V tiles fetched two ahead -- bit-identical, measured 1-2 % slower, profiles/r06_attn_anatomy.log)
#endif
template <int D, int QB>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                              const f16* __restrict__ v, f16* __restrict__ out,
                                                              int heads, int S, int ldq, int ldk, int ldv, int ldo, float c,
                                                              int nqb, int total) {
    // D = 64: K / V tiles arrive by LDS-DMA (buffer_load ... lds: no VGPR staging, no per-tile address arithmetic -- the tile
    // index is the instruction's scalar offset, rows beyond S read as zero through the descriptor's bounds check).  The DMA
    // image is lane-linear, so rows are exactly 128 bytes and conflicts are avoided by swizzling the SOURCE chunk instead of
    // padding: K chunk c of row r sits in slot c ^ ((r >> 1) & 7) (ds_read_b128 fragments), V's two 64-byte halves are
    // swapped in rows with bit 1 set (the 4 key rows x 64 bytes of a transpose read then cover 4 distinct bank quarters).
    // D = 128 (0.1 % of a clip) keeps the register-staged, padded form.
    constexpr bool DMA = D == 64;
    // DMA form: ATT_NBUF_D = 2 buffers, tile t + 1 fetched while tile t is computed.  (-DATT_NBUF_D=3: a ring of three, fetched TWO
    // tiles ahead with a counted end-of-tile wait -- the landing time of a piece is NOT what the end-of-tile wait stalls on: same
    // results bit for bit, 1-2 % slower at S = 9216 and 2304, profiles/r06_attn_anatomy.log)
    constexpr int NBUF = DMA ? ATT_NBUF_D : 2;
    constexpr int ATT_KSTR = DMA ? D : D + 8;
    constexpr int ATT_VSTR = DMA ? D : D + 32; // V tile row stride (halves)
    constexpr int KK = D / 16;       // MFMA k-steps of S^T
    constexpr int DB = D / 32;       // 32-wide output d-blocks
    constexpr int NCH = D / 32;      // 16-byte chunks per thread per tile (K and V each)
    extern __shared__ __attribute__((aligned(16))) char smem_att[];
    f16* sKb = (f16*)smem_att;                          // [NBUF][64 * ATT_KSTR]
    f16* sVb = sKb + NBUF * ATT_TILE * ATT_KSTR;        // [NBUF][64 * ATT_VSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware work order (r04b; 1-D grid of 8 * ceil(total / 8) workgroups): the hardware deals workgroup ids round-robin over
    // the 8 XCDs, so workgroup b lives on XCD b & 7; each XCD takes a CONTIGUOUS range of work ids w = (frame * heads + head) *
    // nqb + query block (the bijective split of igemm_common.h's TileWalk), i.e. the query blocks of one (frame, head) -- which
    // all stream the same K / V (2.4 MB at level 0) -- run side by side on ONE XCD and find each other's tiles in its L2, instead
    // of every XCD pulling every (frame, head)'s K / V through the fabric once: fabric reads per level-0 launch 5.4 -> 1.5 GB,
    // 964 -> 988 TF/s at S = 9216, 772 -> 815 at S = 2304, bit-identical (profiles/r04b_attn_xcd_order_ab.log).
    int head, frame, qb_;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, qn = total >> 3, rn = total & 7;
        const int start = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        if (local >= qn + (xcd < rn ? 1 : 0)) return;
        const int w = start + local, pair = w / nqb;
        qb_ = w - pair * nqb;
        frame = pair / heads;
        head = pair - frame * heads;
    }
    const int q0 = qb_ * (128 * QB) + wave * (32 * QB);

    const f16* kbase = k + (size_t)frame * S * ldk + head * D;
    const f16* vbase = v + (size_t)frame * S * ldv + head * D;

    // Q fragments (B operand of S^T): lane (query l31, half lh) holds Q[q][16*kk + 8*lh .. +8)
    f16x8 qf[QB][KK];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = q0 + b * 32 + l31;
        const f16* qp = q + ((size_t)frame * S + (qi < S ? qi : 0)) * ldq + head * D + lh * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            // Q is held pre-multiplied by c = scale * log2(e) (here: one more fp16 rounding of Q, random, 2^-11 relative --
            // callers that can fold c into their Q projection pass scale <= 0 and skip it): the
            // MFMA then yields the scores in the exp2 domain and, with the accumulator started at -m, already minus the
            // running maximum -- no per-score VALU work before v_exp_f32 (this kernel is bound by VALU issue, not by MFMA)
            const f16x8 v = (qi < S) ? *(const f16x8*)(qp + kk * 16) : zero8;
            qf[b][kk] = v;
            if (c != 1.0f) {                               // (c == 1: the caller folded scale * log2(e) into the Q projection)
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[b][kk][e] = (f16)((float)v[e] * c);
            }
        }
    }

    constexpr bool NEGM = D == 64;
    f32x16 o[QB][DB];
    f32x16 negm[QB];                                      // (NEGM) -m_run[b] in all 16 registers
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[b][r] = 0.f;
        m_run[b] = 0.f; l_run[b] = 0.f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
    }

    // loader mapping: K tile = 64 keys x D/8 chunks(16 B); V^T tile = D rows x 8 chunks; NCH chunks per thread each
    constexpr int CPR = D / 8;
    f16x8 gk[NCH], gv[NCH];
    // per-thread source pointers of tile 0 (K and V rows have the same shape: 64 keys x D), advanced by 64 rows per load:
    // 4 64-bit adds per tile instead of the address arithmetic from scratch; only a tile that reaches beyond S is
    // bounds-checked (wave-uniform branch)
    const f16* kptr[NCH];
    const f16* vptr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int cidx = tid + 256 * i;
        const int krow = cidx / CPR, kcol = cidx - krow * CPR;
        kptr[i] = kbase + (size_t)krow * ldk + kcol * 8;
        vptr[i] = vbase + (size_t)krow * ldv + kcol * 8;
    }
    const size_t kstep = (size_t)ATT_TILE * ldk, vstep = (size_t)ATT_TILE * ldv;
    auto load_tile = [&](int k0) {
        if (__builtin_amdgcn_readfirstlane(k0 + ATT_TILE <= S)) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                gk[i] = *(const f16x8*)kptr[i];
                gv[i] = *(const f16x8*)vptr[i];
            }
        } else {                                            // rows of keys beyond S are zero (K: score masked; V: 0 * p)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const bool ok = k0 + (tid + 256 * i) / CPR < S;
                gk[i] = ok ? *(const f16x8*)kptr[i] : zero8;
                gv[i] = ok ? *(const f16x8*)vptr[i] : zero8;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) { kptr[i] += kstep; vptr[i] += vstep; }
    };
    auto store_tile = [&](int buf) {
        f16* sK = sKb + buf * ATT_TILE * ATT_KSTR;
        f16* sV = sVb + buf * ATT_TILE * ATT_VSTR;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cidx = tid + 256 * i;
            const int krow = cidx / CPR, kcol = cidx - krow * CPR;
            *(f16x8*)&sK[krow * ATT_KSTR + kcol * 8] = gk[i];
            *(f16x8*)&sV[krow * ATT_VSTR + kcol * 8] = gv[i];
        }
    };
    // transpose-read address of this lane inside a V tile (see lds_read_tr4): key row (lane & 15) >> 2 (+ 4 lh), column
    // chunk 4 ((lane & 15) & 3) of the 16-column group (lane >> 4) & 1
    const int tr_off = (((lane & 15) >> 2) + 4 * lh) * ATT_VSTR + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    // ---- DMA form: descriptors over this (frame, head)'s K / V rows; per-lane byte offset of tile 0 ----
    const auto rsk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, (unsigned)(((size_t)(S - 1) * ldk + D) * 2), 0x00020000);
    const auto rsv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (unsigned)(((size_t)(S - 1) * ldv + D) * 2), 0x00020000);
    unsigned dko[2], dvo[2];                              // instruction i of this wave: tile rows (2 wave + i) * 8 + lane / 8
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i) * 8 + (lane >> 3), slot = lane & 7;
        dko[i] = (unsigned)row * (unsigned)ldk * 2u + (unsigned)((slot ^ ((row >> 1) & 7)) * 16);
        dvo[i] = (unsigned)row * (unsigned)ldv * 2u + (unsigned)((slot ^ (4 * ((row >> 1) & 1))) * 16);
    }
    auto dma_tile = [&](int t, int buf) __attribute__((always_inline)) {
        char* bk = (char*)(sKb + buf * ATT_TILE * ATT_KSTR) + (2 * wave) * 1024;
        char* bv = (char*)(sVb + buf * ATT_TILE * ATT_VSTR) + (2 * wave) * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (__attribute__((address_space(3))) void*)(bk + i * 1024), 16, dko[i],
                                                     t * ATT_TILE * ldk * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (__attribute__((address_space(3))) void*)(bv + i * 1024), 16, dvo[i],
                                                     t * ATT_TILE * ldv * 2, 0, 0);
        }
    };
    const int ksw = (l31 >> 1) & 7;                       // K fragment reads: slot = chunk ^ ksw
    const int vsw = (lane >> 3) & 1;                      // V transpose reads: this lane's key row has bit 1 set -> other half

    const int ntiles = (S + ATT_TILE - 1) / ATT_TILE;
    if constexpr (DMA) {
        dma_tile(0, 0);
        if (NBUF == 3 && ntiles > 1) {
            dma_tile(1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // tile 0 has landed; tile 1's four pieces stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

#if defined(ATT_T_NOREADK) || defined(ATT_T_NOREADV)
    f16x8 kstale[KK];                                               // fragments read ONCE (tile 0), reused for every tile
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) kstale[kk] = *(const f16x8*)(sKb + l31 * ATT_KSTR + kk * 16 + lh * 8);
#endif
    int buf = 0;                                                    // t % NBUF
    for (int t = 0; t < ntiles; ++t, buf = (buf + 1 == NBUF ? 0 : buf + 1)) {
        const int k0 = t * ATT_TILE;
#ifndef ATT_T_NODMA
        if constexpr (DMA && NBUF == 3) {
            if (t + 2 < ntiles) dma_tile(t + 2, buf == 0 ? 2 : buf - 1);   // slot (t + 2) % 3 = (t - 1) % 3: released by the last barrier
        } else if (t + 1 < ntiles) {
            if constexpr (DMA) dma_tile(t + 1, buf ^ 1);            // (the other buffer was released by the last barrier)
            else load_tile(k0 + ATT_TILE);
        }
#endif

        // ---- S^T tiles: s[b][ts][r] = score(key = k0 + 32*ts + (r&3) + 8*(r>>2) + 4*lh, query = 32*b + l31) ----
        // the accumulators start at -m_run: NEGM keeps that as a 16-register tuple per query block (it changes only when the
        // reference moves) and the first k-step takes it as its C operand -- no 32 v_mov per tile; the register-staged
        // D = 128 form has no registers to spare for it
        f32x16 s[QB][2];
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            if constexpr (!NEGM) {
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[b][ts][r] = -m_run[b];
            }
            const f16* kp = sKb + buf * ATT_TILE * ATT_KSTR + (ts * 32 + l31) * ATT_KSTR + (DMA ? 0 : lh * 8);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#ifdef ATT_T_NOREADK                           // timing-only builds (wrong results; the data flow stays): tools/attn_anatomy.sh
                const f16x8 kf = kstale[kk];
#else
                const f16x8 kf = *(const f16x8*)(kp + (DMA ? ((2 * kk + lh) ^ ksw) * 8 : kk * 16));
#endif
#pragma unroll
                for (int b = 0; b < QB; ++b)
                    s[b][ts] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[b][kk], (NEGM && kk == 0) ? negm[b] : s[b][ts], 0, 0, 0);
            }
        }
        // ---- mask (tail tile only) + online softmax (per-lane query row; the two halves hold disjoint keys) ----
        if (__builtin_amdgcn_readfirstlane(k0 + ATT_TILE > S)) {
            asm volatile("; tail tile" ::: "memory");   // keep this a real (wave-uniform) branch, not 64 selects per tile
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = k0 + ts * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key >= S) s[b][ts][r] = -1e30f;
                    }
        }
        // s = score * c - m_run (exp2 domain), and the probabilities are v_exp_f32 of the accumulators as they are.  m_run is
        // a REFERENCE, not the maximum: it is the first tile's row maximum and moves only when a tile's probabilities come
        // near the fp16 range (row sum of the tile >= 2^ATT_DEFER; then the true maximum is taken, O, l and this tile's
        // scores are rescaled by the same factor and the tile's probabilities recomputed) -- the quotient O / l does not
        // depend on the reference, fp16 keeps its relative precision at any magnitude, l and O are fp32.  So the common tile
        // needs no row maximum at all: one compare of the row sum it computes anyway (this kernel is VALU-issue bound).
        // (r04: the row sums taken from the matrix core instead -- ones[32x16] . P^T, four extra MFMAs per query block and tile,
        // 132 v_add_f32 fewer -- measured SLOWER, 946 against 972 TF/s on the L0 shape: the sums gate the reference check, so
        // the PV MFMAs wait for the extra MFMAs' results; profiles/r04_attn_rowsum_mfma_ab.log)
        f16x8 pf[QB][2][2];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            float psum = 0.f;
#pragma unroll
            for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#ifdef ATT_T_NOEXP
                    const float p = s[b][ts][r] * 1e-3f;                  // (one full-rate multiply instead of the quarter-rate v_exp_f32)
#else
                    const float p = __builtin_amdgcn_exp2f(s[b][ts][r]);   // raw v_exp_f32
#endif
#if !defined(ATT_T_NOSUM) && !defined(ATT_SUM_MFMA4) && !defined(ATT_SUM_DOT2)
                    psum += p;
#endif
                    pf[b][ts][r >> 3][r & 7] = (f16)p;
                }
#ifdef ATT_SUM_DOT2
            // (experiment) v_dot2_f32_f16 against (1, 1): 16 instead of 32 instructions, fp32 accumulation of the fp16 probabilities
            {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 one2 = {(f16)1.f, (f16)1.f};
#pragma unroll
                for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const h2 pp = {pf[b][ts][u][2 * e], pf[b][ts][u][2 * e + 1]};
                            psum = __builtin_amdgcn_fdot2(pp, one2, psum, false);
                        }
            }
#endif
#ifdef ATT_SUM_MFMA4
            // (experiment, tools/attn_sum_experiment.sh) the lane's 32 probabilities summed by eight v_mfma_f32_4x4x4_16B_f16 with an all-ones
            // A operand: every lane of a 4-lane block gets the sum of ITS OWN four fp16 values in all four result registers
            {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 ones = {(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};
                f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const h4 lo = {pf[b][ts][u][0], pf[b][ts][u][1], pf[b][ts][u][2], pf[b][ts][u][3]};
                        const h4 hi = {pf[b][ts][u][4], pf[b][ts][u][5], pf[b][ts][u][6], pf[b][ts][u][7]};
                        t4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, lo, t4, 0, 0, 0);
                        t4 = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, hi, t4, 0, 0, 0);
                    }
                psum = t4[0];
            }
#endif
            const float ptot = psum + __shfl_xor(psum, 32, 64);           // both key halves of the query row
            const bool move = !(ptot < ATT_DEFER_SUM) || t == 0;           // (NaN-safe; tile 0: m_run = 0 is no reference yet)
            if (__any(move)) {
                asm volatile("; reference moves" ::: "memory");
                float mx = s[b][0][0];
#pragma unroll
                for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][ts][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float delta = move ? mx : 0.f;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[b] += delta;
                if constexpr (NEGM) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[b][r] = -m_run[b];
                }
                l_run[b] *= alpha;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][db][r] *= alpha;
                psum = 0.f;
#pragma unroll
                for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(s[b][ts][r] - delta);
                        psum += p;
                        pf[b][ts][r >> 3][r & 7] = (f16)p;
                    }
            }
            l_run[b] += psum;
        }

        // ---- O^T[d][q] += V^T[d][key] * P^T[key][q]; k-slot (8*lh + jj) of MFMA (ts,u) = key
        //      32*ts + 16*u + 4*lh + (jj&3) + 8*(jj>>2), identical for both operands ---------------
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f16* vp = sVb + buf * ATT_TILE * ATT_VSTR + tr_off + (DMA ? (db ^ vsw) : db) * 32;
#pragma unroll
            for (int ts = 0; ts < 2; ++ts)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#ifdef ATT_T_NOREADV
                    const f16x8 vf = kstale[(db + ts + u) % KK];
#else
                    const f16x4 lo = lds_read_tr4(vp + (ts * 32 + u * 16) * ATT_VSTR);        // keys k0 + 4 lh + 0..3
                    const f16x4 hi = lds_read_tr4(vp + (ts * 32 + u * 16 + 8) * ATT_VSTR);    // keys k0 + 4 lh + 8..11
                    const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#endif
#pragma unroll
                    for (int b = 0; b < QB; ++b) o[b][db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[b][ts][u], o[b][db], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles) {
            // this wave's pieces of tile t + 1 have landed (the loop has no other vector-memory operation: with the ring of three,
            // the four pieces of tile t + 2 -- the youngest -- may stay in flight)
            if constexpr (DMA && NBUF == 3) {
                if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else store_tile(buf ^ 1);
        }
#ifndef ATT_T_NOBAR
        __syncthreads();
#endif
    }

#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + b * 32 + l31;
        if (qi < S) {
            f16* op = out + ((size_t)frame * S + qi) * ldo + head * D;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    f16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (f16)(o[b][db][4 * qd + e] * inv);
                    *(f16x4*)(op + db * 32 + 8 * qd + 4 * lh) = v;
                }
        }
    }
}

template <int D, int QB>
static int launch_attn_spatial(const void* q, const void* k, const void* v, void* out, int nframes, int heads, int S,
                               int ldq, int ldk, int ldv, int ldo, float c, hipStream_t st) {
    constexpr int LDS = D == 64 ? ATT_NBUF_D * (ATT_TILE * D + ATT_TILE * D) * 2 : 2 * (ATT_TILE * (D + 8) + ATT_TILE * (D + 32)) * 2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_spatial_kernel<D, QB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
            hipSuccess)
            return MOFA_ELAUNCH;
        attr_set = true;
    }
    const int nqb = cdiv(S, 128 * QB);
    const long long total = (long long)nqb * heads * nframes;
    if (total > 0x7ffffff0LL) return MOFA_EINVAL;
    dim3 grid((unsigned)(8 * ((total + 7) / 8)));              // (the kernel's XCD-aware work order)
    hipLaunchKernelGGL((attn_spatial_kernel<D, QB>), grid, dim3(256), LDS, st, (const f16*)q, (const f16*)k, (const f16*)v,
                       (f16*)out, heads, S, ldq, ldk, ldv, ldo, c, nqb, (int)total);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

extern "C" int mofa_attn_spatial_qb_f16(const void* q, const void* k, const void* v, void* out, int nframes, int heads,
                                        int head_dim, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                                        int query_blocks, mofa_stream_t stream) {
    if (!q || !k || !v || !out || nframes <= 0 || heads <= 0 || S <= 0) return MOFA_EINVAL;
    if (ldq % 8 != 0 || ldk % 8 != 0 || ldv % 8 != 0 || ldo % 4 != 0) return MOFA_EINVAL;
    if (query_blocks < 0 || query_blocks > 2 || (query_blocks == 2 && head_dim != 64)) return MOFA_EINVAL;
    // scale <= 0: q already holds Q * head_dim^-0.5 * log2(e) (folded into the projection weights: no rounding of Q here)
    const float c = scale > 0.f ? scale * 1.4426950408889634f : 1.0f;
    // 64 queries per wave (256-row workgroups: +6-7 % at S = 9216 / 2304) when S tiles by 256 with <= 1/16 waste and the
    // grid still gives >= 4 workgroups per CU; query_blocks = 1 | 2 forces either (parity tests)
    const bool two = query_blocks ? query_blocks == 2
                                  : ((long long)cdiv(S, 256) * 256 * 16 <= (long long)S * 17 && (long long)cdiv(S, 256) * heads * nframes >= 1024);
    if (head_dim == 64)
        return two ? launch_attn_spatial<64, 2>(q, k, v, out, nframes, heads, S, ldq, ldk, ldv, ldo, c, (hipStream_t)stream)
                   : launch_attn_spatial<64, 1>(q, k, v, out, nframes, heads, S, ldq, ldk, ldv, ldo, c, (hipStream_t)stream);
    if (head_dim == 128)
        return launch_attn_spatial<128, 1>(q, k, v, out, nframes, heads, S, ldq, ldk, ldv, ldo, c, (hipStream_t)stream);
    return MOFA_EINVAL;
}

extern "C" int mofa_attn_spatial_f16(const void* q, const void* k, const void* v, void* out, int nframes, int heads,
                                     int head_dim, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                                     mofa_stream_t stream) {
    return mofa_attn_spatial_qb_f16(q, k, v, out, nframes, heads, head_dim, S, ldq, ldk, ldv, ldo, scale, 0, stream);
}

// ---------------------------------------------------------------------------------------------------
// V [tokens][ldv] columns c (64-column blocks) -> V^T [(frame*ncb + cb)*64 + d][S]  (= [frame][C][S], any head dim).
// Not used by the attention kernels any more (they read V row-major through the LDS transpose read): the VAE mid-block
// attention, which materialises its scores with two implicit-GEMM launches, takes V^T as the weight operand of the second.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_v_kernel(const f16* __restrict__ v, f16* __restrict__ vt, int heads,
                                                          int S, int ldv) {
    __shared__ f16 s[64][66];
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64, head = blockIdx.y, frame = blockIdx.z;
    {
        const int key = tid >> 2, dc = (tid & 3) * 16;
        if (k0 + key < S) {
            const f16* p = v + ((size_t)frame * S + k0 + key) * ldv + head * 64 + dc;
            const f16x8 a = *(const f16x8*)p, b = *(const f16x8*)(p + 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[dc + e][key] = a[e]; s[dc + 8 + e][key] = b[e]; }
        }
    }
    __syncthreads();
    {
        const int d = tid >> 2, kc = (tid & 3) * 16;
        f16* p = vt + ((size_t)(frame * heads + head) * 64 + d) * S + k0 + kc;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (k0 + kc + half * 8 < S) {
                f16x8 a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = s[d][kc + half * 8 + e];
                *(f16x8*)(p + half * 8) = a;
            }
        }
    }
}

extern "C" int mofa_transpose_v_f16(const void* v, void* vt, int nframes, int heads, int S, int ldv,
                                    mofa_stream_t stream) {
    if (!v || !vt || nframes <= 0 || heads <= 0 || S <= 0 || S % 8 != 0 || ldv % 8 != 0) return MOFA_EINVAL;
    dim3 grid(cdiv(S, 64), heads, nframes);
    hipLaunchKernelGGL(transpose_v_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)v, (f16*)vt, heads, S,
                       ldv);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

// ---------------------------------------------------------------------------------------------------
// Temporal attention: sequence = the T <= 32 frames of one (clip, pixel, head).  One wave per sequence, on the matrix
// cores: S^T = K Q^T (one 32 x 32 tile, D / 16 MFMAs; a lane owns one query, the two lane halves disjoint keys), softmax
// per lane, O^T = V^T P^T (D / 32 d-blocks x 2 MFMAs) with P straight from the S^T accumulators -- the same fragment
// algebra as attn_spatial_kernel on a single key tile.  K and V go to LDS as they are ([key][D + 8] / [key][D + 32]); the
// V^T fragments come from the LDS transpose read (lds_read_tr4).  The r01 kernel did
// the two products with fp32 VALU FMAs (2 048 per lane and sequence = 8 192 cycles per wave for 12.8 KB of traffic):
// VALU-bound at 2.5 TB/s, not HBM-bound as its roofline entry said.
// Tq query frames (rows of q / out, clip stride Tq*HW), T key/value frames (rows of k / v, clip stride T*HW): Tq < T when
// the clip's frames are sharded over ranks and K|V were all-gathered.  key_mask: bit j clear = key frame j does not exist
// (padding rows of uneven frame shards in the gathered buffer: never read, weight exactly 0).
// ---------------------------------------------------------------------------------------------------
template <int D, int WPB>
__global__ __launch_bounds__(64 * WPB) void attn_temporal_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                                 const f16* __restrict__ v, f16* __restrict__ out,
                                                                 long long nseq, int Tq, int T, int HW, int heads, int ld,
                                                                 int ldkv, int ldo, float scale, unsigned key_mask) {
    constexpr int DC = D / 8;       // 16-byte chunks per row
    constexpr int KK = D / 16;      // MFMA k-steps of S^T
    constexpr int DB = D / 32;      // 32-wide output d-blocks
    constexpr int KSTR = D + 8;     // K row stride in halves (conflict-free ds_read_b128)
    constexpr int VSTR = D + 32;    // V row stride in halves (see lds_read_tr4)
    __shared__ __attribute__((aligned(16))) f16 sK[WPB][32 * KSTR];
    __shared__ __attribute__((aligned(16))) f16 sV[WPB][32 * VSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long seq = (long long)blockIdx.x * WPB + wave;
    if (seq >= nseq) return;                                       // (no workgroup barrier below: LDS regions are per wave)
    const int l31 = lane & 31, lh = lane >> 5;
    const int head = (int)(seq % heads);
    const long long bp = seq / heads;
    const int p = (int)(bp % HW);
    const int b = (int)(bp / HW);
    const size_t base = ((size_t)b * T * HW + p);                  // k / v token row of frame 0
    const size_t qbase = ((size_t)b * Tq * HW + p);                // q / out token row of frame 0
    f16* wK = sK[wave];
    f16* wV = sV[wave];
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- K rows and V^T columns of the existing key frames; everything else stays zero (a masked key's probability is
    //      exactly 0, but 0 * NaN from never-written memory would not be) ----
    for (int c = lane; c < 32 * DC; c += 64) {
        *(f16x8*)&wK[(c / DC) * KSTR + (c % DC) * 8] = zero8;
        *(f16x8*)&wV[(c / DC) * VSTR + (c % DC) * 8] = zero8;
    }
    // Q fragments (B operand of S^T): lane (query l31, half lh) holds Q[q][16 kk + 8 lh .. + 8)
    f16x8 qf[KK];
    {
        const f16* qp = q + (qbase + (size_t)(l31 < Tq ? l31 : 0) * HW) * ld + head * D + lh * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) qf[kk] = l31 < Tq ? *(const f16x8*)(qp + kk * 16) : zero8;
    }
    for (int c = lane; c < T * DC; c += 64) {
        const int t = c / DC, cc = c - t * DC;
        if (!((key_mask >> t) & 1u)) continue;
        const size_t row = base + (size_t)t * HW;
        const f16x8 kv = *(const f16x8*)(k + row * ldkv + head * D + cc * 8);
        const f16x8 vv = *(const f16x8*)(v + row * ldkv + head * D + cc * 8);
        *(f16x8*)&wK[t * KSTR + cc * 8] = kv;
        *(f16x8*)&wV[t * VSTR + cc * 8] = vv;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // this wave's LDS writes are in place
    __builtin_amdgcn_wave_barrier();

    // ---- S^T: s[r] = score(key = (r & 3) + 8 (r >> 2) + 4 lh, query = l31) ----
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
        const f16* kp = wK + l31 * KSTR + lh * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const f16x8 kf = *(const f16x8*)(kp + kk * 16);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[kk], s, 0, 0, 0);
        }
    }
    const float c2 = scale * 1.4426950408889634f;
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= T || !((key_mask >> key) & 1u)) s[r] = -1e30f;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c2;
    float sum = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -mc));   // masked keys: exp2(-huge) = 0
        sum += pr;
        pf[r >> 3][r & 7] = (f16)pr;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T[d][q] += V^T[d][key] P^T[key][q]; k-slot (8 lh + jj) of MFMA u = key 16 u + 4 lh + (jj & 3) + 8 (jj >> 2) ----
    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        const f16* vp = wV + (((lane & 15) >> 2) + 4 * lh) * VSTR + db * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f16x4 lo = lds_read_tr4(vp + (u * 16) * VSTR), hi = lds_read_tr4(vp + (u * 16 + 8) * VSTR);
            const f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[u], o[db], 0, 0, 0);
        }
    }
    if (l31 < Tq) {
        f16* op = out + (qbase + (size_t)l31 * HW) * ldo + head * D;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                f16x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (f16)(o[db][4 * qd + e] * inv);
                *(f16x4*)(op + db * 32 + 8 * qd + 4 * lh) = w;
            }
    }
}

extern "C" int mofa_attn_temporal_masked_f16(const void* q, const void* k, const void* v, void* out, int nclips, int Tq,
                                             int T, int HW, int heads, int head_dim, int ld, int ldkv, int ldo, float scale,
                                             uint32_t key_mask, mofa_stream_t stream) {
    if (!q || !k || !v || !out || nclips <= 0 || T <= 0 || T > 32 || Tq <= 0 || Tq > T || HW <= 0 || heads <= 0)
        return MOFA_EINVAL;
    if (T < 32) key_mask &= (1u << T) - 1u;
    if (key_mask == 0) return MOFA_EINVAL;
    if (ld % 8 != 0 || ldkv % 8 != 0 || ldo % 8 != 0) return MOFA_EINVAL;
    const long long nseq = (long long)nclips * HW * heads;
    if (head_dim == 64) {
        hipLaunchKernelGGL((attn_temporal_kernel<64, 4>), dim3(cdiv(nseq, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, nseq, Tq, T, HW, heads, ld, ldkv, ldo, scale,
                           key_mask);
    } else if (head_dim == 128) {
        hipLaunchKernelGGL((attn_temporal_kernel<128, 2>), dim3(cdiv(nseq, 2)), dim3(128), 0, (hipStream_t)stream,
                           (const f16*)q, (const f16*)k, (const f16*)v, (f16*)out, nseq, Tq, T, HW, heads, ld, ldkv, ldo, scale,
                           key_mask);
    } else {
        return MOFA_EINVAL;
    }
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}

extern "C" int mofa_attn_temporal_f16(const void* q, const void* k, const void* v, void* out, int nclips, int Tq, int T,
                                      int HW, int heads, int head_dim, int ld, int ldkv, int ldo, float scale,
                                      mofa_stream_t stream) {
    return mofa_attn_temporal_masked_f16(q, k, v, out, nclips, Tq, T, HW, heads, head_dim, ld, ldkv, ldo, scale, 0xffffffffu,
                                         stream);
}

// ---------------------------------------------------------------------------------------------------
// In-place row softmax (VAE mid-block attention: 1 head x 512, scores materialised per frame)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(f16* __restrict__ x, int cols, int ld) {
    __shared__ float red[8];
    f16* row = x + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -1e30f;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        const f16x8 a = *(const f16x8*)(row + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)a[e]);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        const f16x8 a = *(const f16x8*)(row + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf((float)a[e] - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        f16x8 a = *(const f16x8*)(row + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (f16)(__expf((float)a[e] - mx) * inv);
        *(f16x8*)(row + c) = a;
    }
}

extern "C" int mofa_softmax_rows_f16(void* x, int rows, int cols, int ld, mofa_stream_t stream) {
    if (!x || rows <= 0 || cols <= 0 || cols % 8 != 0 || ld % 8 != 0) return MOFA_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (f16*)x, cols, ld);
    MOFA_CHECK_LAUNCH();
    return MOFA_OK;
}
