/*
 * mofa_hip.h -- C ABI of libmofa_hip.so, the MI355X (gfx950) kernels behind the
 * MOFA-Video denoising hot path.
 *
 * The reference (MyNiuuu/MOFA-Video) has no C/FFI boundary of its own: its only native
 * code is three CUDA kernels held as Python strings and JIT-launched through CuPy with raw
 * device pointers + the torch stream (MOFA-Video-Traj/models/softsplat.py:219-226,
 * :341-345); everything else dispatches through torch/diffusers.  This header is the
 * boundary a maintainer binds instead -- same calling discipline as that CuPy launch:
 * plain device pointers, sizes, and the caller's stream.  No torch types, no allocation
 * inside the library, no global state.  Every entry point returns 0 on success or a
 * negative MOFA_E* code; kernels are enqueued on `stream` and nothing synchronises.
 *
 * Layout convention: activations are fp16 "token-major" (NHWC): a tensor
 * [frames, H, W, C] is a row-major matrix of frames*H*W rows by C channels, row stride
 * `ld*` in elements.  Weights are fp16 [N][K] with K contiguous (conv: K = taps*Cin,
 * tap-major).  Small per-channel vectors (bias, norm affine, row vectors) are fp32.
 *
 * Each entry point cites the reference interface it replaces (file:line under
 * /root/reference/MOFA-Video-Traj unless noted; "diffusers" = diffusers==0.24.0, the
 * reference's pinned third-party dependency that holds the block arithmetic).
 */
#ifndef MOFA_HIP_H
#define MOFA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mofa_stream_t; /* hipStream_t */

#define MOFA_OK 0
#define MOFA_EINVAL (-22)
#define MOFA_ELAUNCH (-5)

/* library version / build probe */
int mofa_version(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM on MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate).
 *   out[m, n] = act( s_acc * (sum_{tap,k} X[src(m,tap), k] * W[n, tap*Cin + k] + bias[n] + rowvec[idx(m), n])
 *                    + s1 * R1[m, n] + s2 * R2[m, n] )
 * Rounding: fp32 accumulate; WITH a residual (R1 or R2) the term s_acc * (...) is rounded to fp16 before the residuals are added
 * in fp32 and the sum is rounded once more -- the reference's fp16 modules produce the layer's output in fp16 and then add --;
 * without residuals there is one rounding.  Every tile kernel rounds this way, so the tile choice never changes the bits of an
 * exactly representable sum (tests/test_igemm_tiles_gpu.py::test_all_tiles_round_residual_adds_alike).
 * Replaces every nn.Linear / nn.Conv2d(1x1, 3x3 s1/s2, nearest-2x + 3x3) / nn.Conv3d((3,1,1))
 * the reference reaches through diffusers blocks (ResnetBlock2D, TemporalResnetBlock,
 * Downsample2D, Upsample2D, Attention.to_q/k/v/out, FeedForward; built at
 * models/unet_spatio_temporal_condition_controlnet.py:169-232, models/controlnet_sdv.py:259-309)
 * and the adapter's own convs (models/svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py:66-155).
 * ---------------------------------------------------------------------------------------- */
enum { MOFA_MODE_PLAIN = 0, MOFA_MODE_CONV3X3 = 1, MOFA_MODE_CONVT3 = 2 };   /* CONV3X3 = k x k conv, k = ksize (1, 3, 5 or 7) */
/* MOFA_ACT_GEGLU_PAIR: W holds the GEGLU projection with value and gate rows INTERLEAVED IN BLOCKS OF 16 (rows 32 b .. 32 b + 15 =
 * value rows of outputs 16 b .. 16 b + 15, rows 32 b + 16 .. 32 b + 31 = their gate rows; bias alike): out[m][16 b + c] =
 * s_acc * val * gelu(s_acc * gate), N / 2 output columns (diffusers GEGLU, erf GELU).  mofa_video_amd/weights.py packs it. */
enum { MOFA_ACT_NONE = 0, MOFA_ACT_SILU = 1, MOFA_ACT_GEGLU_PAIR = 2, MOFA_ACT_RELU = 3,
       MOFA_ACT_GELU = 4 /* exact (erf) GELU: CLIP vision MLP, transformers CLIPMLP with hidden_act = "gelu" */ };

typedef struct mofa_igemm_args {
    const void* x;      /* fp16 activations                                                  */
    const void* w;      /* fp16 weights [N][taps*Cin]                                        */
    const float* bias;  /* fp32 [N] or NULL                                                  */
    const float* rowvec;/* fp32 [nvec][N] or NULL; idx(m) = ((m / rv_div) * rv_mul + (m % rv_mod_in)) % rv_mod_out */
    const void* r1;     /* fp16 residual [M][ldr1] or NULL                                   */
    const void* r2;     /* fp16 residual [M][ldr2] or NULL                                   */
    void* out;          /* fp16 [M][ldo]  (N columns; N/2 for MOFA_ACT_GEGLU_PAIR)           */
    int32_t M, N, Cin;  /* Cin % 64 == 0, N % 4 == 0                                         */
    int32_t ldx, ldo, ldr1, ldr2;
    int32_t mode;       /* MOFA_MODE_*                                                       */
    /* MOFA_MODE_CONV3X3: rows are (img, oy, ox); k x k taps (ksize 1, 3, 5 or 7; 0 means 3), dilation dil,  */
    /* pad dil*(k/2); stride 1|2; up = 1|2 (nearest-neighbour upsampling of the input before the conv)      */
    int32_t Hin, Win, Hout, Wout, stride, up, ksize;
    /* MOFA_MODE_CONVT3: rows are (frame, pixel); frames grouped in clips of T; T = 0: no     */
    /* clipping at clip ends (the caller placed halo frames before/after the rows)           */
    int32_t T, HW;
    int32_t rv_div, rv_mul, rv_mod_in, rv_mod_out;
    int32_t act;        /* MOFA_ACT_*                                                        */
    float s_acc, s1, s2;
    int32_t dil;        /* MOFA_MODE_CONV3X3: tap dilation (0 means 1); the CMP encoder's de-strided ResNet stages use 2 and 4
                         * (Traj/models/cmp/models/backbone/resnet.py:118-129) */
    int32_t pad;        /* MOFA_MODE_CONV3X3: MOFA_PAD_SAME (0) = dil*(k/2) on every side; MOFA_PAD_TRAILING (1) = no
                         * leading padding, taps start at input pixel stride*o (diffusers Downsample2D(padding=0) of the
                         * VAE encoder: F.pad(x, (0,1,0,1)) then a stride-2 conv)                                    */
    int32_t tile;       /* MOFA_TILE_AUTO (0): the launcher's cost model picks the output tile; any other MOFA_TILE_*
                         * forces it (parity tests run every shape through every tile) */
    void* workspace;    /* optional device scratch (16-byte aligned) owned by the caller, private to `stream` for the duration
                         * of the launch, or NULL.  With it the 256x320 tile splits the tiles of a partial last round of
                         * workgroups along K (fp32 partial tiles here, added in a fixed order by a fix-up launch on the same
                         * stream); without it every tile is computed whole.  Same result up to fp32 summation order.          */
    int64_t workspace_bytes;   /* 84 MB (256 partial tiles of 256 x 320 fp32) is never exceeded.                              */
    float* stats;       /* optional fp32 [M / 64][N]: for every block of 64 output rows and every PAIR of output columns (2 p, 2 p + 1)
                         * the sum (element 2 p) and the sum of squares (element 2 p + 1) of the fp16 outputs just written -- the
                         * GroupNorm partial sums of the layer that follows (diffusers ResnetBlock2D / TemporalResnetBlock: conv ->
                         * GroupNorm, models/controlnet_sdv.py:270-309), emitted by the producing epilogue so that the consumer does
                         * not read the activations a third time (mofa_gn_partial_from_stats turns them into `part` entries).
                         * Only where mofa_igemm_stats_ok() says so (256x320 tile; M % 64 == 0, N % 320 == 0, no activation, at most
                         * one residual, a row vector that is constant over 64-row blocks); MOFA_EINVAL otherwise.
                         * sizeof(mofa_igemm_args) = 192 */
} mofa_igemm_args;
enum { MOFA_PAD_SAME = 0, MOFA_PAD_TRAILING = 1 };
/* output tiles of the implicit GEMM: 4 waves / 2 workgroups per CU (128x128, 192x128) and the 8-wave phase-pipelined
 * 256x256 tile (needs 16-byte aligned rows and no activation on residual kinds: MOFA_EINVAL if forced on an ineligible
 * call) */
enum { MOFA_TILE_AUTO = 0, MOFA_TILE_128X128 = 2, MOFA_TILE_192X128 = 4, MOFA_TILE_256X256 = 5, MOFA_TILE_256X320 = 6 };

int mofa_igemm_f16(const mofa_igemm_args* a, mofa_stream_t stream);
/* 1 when a launch with these arguments can emit `stats` (the value of a->stats itself is ignored), else 0 */
int mofa_igemm_stats_ok(const mofa_igemm_args* a);

/* ------------------------------------------------------------------------------------------
 * Attention.  q/k/v are column blocks of token-major matrices: element (token, head, d) at
 * base[token*ld + head*head_dim + d].  Replaces diffusers AttnProcessor2_0 / F.scaled_dot_product_attention
 * inside BasicTransformerBlock.attn1 (spatial) and TemporalBasicTransformerBlock.attn1 (temporal), and the
 * softmax(QK^T)V of transformers CLIPAttention (head dim 80 in 128-column slots, mofa_video_amd/clip.py).
 * ---------------------------------------------------------------------------------------- */
/* spatial self-attention, head_dim 64 or 128: batch = nframes, sequence = S tokens per frame.
 * (The MOFA ControlNet trunk is built with heads (5,10,10,20) -- FlowControlNet calls super().__init__()
 *  without arguments, svdxt_..._norefine.py:213 -> controlnet_sdv.py:180 -- so its 1280-channel level runs
 *  10 heads x 128; the SVD-XT UNet runs 64 everywhere.)
 * v is read as it is, row-major [tokens][ldv] like q and k (the transposed fragments of the PV product come from the LDS
 * transpose read of gfx950; no pre-transposed copy of V).
 * scale > 0: softmax(scale * Q K^T) V.  scale <= 0: q already holds Q * head_dim^-0.5 * log2(e) -- the caller folded
 * that constant into its Q projection weights (blocks.SelfAttn(fold_q_scale=True)), so the kernel neither multiplies nor
 * re-rounds Q; with scale > 0 Q is multiplied by scale * log2(e) and rounded to fp16 once more inside the kernel. */
int mofa_attn_spatial_f16(const void* q, const void* k, const void* v, void* out,
                          int nframes, int heads, int head_dim, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                          mofa_stream_t stream);
/* the same with the number of 32-query blocks per wave chosen by the caller: 0 = the launcher's rule (as above), 1 = 128-row
 * workgroups, 2 = 256-row workgroups (head_dim 64 only).  Same result up to fp32 summation order; the parity tests drive
 * both kernels over the bench's shapes with it. */
int mofa_attn_spatial_qb_f16(const void* q, const void* k, const void* v, void* out,
                             int nframes, int heads, int head_dim, int S, int ldq, int ldk, int ldv, int ldo, float scale,
                             int query_blocks, mofa_stream_t stream);
/* [tokens][ld] columns in blocks of 64 (ncb = C/64 blocks) -> vt[((frame*ncb + cb)*64 + d)*S + key] = [frame][C][S]: the
 * weight operand of the VAE mid-block attention's second implicit GEMM (vae.py); the attention kernels do not need it */
int mofa_transpose_v_f16(const void* v, void* vt, int nframes, int ncb, int S, int ldv, mofa_stream_t stream);
/* temporal self-attention over T frames per (clip, pixel, head), head_dim 64 or 128, T <= 32.
 * k/v: token row of (clip b, frame t, pixel p) = (b*T + t)*HW + p, leading dim ldkv.
 * q/out: Tq <= T query frames, row (b*Tq + i)*HW + p (Tq < T when a clip's frames are sharded over ranks and the
 * keys/values were all-gathered; Tq == T otherwise). */
int mofa_attn_temporal_f16(const void* q, const void* k, const void* v, void* out,
                           int nclips, int Tq, int T, int HW, int heads, int head_dim, int ld, int ldkv, int ldo,
                           float scale, mofa_stream_t stream);
/* the same with a key-frame validity mask (bit j = key frame j exists).  Frame-sharded clips all-gather K|V into ONE
 * buffer of ranks x (largest shard) frames; with uneven shards (25 frames over 4 ranks = 7/6/6/6) the padding frames of
 * the shorter shards are masked here instead of being compacted away by a copy (rows of masked frames are never read). */
int mofa_attn_temporal_masked_f16(const void* q, const void* k, const void* v, void* out,
                                  int nclips, int Tq, int T, int HW, int heads, int head_dim, int ld, int ldkv, int ldo,
                                  float scale, uint32_t key_mask, mofa_stream_t stream);
/* in-place row softmax of an fp16 [rows][cols] matrix (VAE mid-block attention, 1 head x 512) */
int mofa_softmax_rows_f16(void* x, int rows, int cols, int ld, mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation.  GroupNorm(32) over token-major data; statistics either per frame (2-D
 * ResnetBlock2D / transformer norm) or per clip of T frames (TemporalResnetBlock: stats span
 * T*H*W).  Three launches: partial sums -> finalize to per-(frame,channel) scale/shift -> apply.
 * Replaces nn.GroupNorm + SiLU in diffusers ResnetBlock2D / TemporalResnetBlock /
 * TransformerSpatioTemporalModel.norm / conv_norm_out (unet_..._controlnet.py:236).
 * ---------------------------------------------------------------------------------------- */
/* workspace `part`: fp32 [nframes][nparts][32][2], nparts = mofa_gn_nparts(HW, C) */
int mofa_gn_nparts(int HW, int C);
int mofa_gn_partial_f16(const void* x, float* part, int nframes, int HW, int C, int ldx, mofa_stream_t stream);
/* the same `part` entries from the pair sums an implicit-GEMM epilogue emitted (mofa_igemm_args.stats: fp32 [nframes * HW / 64][C]),
 * 64-row blocks dealt to the nparts entries of a frame in order, fixed summation order; HW % 64 == 0, (C / 32) even */
int mofa_gn_partial_from_stats(const float* stats, float* part, int nframes, int HW, int C, mofa_stream_t stream);
/* frames_per_stat = 1 (spatial) or T (temporal); writes scale/shift fp32 [nframes][C] */
int mofa_gn_finalize(const float* part, const float* gamma, const float* beta, float* scale, float* shift,
                     int nframes, int HW, int C, int frames_per_stat, float eps, mofa_stream_t stream);
/* split form for frame-sharded clips: partials -> fp64 [nstat][32][2] (sum, sum of squares); the caller all-reduces
 * them over the ranks holding the clip's other frames, then finalizes with the global per-group element count */
int mofa_gn_reduce(const float* part, double* sums, int nframes, int HW, int C, int frames_per_stat, mofa_stream_t stream);
int mofa_gn_finalize_sums(const double* sums, const float* gamma, const float* beta, float* scale, float* shift,
                          int nframes, int C, int frames_per_stat, double count_per_group, float eps,
                          mofa_stream_t stream);
/* fused finalize + apply (single-rank path): every workgroup combines the partial sums of its statistics set itself (fp64,
 * fixed order) and applies y = (x - mean) * rstd * gamma + beta, optional SiLU -- no scale / shift buffers, no finalize launch.
 * Meant for frames_per_stat * mofa_gn_nparts(HW, C) <= 512 entries (100 KB of partials re-read per workgroup from L2).
 * C <= 4096 (the same limit as mofa_gn_partial_f16), C % 32 == 0. */
int mofa_gn_apply_f16(const void* x, const float* part, const float* gamma, const float* beta, void* y,
                      int nframes, int HW, int C, int ldx, int ldy, int frames_per_stat, float eps, int silu,
                      mofa_stream_t stream);
/* the same for a statistics set whose frames are sharded over ranks (new work; the reference is single-GPU): `part_all` = the
 * all-gathered partials of the WHOLE set, `nentries` x [32][2] fp32 (zero entries for padding frames), combined by every
 * workgroup in entry order; `count_per_group` = elements per group over the whole set.  Applies that one set to the `nframes`
 * frames of x (the rank's own frames, or a halo frame received raw from a neighbour shard).  C <= 4096, nentries <= 4096. */
int mofa_gn_apply_gathered_f16(const void* x, const float* part_all, int nentries, double count_per_group,
                               const float* gamma, const float* beta, void* y, int nframes, int HW, int C,
                               int ldx, int ldy, float eps, int silu, mofa_stream_t stream);
/* y = x*scale[frame][c] + shift[frame][c]; optional SiLU */
int mofa_affine_act_f16(const void* x, const float* scale, const float* shift, void* y,
                        int nframes, int HW, int C, int ldx, int ldy, int silu, mofa_stream_t stream);
/* LayerNorm over C per token (eps, affine); optional pre-add of rowvec[(m / rv_div) % rv_mod][C] */
int mofa_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int M, int C,
                       int ldx, int ldy, float eps, const float* rowvec, int rv_div, int rv_mod,
                       mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused feed-forward for 320-channel tokens (level 0 of the UNet / ControlNet): ONE launch for
 *   LayerNorm -> GEGLU projection (320 -> 2 x 1280) -> value * gelu(gate) -> output projection (1280 -> 320) -> residuals
 * i.e. the feed-forward legs of diffusers' BasicTransformerBlock (norm3 -> ff) and TemporalBasicTransformerBlock (norm_in -> ff_in,
 * norm3 -> ff) as the reference builds them (models/unet_spatio_temporal_condition_controlnet.py:169-232,
 * models/controlnet_sdv.py:259-309); replaces mofa_layernorm_f16 + two mofa_igemm_f16 launches and the [M, 1280] hidden tensor.
 *   x'[m]   = x[m] + pos[(m / HW) % T]                                  (pos == NULL: x' = x)
 *   out[m]  = f16( f16( s_acc * (W2 . (val * gelu_erf(gate)) + b2) ) + s1 * x'[m] + s2 * r2[m] ),
 *             [val | gate] = W1 . LayerNorm_eps(x'[m]) + b1
 *   out_ln[m] = LayerNorm_ln_eps(out[m]) * ln_gamma + ln_beta          (optional second output: the norm of the NEXT projection)
 * Packed operands (mofa_video_amd/weights.py::pack_ff320 writes them; all index ranges below are inclusive-exclusive):
 *   w1p  fp16 [40 chunks][2 (value, gate)][20 k-steps][64 lanes][8]: element e of lane l = W1g[t * 1280 + 32 c + (l & 31)]
 *        [16 s + 8 (l >> 5) + e], W1g = net.0.proj.weight * LayerNorm gain (per input channel), rows [0,1280) value, [1280,2560) gate
 *   b1   fp32 [2560] = net.0.proj.bias + net.0.proj.weight . LayerNorm bias                      (natural order)
 *   w2p  fp16 [40 chunks][10 out tiles][2 k-steps][64 lanes][8]: element jj of lane l = net.2.weight[32 j + (l & 31)]
 *        [32 c + 16 u + 4 (l >> 5) + (jj & 3) + 8 (jj >> 2)]
 *   b2   fp32 [320]
 * All pointers 16-byte aligned, ld* % 8 == 0.  fp16 MFMA, fp32 accumulation / LayerNorm / GELU / residual arithmetic. */
typedef struct mofa_ff320_args {
    const void* x;          /* fp16 [M][ldx]                                    */
    const float* pos;       /* fp32 [T][320] or NULL                            */
    const void* w1p;        /* packed, see above                                */
    const float* b1;
    const void* w2p;
    const float* b2;
    const void* r2;         /* fp16 [M][ldr2] or NULL                           */
    void* out;              /* fp16 [M][ldo]                                    */
    void* out_ln;           /* fp16 [M][ldoln] or NULL                          */
    const float* ln_gamma;  /* fp32 [320] (with out_ln)                         */
    const float* ln_beta;
    int32_t M, ldx, ldo, ldr2, ldoln, HW, T;
    float eps, s_acc, s1, s2, ln_eps;
    int32_t reserved[4];    /* must be 0; sizeof(mofa_ff320_args) = 152         */
} mofa_ff320_args;
int mofa_ff320_f16(const mofa_ff320_args* a, mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Linear layers on 320-channel tokens (level 0), transposed form, optional LayerNorm in front:
 *   out[m, n] = f16( f16( s_acc * (W[n, :] . xhat[m] + bias[n] + rowvec[idx(m), n]) ) + s1 * r1[m, n] ),     n < N, N % 64 == 0
 *   xhat[m] = norm ? LayerNorm_eps(x[m]) without affine part (the norm's gain / bias are folded into W / bias) : x[m]
 *   idx(m) = ((m / rv_div) * rv_mul + (m % rv_mod_in)) % rv_mod_out          (as mofa_igemm_f16; rowvec row idx at rowvec + idx * N)
 * i.e. norm1 -> to_q | to_k | to_v (N = 960), to_out.0 + attn2's vector + residual, proj_in of diffusers' BasicTransformerBlock /
 * TemporalBasicTransformerBlock / TransformerSpatioTemporalModel at 320 channels (built at
 * models/unet_spatio_temporal_condition_controlnet.py:169-232, models/controlnet_sdv.py:259-309); replaces mofa_layernorm_f16 +
 * mofa_igemm_f16 for them.  wp: fp16 [N / 64 chunks][2 tiles][20 k-steps][64 lanes][8]: element e of lane l =
 * W'[64 c + 32 t + (l & 31)][16 s + 8 (l >> 5) + e], W' = W * LayerNorm gain (mofa_video_amd/weights.py::pack_lin320);
 * bias fp32 [N] (+ W . LayerNorm bias) or NULL.  Pointers 16-byte aligned, ld* % 8 == 0.  The output is addressed through a 32-bit
 * buffer descriptor (rows beyond M are dropped by its bounds check): (M + 256) * ldo * 2 bytes must stay below 4 GB, MOFA_EINVAL
 * otherwise (the caller then uses mofa_layernorm_f16 + mofa_igemm_f16: mofa_video_amd/ops.py::lin320_fits). */
typedef struct mofa_lin320_args {
    const void* x;          /* fp16 [M][ldx], 320 channels                      */
    const void* wp;         /* packed, see above                                */
    const float* bias;      /* fp32 [N] or NULL                                 */
    const float* rowvec;    /* fp32 rows of N or NULL                           */
    const void* r1;         /* fp16 [M][ldr1] or NULL                           */
    void* out;              /* fp16 [M][ldo]                                    */
    int32_t M, N, ldx, ldo, ldr1, norm;
    int32_t rv_div, rv_mul, rv_mod_in, rv_mod_out;
    float eps, s_acc, s1;
    int32_t reserved[3];    /* must be 0; sizeof(mofa_lin320_args) = 112         */
} mofa_lin320_args;
int mofa_lin320_f16(const mofa_lin320_args* a, mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise / data movement
 * ---------------------------------------------------------------------------------------- */
/* y = a*x + b*y on contiguous fp32 vectors (latent window accumulation / averaging of the Keypoint loop,
 * MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:502-511) */
int mofa_axpby_f32(const float* x, float* y, int64_t n, float a, float b, mofa_stream_t stream);
/* nearest-neighbour resize of fp32 planes [n][H][W] -> [n][h][w], src = floor(dst * in/out) (F.interpolate 'nearest') */
int mofa_resize_nearest_f32(const float* x, float* y, int n, int H, int W, int h, int w, mofa_stream_t stream);
/* out[m][c] = a[m][c]*w[m % HW] + b[m][c]*(1 - w[m % HW]): Hybrid residual blend by the user mask
 * (MOFA-Video-Hybrid/pipeline/pipeline.py:479-489); w fp32 [HW]; C % 8 == 0 */
int mofa_mask_blend_f16(const void* a, const void* b, const float* w, void* out, int M, int C, int HW, int lda,
                        int ldb, int ldo, mofa_stream_t stream);
/* ForegroundMatting tail (MOFA-Video-Hybrid/models/occlusion/hourglass.py:266-280):
 * mask = sigmoid(logit[m]); out[m][c] = warped[m][c]*mask + matting[m][c]*(1-mask); mask_out fp32 [M] (optional) */
int mofa_matting_blend_f16(const void* warped, const void* matting, const void* logit, void* out, float* mask_out,
                           int M, int C, int ldw, int ldm, int ldl, int ldo, mofa_stream_t stream);
/* F.interpolate(scale_factor=1/s, nearest) of token-major maps [n][H][W][C] -> [n][H/s][W/s][C]
 * (landmark embedding pyramid, MOFA-Video-Hybrid/models/ldmk_ctrlnet.py:399-403) */
int mofa_subsample_tokens_f16(const void* x, void* y, int n, int H, int W, int s, int C, int ldx, int ldy,
                              mofa_stream_t stream);
/* y[m][c] = a * x[m][c] + b * y[m][c]  (fp16 storage, fp32 math); C % 8 == 0 */
int mofa_axpby_f16(const void* x, void* y, int M, int C, int ldx, int ldy, float a, float b, mofa_stream_t stream);
/* out[m][c] = a * x[m][c] + b * y[m][c], written elsewhere (x, y untouched): the UNet's skip + ControlNet residual sum goes
 * straight into its column slice of the decoder's concat buffer (unet_spatio_temporal_condition_controlnet.py:447-459, :478-483) */
int mofa_axpby_out_f16(const void* x, const void* y, void* out, int M, int C, int ldx, int ldy, int ldo, float a, float b,
                       mofa_stream_t stream);
/* out[m][j] = x[m][j] * gelu(x[m][Ch + j]), j < Ch  (diffusers GEGLU, erf gelu) */
int mofa_geglu_f16(const void* x, void* out, int M, int Ch, int ldx, int ldo, mofa_stream_t stream);
/* strided 2-D copy of a column block: dst[m][0..C) = src[m][0..C); C % 8 == 0 */
int mofa_copy2d_f16(const void* src, void* dst, int M, int C, int lds, int ldd, mofa_stream_t stream);
/* y = silu(x) on fp32 vectors (time-embedding non-linearity) */
int mofa_silu_f32(const float* x, float* y, int n, mofa_stream_t stream);
/* fp32 -> fp16 / fp16 -> fp32 contiguous casts */
int mofa_cast_f32_to_f16(const float* x, void* y, int64_t n, mofa_stream_t stream);
int mofa_cast_f16_to_f32(const void* x, float* y, int64_t n, mofa_stream_t stream);
/* NCHW fp32 [n][C][H][W] * scale -> token-major fp16 [n][H*W][ldo] (channels >= C left untouched) and back */
int mofa_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int C, int HW, int ldo, float scale,
                              mofa_stream_t stream);
int mofa_nhwc_f16_to_nchw_f32(const void* x, float* y, int n, int C, int HW, int ldx, mofa_stream_t stream);
/* sinusoidal Timesteps(dim, flip_sin_to_cos=True, shift=0): out fp32 [n][dim]  (diffusers embeddings.py) */
int mofa_timestep_embedding(const float* t, float* out, int n, int dim, mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MOFA-Adapter warp: softsplat(tenIn, tenFlow, None, 'avg')  (models/softsplat.py:232-274 +
 * kernel softsplat_out :284-345).  Deterministic gather form, fused normalisation.
 *   feat : fp16 token-major [HW][ldf]      (first-frame feature, one image)
 *   flow : fp32 [nflows][2][H][W]          (flow frame 0 -> frame i+1 at feature resolution)
 *   out  : fp16 token-major [nflows][HW][ldo]
 *   ws   : int32/fp32 workspace, mofa_softsplat_ws_bytes(nflows, H, W) bytes
 * mofa_softsplat_scatter_f32 is the atomicAdd form of the reference kernel on NCHW fp32
 * (same op order class as the CUDA kernel; used for parity classing and as a bench baseline).
 * ---------------------------------------------------------------------------------------- */
int64_t mofa_softsplat_ws_bytes(int nflows, int H, int W);
int mofa_softsplat_avg_f16(const void* feat, const float* flow, void* out, void* ws,
                           int nflows, int H, int W, int C, int ldf, int ldo, mofa_stream_t stream);
int mofa_softsplat_scatter_f32(const float* in, const float* flow, float* out_sum, int N, int C, int H, int W,
                               mofa_stream_t stream);
/* the wrapper's metric-weighted modes (models/softsplat.py:243-270; not on the inference path): 'linear' (mode 1) / 'soft' (mode 2)
 * build [in * w | w], w = metric / exp(metric), fp32 [N][C+1][H][W]; after mofa_softsplat_scatter_f32 of that, the normalisation
 * divides by the splatted last channel: eps_mode 0 '+1e-7' ('' / 'addeps'), 1 'zeroeps', 2 'clipeps', 3 unchanged */
int mofa_softsplat_weight_f32(const float* in, const float* metric, float* out, int N, int C, int H, int W, int mode,
                              mofa_stream_t stream);
int mofa_softsplat_normalize_f32(const float* summed, float* out, int N, int C, int H, int W, int eps_mode, mofa_stream_t stream);
/* F.interpolate(flow, scale_factor=1/s) (nearest) / s  (svdxt_..._norefine.py:302-309): fp32 [n][2][H][W] -> [n][2][H/s][W/s] */
int mofa_flow_downscale_f32(const float* flow, float* out, int n, int H, int W, int s, mofa_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Scheduler math (pipeline/pipeline.py:449-452, :495-500; utils/scheduling_euler_discrete_karras_fix.py:264-288, :418-528)
 * ---------------------------------------------------------------------------------------- */
/* model input: out[2][T][HW][ldo] fp16: cols 0..3 = latents/sqrt(sigma^2+1) (both CFG halves),
 * cols 4..7 = image_latents[half] ; latents fp32 [T][4][HW] (NCHW), image_latents fp32 [2][4][HW] */
int mofa_prepare_model_input(const float* latents, const float* image_latents, void* out,
                             int T, int HW, int ldo, float sigma, mofa_stream_t stream);
/* CFG + v-prediction Euler step, fp32 latents in place.
 * noise_pred fp16 token-major [2][T][HW][ldn] (uncond first), g_f = gmin + (gmax-gmin)*f/(T-1) */
int mofa_cfg_euler_step(float* latents, const void* noise_pred, int T, int HW, int ldn,
                        float sigma, float sigma_next, float gmin, float gmax, mofa_stream_t stream);
/* The same two with the step's scalars read from DEVICE memory: `step_scalars` = one row of a per-clip step table, fp32
 * [MOFA_STEP_SCALARS] = {sigma, sigma_next, timestep, timestep, 1/sqrt(sigma^2+1), 0, 0, 0} (the scheduler's sigma table
 * :337-349 uploaded once per clip; columns 2..3 double as the fp32 [2] timestep tensor of mofa_timestep_embedding).  No
 * host scalar enters a denoise step, so a whole step can be captured in a hipGraph and replayed: mofa_step_select is the
 * graph's first node -- cur[..] = table[*counter][..]; ++*counter -- and every other node reads `cur`. */
#define MOFA_STEP_SCALARS 8
int mofa_prepare_model_input_dev(const float* latents, const float* image_latents, void* out,
                                 int T, int HW, int ldo, const float* step_scalars, mofa_stream_t stream);
int mofa_cfg_euler_step_dev(float* latents, const void* noise_pred, int T, int HW, int ldn,
                            const float* step_scalars, float gmin, float gmax, mofa_stream_t stream);
int mofa_step_select(const float* table, int* counter, float* cur, int nsteps, mofa_stream_t stream);


/* ---- output stage (the step after the path) -------------------------------------------------------------
 * Replaces tensor2vid -> VaeImageProcessor.postprocess (Traj/pipeline/pipeline.py:57-69, :518): decoded fp32 frames
 * [nframes][3][H][W] -> (x/2+0.5).clamp(0,1) as fp32 NCHW ("pt"), fp32 NHWC ("np") or uint8 NHWC ("pil":
 * round-half-even(x*255)). */
#define MOFA_FRAMES_PT 0
#define MOFA_FRAMES_NP 1
#define MOFA_FRAMES_U8 2
int mofa_frames_postprocess_f32(const float* frames_nchw, void* out, int nframes, int H, int W, int mode,
                                mofa_stream_t stream);
/* Middlebury colour coding of one flow field, fp32 [H][W][2] -> uint8 [H][W][3]; replaces flow_to_image
 * (Traj/utils/flow_viz.py:241-277 with compute_color :196-238).  workspace: mofa_flow_to_image_ws_bytes(H, W). */
int64_t mofa_flow_to_image_ws_bytes(int H, int W);
int mofa_flow_to_image_u8(const float* flow_hw2, unsigned char* out_hw3, int H, int W, void* workspace,
                          mofa_stream_t stream);

/* ---- image conditioning front end (the step before the loop) ------------------------------------------------
 * _resize_with_antialiasing (MOFA-Video-Traj/pipeline/pipeline.py:531-562) = separable Gaussian blur with reflect
 * padding (_gaussian_blur2d :632-645, _filter2d :587-610; x pass then y pass) + F.interpolate(bicubic, align_corners=True).
 * fp32 planes [nplanes][H][W]; taps = k fp32 weights on the device; axis 1 = along W, 0 = along H; out != x. */
int mofa_filter1d_reflect_f32(const float* x, float* out, const float* taps, int nplanes, int H, int W, int k, int axis,
                              mofa_stream_t stream);
int mofa_resize_bicubic_ac_f32(const float* x, float* out, int nplanes, int Hin, int Win, int Hout, int Wout,
                               mofa_stream_t stream);
/* operand of the CLIP patch embedding (transformers CLIPVisionEmbeddings.patch_embedding, stride = kernel = p):
 * fp32 [nimg][C][H][W] -> fp16 [nimg*(H/p)*(W/p)][ld], column c*p*p + py*p + px, columns >= C*p*p zero */
int mofa_patchify_f16(const float* x, void* out, int nimg, int C, int H, int W, int p, int ld, mofa_stream_t stream);

/* ---- CMP sparse-to-dense motion encoder, non-convolution pieces (the step before the path) -----------------
 * Token-major fp16 maps [nimg*H*W][ld].  pool2d: nn.MaxPool2d (mode 0, padding ignored) / nn.AvgPool2d (mode 1)
 * (Traj/models/cmp/models/backbone/resnet.py:108, modules/shallownet.py:16-21, modules/decoder.py:115-139). */
int mofa_pool2d_f16(const void* x, void* out, int nimg, int Hin, int Win, int C, int ldx, int ldo, int k, int stride, int pad,
                    int mode, mofa_stream_t stream);
/* F.interpolate(mode="bilinear", align_corners=True): token-major fp16 maps (modules/decoder.py:192-211) and fp32
 * planes [nplanes][H][W] (svdxt_..._norefine.py:58-60). */
int mofa_resize_bilinear_ac_f16(const void* x, void* out, int nimg, int Hin, int Win, int Hout, int Wout, int C, int ldx, int ldo,
                                mofa_stream_t stream);
int mofa_resize_bilinear_ac_f32(const float* x, float* out, int nplanes, int Hin, int Win, int Hout, int Wout,
                                mofa_stream_t stream);
/* Fuser.convert_flow (cmp/utils/visualize_utils.py:6-19): logits fp16 [nimg*HW][ld], bins [0,nbins) x, [nbins,2nbins) y
 * -> fp32 flow [nimg][2][HW] = per-axis softmax expectation over the bin centres (b + 1/2) * 2*fmax/nbins - fmax. */
int mofa_flow_expectation_f16(const void* logits, float* flow_nchw, int nimg, int HW, int ld, int nbins, float fmax,
                              mofa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOFA_HIP_H */
