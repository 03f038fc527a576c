"""Parity near the fp16 range (round-5 verdict, missing 7 / weak 3): every other GPU test and the bench run default-init-scale random
weights whose activations stay O(1); trained SVD weights do not (the reference carries ``force_upcast`` because its VAE overflows
fp16, MOFA-Video-Traj/pipeline/pipeline.py:343-352).  Here the kernels are driven with trained-like statistics -- outlier channels
(weight rows x 30-100), pre-activations of several thousand, attention logits that cross the deferred-reference threshold with
|V| up to 1e3, GroupNorm / LayerNorm inputs around 1e4 -- and compared with

  * the fp32 reference of the op (plain PyTorch, the oracle's arithmetic for these blocks: oracle/blocks.py), and
  * the same reference with the reference's fp16 module boundaries applied (each module's output rounded to fp16: what the
    reference's fp16 UNet computes, diffusers modules under ``torch_dtype=torch.float16``).

Stated bars: wherever the fp16-rounded reference is finite the product is finite and within the op's stated tolerance of the fp32
reference (2e-3 * (max|ref| + |ref|) for GEMM-class ops and norms, 4e-3 attention, 3e-3 the fused feed-forward); where the rounded
reference overflows to +-inf the product may be +-inf or the saturated fp16 value -- never NaN from a finite input."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
F16_MAX = 65504.0


@pytest.fixture(scope="module")
def ops():
    from mofa_video_amd import lib
    from mofa_video_amd import ops as o
    lib.load()
    return o


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def _check(got, ref32, ref16, tol, what):
    """ref16: the fp16-boundary reference (may hold inf).  Finite there => product finite and close to the fp32 reference."""
    got = got.float()
    fin = torch.isfinite(ref16.float())
    assert fin.float().mean().item() > 0.5, f"{what}: the test drives most of the tensor to overflow -- not a range test any more"
    assert not torch.isnan(got).any(), f"{what}: NaN in the product from finite inputs"
    assert torch.isfinite(got[fin]).all(), f"{what}: product overflows where the fp16-rounded reference is finite"
    scale = ref32[fin].abs().max()
    err = (got[fin] - ref32[fin]).abs()
    bad = err > tol * (scale + ref32[fin].abs())
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {int(fin.sum())} outside {tol:g}, max err {err.max().item():.4e} at scale "
                           f"{scale.item():.4e}")
    return scale.item()


@pytest.mark.parametrize("tile", [2, 5, 6])                        # 128x128, 256x256, 256x320 (lib.TILE_*)
@pytest.mark.parametrize("kind", ["bias", "residual", "silu", "gelu"])
def test_igemm_outlier_channels(ops, tile, kind):
    """a projection whose weight has outlier rows (x 30 .. x 100, as trained transformer / VAE weights do) on inputs of a few
    hundred: pre-activations up to ~4e4, the residual add after the fp16 rounding of the layer's own output"""
    from mofa_video_amd import lib as L
    M, N, K = 2048 + 77, 640, 320
    g = _gen(1)
    x = (torch.randn(M, K, generator=g, device=DEV) * 60).half()
    w = torch.randn(N, K, generator=g, device=DEV) * K ** -0.5
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(2))[:8].to(DEV)
    w[rows] *= torch.linspace(30, 100, 8, device=DEV)[:, None]
    w = w.half()
    bias = torch.randn(N, generator=g, device=DEV) * 10
    acc = x.float() @ w.float().T + bias
    assert acc.abs().max().item() > 2e4, acc.abs().max().item()
    kw = {}
    if kind == "residual":
        r1 = (torch.randn(M, N, generator=g, device=DEV) * 3e3).half()
        kw.update(r1=r1, s1=1.0)
        ref32, ref16 = acc + r1.float(), (acc.half().float() + r1.float()).half()
    elif kind == "silu":
        kw.update(act=L.ACT_SILU)
        ref32, ref16 = F.silu(acc), F.silu(acc.half().float()).half()
    elif kind == "gelu":
        kw.update(act=L.ACT_GELU)
        ref32, ref16 = F.gelu(acc), F.gelu(acc.half().float()).half()
    else:
        ref32, ref16 = acc, acc.half()
    out = ops.igemm(x, w, bias, tile=tile, **kw)
    _check(out, ref32, ref16, 2e-3, f"igemm {kind} tile {tile}")


def test_igemm_overflow_saturates_like_the_fp16_reference(ops):
    """outputs beyond 65504: the fp16-rounded reference holds +-inf there; the product must hold +-inf (or the same sign's maximum),
    never NaN, and stay exact elsewhere"""
    M, N, K = 512, 320, 320
    g = _gen(3)
    x = (torch.randn(M, K, generator=g, device=DEV) * 60).half()
    w = (torch.randn(N, K, generator=g, device=DEV) * 30).half()
    acc = x.float() @ w.float().T
    ref16 = acc.half()
    over = ~torch.isfinite(ref16.float())
    assert 0.001 < over.float().mean().item() < 0.5
    out = ops.igemm(x, w).float()
    assert not torch.isnan(out).any()
    assert (torch.sign(out[over]) == torch.sign(acc[over])).all() and (out[over].abs() >= F16_MAX).all()
    fin = ~over
    assert ((out[fin] - acc[fin]).abs() <= 2e-3 * (acc[fin].abs().max() + acc[fin].abs())).all()


def test_geglu_projection_large_gate(ops):
    """GEGLU projection (diffusers GEGLU: value * gelu(gate)) with gates of +-30 (erf saturated: gelu(g) = g or 0 exactly) and
    values of a few hundred: hidden states up to ~2e4"""
    from mofa_video_amd import lib as L
    from mofa_video_amd.weights import interleave_geglu
    M, K, Ch = 1024, 320, 640
    g = _gen(4)
    x = (torch.randn(M, K, generator=g, device=DEV) * 8).half()
    w = torch.randn(2 * Ch, K, generator=g, device=DEV) * K ** -0.5
    w[:Ch] *= 4.0                                                   # values
    w[Ch:] *= 0.5                                                   # gates ~ +-4 typical
    w[Ch:Ch + 16] *= 8.0                                            # a block of saturated gates
    w = w.half()
    b = torch.randn(2 * Ch, generator=g, device=DEV)
    wi, bi = interleave_geglu(w, b)
    out = ops.igemm(x, wi.contiguous(), bi.float().contiguous(), act=L.ACT_GEGLU_PAIR)
    p = x.float() @ w.float().T + b
    ref32 = p[:, :Ch] * F.gelu(p[:, Ch:])
    _check(out, ref32, ref32.half(), 2e-3, "GEGLU projection")


@pytest.mark.parametrize("kind", ["plain", "r2"])
def test_fused_ff320_outliers(ops, kind):
    """the fused level-0 feed-forward with a LayerNorm gain spread of 0.05 .. 20 (folded into W1), outlier projection rows, inputs with
    a large common offset (mean 300, std 20: the norm must remove the offset exactly) and a hidden state of several thousand"""
    from mofa_video_amd.weights import pack_ff320
    g = torch.Generator().manual_seed(5)
    w1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5
    w1[:8] *= 40.0                                                  # outlier value rows
    w1[1280:1288] *= 6.0                                            # ... and their gates well into saturation
    w1 = w1.half()
    b1 = torch.randn(2560, generator=g) * 0.5
    w2 = (torch.randn(320, 1280, generator=g) * 1280 ** -0.5).half()
    b2 = torch.randn(320, generator=g)
    gamma = torch.exp(torch.randn(320, generator=g) * 1.2).clamp(0.05, 20.0)
    beta = torch.randn(320, generator=g) * 0.5
    w1p, b1f, w2p = [t.to(DEV) for t in pack_ff320(w1, b1, w2, gamma, beta)]
    M = 128 * 9 + 50
    gg = _gen(6)
    x = (torch.randn(M, 320, generator=gg, device=DEV) * 20 + 300).half()
    kw, r2 = {}, None
    if kind == "r2":
        r2 = (torch.randn(M, 320, generator=gg, device=DEV) * 2e3).half()
        kw.update(r2=r2, s_acc=0.5, s1=0.5, s2=0.5)
    out = ops.ff320(x, w1p, b1f, w2p, b2.to(DEV), **kw)
    W1, B1, W2, B2, G, Bt = [t.to(DEV).float() for t in (w1, b1, w2, b2, gamma, beta)]
    xn = F.layer_norm(x.float(), (320,), G, Bt, 1e-5)
    p = xn @ W1.T + B1
    h = p[:, :1280] * F.gelu(p[:, 1280:])
    assert h.abs().max().item() > 2e3
    s_acc, s1, s2 = kw.get("s_acc", 1.0), kw.get("s1", 1.0), kw.get("s2", 0.0)
    y = s_acc * (h @ W2.T + B2) + s1 * x.float() + (s2 * r2.float() if r2 is not None else 0.0)
    # fp16 module boundaries of the reference: norm output, hidden state, ff output, sum
    p16 = F.layer_norm(x.float(), (320,), G, Bt, 1e-5).half().float() @ W1.T + B1
    h16 = (p16[:, :1280].half().float() * F.gelu(p16[:, 1280:].half().float())).half().float()
    y16 = ((s_acc * (h16 @ W2.T + B2)).half().float() + s1 * x.float() + (s2 * r2.float() if r2 is not None else 0.0)).half()
    # the norm's gain is folded into W1 (rounded to fp16 AFTER the fold) instead of being applied to the normalised tokens: one
    # rounding of w * g instead of one of xhat * g -- the same size of error, so the op's stated 3e-3 applies
    _check(out, y, y16, 3e-3, f"fused ff320 {kind}")


@pytest.mark.parametrize("hd,qb,S", [(64, 2, 1024), (64, 1, 200), (128, 1, 200)])
def test_attn_spatial_large_logits_and_values(ops, hd, qb, S):
    """logits up to ~ +-90 (natural units) with a few dominant keys per query -- the deferred reference has to move (a tile's
    probability sum reaches 2^14) -- and |V| up to 1e3: fp16 P times fp16 V accumulates to outputs of ~1e3"""
    frames, heads = (16, 2) if qb == 2 else (2, 1)
    Cc = heads * hd
    g = _gen(7)
    q = torch.randn(frames * S, Cc, generator=g, device=DEV) * 4.0
    k = torch.randn(frames * S, Cc, generator=g, device=DEV) * 4.0
    v = torch.randn(frames * S, Cc, generator=g, device=DEV) * 250.0
    v[::37] *= 4.0                                                   # rows of ~1e3
    qkv = torch.cat([q, k, v], 1).half()
    out = ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], frames, heads, S, head_dim=hd, query_blocks=qb)
    qf, kf, vf = [t.float().reshape(frames, S, heads, hd).transpose(1, 2) for t in qkv.split(Cc, dim=1)]
    logits = (qf @ kf.transpose(-1, -2)) * hd ** -0.5
    assert logits.abs().max().item() > 60 and vf.abs().max().item() > 1e3
    ref = (torch.softmax(logits, -1) @ vf).transpose(1, 2).reshape(-1, Cc)
    _check(out, ref, ref.half(), 4e-3, f"attn spatial hd {hd} qb {qb}")


def test_attn_temporal_large_logits_and_values(ops):
    T, HW, heads, hd = 25, 64, 5, 64
    Cc = heads * hd
    g = _gen(8)
    q = torch.randn(T * HW, Cc, generator=g, device=DEV) * 4.0
    k = torch.randn(T * HW, Cc, generator=g, device=DEV) * 4.0
    v = torch.randn(T * HW, Cc, generator=g, device=DEV) * 400.0
    qkv = torch.cat([q, k, v], 1).half()
    out = ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], 1, T, HW, heads, head_dim=hd)

    def split(t):                                                   # [T*HW, C] -> [HW, heads, T, hd]
        return t.float().reshape(T, HW, heads, hd).permute(1, 2, 0, 3)
    qf, kf, vf = split(qkv[:, :Cc]), split(qkv[:, Cc:2 * Cc]), split(qkv[:, 2 * Cc:])
    logits = (qf @ kf.transpose(-1, -2)) * hd ** -0.5
    assert logits.abs().max().item() > 60
    ref = (torch.softmax(logits, -1) @ vf).permute(2, 0, 1, 3).reshape(T * HW, Cc)
    _check(out, ref, ref.half(), 4e-3, "attn temporal")


@pytest.mark.parametrize("C,HW,frames,fps,silu", [(320, 2304, 4, 1, True), (320, 576, 8, 4, True), (1280, 144, 6, 3, False), (128, 9216, 2, 1, True)])
def test_group_norm_inputs_near_1e4(ops, C, HW, frames, fps, silu):
    """GroupNorm(32) (+ SiLU) over activations of mean 6e3, std 2.5e3 with outlier channels up to ~4e4: the partial sums of squares
    are ~1e13 per statistics set (fp32 partials, fp64 combination); per frame and per clip of `fps` frames (TemporalResnetBlock)"""
    g = _gen(9)
    x = torch.randn(frames * HW, C, generator=g, device=DEV) * 2.5e3 + 6e3
    x[:, ::41] *= 4.0
    x = x.clamp(-6e4, 6e4).half()
    assert x.float().abs().max().item() > 3e4 and torch.isfinite(x.float()).all()
    gamma = 1 + 0.3 * torch.randn(C, generator=g, device=DEV)
    beta = 0.3 * torch.randn(C, generator=g, device=DEV)
    out = ops.group_norm(x, gamma, beta, frames, HW, 1e-6, frames_per_stat=fps, silu=silu)
    xr = x.float().reshape(frames // fps, fps * HW, C).transpose(1, 2)
    ref = F.group_norm(xr.double(), 32, gamma.double(), beta.double(), eps=1e-6).float()
    if silu:
        ref = F.silu(ref)
    ref = ref.transpose(1, 2).reshape(frames * HW, C)
    _check(out, ref, ref.half(), 2e-3, f"GroupNorm C {C} HW {HW} fps {fps}")


def test_group_norm_from_epilogue_statistics_large_values(ops):
    """the same through the pair sums the producing implicit GEMM emits (mofa_igemm_args.stats): sums of fp16 outputs of ~1e4"""
    C, HW, frames = 320, 2304, 4
    g = _gen(10)
    x = (torch.randn(frames * HW, 64, generator=g, device=DEV) * 30).half()
    w = (torch.randn(C, 64, generator=g, device=DEV) * 30).half()
    bias = torch.randn(C, generator=g, device=DEV) * 4e3 + 6e3
    y = ops.igemm(x, w, bias, stats=True)
    assert getattr(y, "gn_stats", None) is not None and y.float().abs().max().item() > 2e4 and torch.isfinite(y.float()).all()
    yc = y.clone()
    gamma = 1 + 0.3 * torch.randn(C, generator=g, device=DEV)
    beta = 0.3 * torch.randn(C, generator=g, device=DEV)
    out = ops.group_norm(y, gamma, beta, frames, HW, 1e-6, silu=True)
    ref = F.silu(F.group_norm(yc.double().reshape(frames, HW, C).transpose(1, 2), 32, gamma.double(), beta.double(), eps=1e-6)).float()
    ref = ref.transpose(1, 2).reshape(frames * HW, C)
    _check(out, ref, ref.half(), 2e-3, "GroupNorm from epilogue statistics")


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layer_norm_large_offset(ops, C):
    """LayerNorm over rows with a common offset of 2e4 and std 30 (|x| < 65504): mean subtraction in fp32, two passes"""
    M = 4096 + 33
    g = _gen(11)
    x = (torch.randn(M, C, generator=g, device=DEV) * 30 + torch.randn(M, 1, generator=g, device=DEV) * 2e4).clamp(-6e4, 6e4).half()
    gamma = 1 + 0.3 * torch.randn(C, generator=g, device=DEV)
    beta = 0.3 * torch.randn(C, generator=g, device=DEV)
    out = ops.layer_norm(x, gamma, beta, 1e-5)
    ref = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5).float()
    _check(out, ref, ref.half(), 2e-3, f"LayerNorm C {C}")


@pytest.mark.parametrize("OUTLIER", [8.0, 30.0])
def test_unet_forward_trained_like_statistics_vs_fp32_and_fp16_oracle(OUTLIER):
    """the WHOLE UNet forward (level 0 = 320 channels: the fused feed-forward and norm + q|k|v launches included) with trained-like
    weight statistics instead of the default-init ones of every other test: outlier output channels (x 8 / x 30 on one row in 40 of
    every convolution / linear weight), norm gains spread over exp(N(0, 0.4)), q / k projections x 2.5 (logits x 6).  With x 30 the
    residual stream itself passes 65 504 (the fp32 oracle's activations reach ~1e5): the fp16 reference overflows, and so may the
    product -- the bar there is only "finite wherever the fp16 reference is".  Two references: the
    fp32 oracle, and the SAME oracle modules run in fp16 by PyTorch (torch_dtype=float16: what the reference's own fp16 UNet does,
    every module boundary rounded).  Bar: the product is finite and no further from the fp32 oracle than twice the fp16 oracle is
    (+ 2e-3): its arithmetic (fp16 storage, fp32 accumulation and statistics) must not be the less accurate of the two."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import LDMK_UNET, rel_l2, synthetic_inputs
    from mofa_video_amd import schema
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as Oracle
    cfg = LDMK_UNET
    sd = schema.synthetic_state_dict(schema.unet_schema(cfg), seed=21)
    g = torch.Generator().manual_seed(22)
    for k in list(sd):
        t = sd[k].float()
        if k.endswith(".weight") and t.dim() >= 2 and t.shape[0] >= 40 and "conv_out" not in k and "time_emb" not in k and "time_pos" not in k:
            rows = torch.randperm(t.shape[0], generator=g)[: max(t.shape[0] // 40, 1)]
            t[rows] *= OUTLIER
            if k.endswith("to_q.weight") or k.endswith("to_k.weight"):
                t *= 2.5
        elif k.endswith(".weight") and t.dim() == 1 and ("norm" in k):
            t = t * torch.exp(torch.randn(t.shape, generator=g) * 0.4)
        sd[k] = t.half()
    T, H, W = 3, 256, 256
    inp = synthetic_inputs(T, H, W, cross_dim=cfg["cross_attention_dim"], seed=23)
    lat = inp["latents"] * 5.0
    x = torch.cat([torch.cat([lat] * 2) / 10 ** 0.5, inp["image_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)], dim=2)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    tt = torch.tensor(0.8)
    boc, h, w = cfg["block_out_channels"], H // 8, W // 8
    shapes = [(boc[0], h, w)] * 3 + [(boc[0], h // 2, w // 2)] + [(boc[1], h // 2, w // 2)] * 2 + \
             [(boc[1], h // 4, w // 4)] + [(boc[2], h // 4, w // 4)] * 2 + [(boc[2], h // 8, w // 8)] + [(boc[3], h // 8, w // 8)] * 2
    res = [(torch.randn(2 * T, *s_, generator=g) * 0.3).half().to(DEV) for s_ in shapes]      # the adapter's residuals (F8: required)
    mid = (torch.randn(2 * T, boc[3], h // 8, w // 8, generator=g) * 0.3).half().to(DEV)
    ou = Oracle(**cfg).eval()
    ou.load_state_dict({k: v.float() for k, v in sd.items()})
    emb = inp["image_embeddings"].to(DEV)
    with torch.no_grad():
        ref32 = ou.to(DEV)(x.to(DEV), tt.to(DEV), emb, down_block_additional_residuals=[r.float() for r in res],
                           mid_block_additional_residual=mid.float(), return_dict=False, added_time_ids=ids.to(DEV))[0].float()
        ref16 = ou.half()(x.to(DEV).half(), tt.to(DEV), emb.half(), down_block_additional_residuals=res,
                          mid_block_additional_residual=mid, return_dict=False, added_time_ids=ids.to(DEV))[0].float()
    hu = UNetSpatioTemporalConditionControlNetModel(sd, cfg, DEV)
    got = hu(x.to(DEV), tt, emb, down_block_additional_residuals=[r.float() for r in res], mid_block_additional_residual=mid.float(),
             return_dict=False, added_time_ids=ids.to(DEV))[0].float()
    assert torch.isfinite(ref32).all()
    f16_ok, prod_ok = bool(torch.isfinite(ref16).all()), bool(torch.isfinite(got).all())
    e_prod = rel_l2(got, ref32) if prod_ok else float("inf")
    e_f16 = rel_l2(ref16, ref32) if f16_ok else float("inf")
    print(f"UNet forward, trained-like statistics (outliers x {OUTLIER:g}): product vs fp32 oracle {e_prod:.3e}; PyTorch-fp16 oracle vs fp32 "
          f"oracle {e_f16:.3e}; |out| max {ref32.abs().max().item():.2e}")
    if not f16_ok:
        return                                                      # beyond fp16 for the reference's own fp16 run: nothing to hold the product to
    assert prod_ok, "the product overflows where the reference's fp16 run does not"
    assert e_prod <= 2.0 * e_f16 + 2e-3, (e_prod, e_f16)
    assert e_prod < 3e-2, e_prod
