"""Full-geometry, full-width parity: the HIP path against the ORACLE RUN ON THE GPU IN FP32 (test-side only) on the same
seeded weights and inputs, at the sizes the bench runs -- 25 frames, 576x1024, the full SVD-XT widths (round-2 verdict,
"What's weak" 1-3).  The CPU oracle cannot finish these in test time; the very same ``oracle/`` modules can, moved to
``cuda`` with MIOpen switched off (ATen's im2col + rocBLAS sgemm convolutions, no reduced-precision path) and the
attention evaluated in exact batch chunks (``oracle.blocks.SDPA`` hook: the materialised fp32 scores of all 50 frames at
S = 9216 would be 85 GB).  The product never sees any of this: the oracle stays the checker.

  (a) BASELINE config 2, ONE denoise step at 25 f 576x1024: the 12 + 1 ControlNet residuals and the UNet noise prediction;
  (b) one 8-frame temporal-VAE chunk at 576x1024;
  (c) config 3, the landmark adapter at FULL width (hourglass in-channels 642 / 1282 / 2562, 7x7 matting heads, zero-outs,
      occlusion masks) at 25 f 576x1024;
  (d) config 4, the Hybrid dual-adapter blend at full width: one denoise step of the whole pipeline;
  (e) kernels at bench size: GroupNorm spanning 8 x 589 824 positions per group (C = 128), softsplat 72x128x320 with the
      bench's 64-px flow, temporal attention T = 25 / HW = 9 216, spatial attention with 64 queries per wave at S = 9 216 and
      at the ragged S = 9 176, and the K/V tail tile with NaN-poisoned LDS and neighbours (round-2 advice).

Stated fp16 tolerance (fp16 storage, fp32 accumulate, against fp32): rel-L2 <= 1e-2 per forward tensor, <= 2e-2 for
latents after a step and decoded frames.  Every test prints the rel-L2 it measured.
"""
import contextlib

import pytest
import torch
import torch.nn.functional as F

import bench

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, H, W = bench.T, bench.H, bench.W


def rel(a, b):
    a, b = a.to(DEV, torch.float32), b.to(DEV, torch.float32)
    assert tuple(a.shape) == tuple(b.shape), (a.shape, b.shape)
    assert bool(torch.isfinite(a).all()), "non-finite product output"
    return float((a - b).norm() / (b.norm() + 1e-12))


def chunked_sdpa(q, k, v):
    """softmax(q k^T / sqrt(d)) v in fp32, at most 2 GiB of scores at a time (exactly the formula of F.scaled_dot_product_attention)"""
    B, h, S, d = q.shape
    per = h * S * k.shape[2] * 4
    step = max(1, (1 << 31) // max(per, 1))
    out = torch.empty_like(q)
    for b0 in range(0, B, step):
        sc = torch.matmul(q[b0:b0 + step], k[b0:b0 + step].transpose(-1, -2)) * d ** -0.5
        out[b0:b0 + step] = torch.matmul(torch.softmax(sc, -1), v[b0:b0 + step])
        del sc
    return out


@contextlib.contextmanager
def exact_fp32_gpu():
    import oracle.blocks as ob
    old = (torch.backends.cudnn.enabled, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32, ob.SDPA)
    torch.backends.cudnn.enabled = False            # ROCm: no MIOpen -> ATen slow_conv2d / slow_conv_dilated3d (fp32 sgemm)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    ob.SDPA = chunked_sdpa
    try:
        with torch.no_grad():
            yield
    finally:
        torch.backends.cudnn.enabled, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.allow_tf32, ob.SDPA = old
        torch.cuda.empty_cache()


def gpu_oracle(cls, sd, **kw):
    """an oracle module on the GPU in fp32 holding the (fp16-valued) weights of ``sd``; built on the meta device so that the
    1.5 G parameters are never initialised on the host"""
    with torch.device("meta"):
        m = cls(**kw)
    m = m.to_empty(device=DEV)
    missing, unexpected = m.load_state_dict({k: v.to(DEV, torch.float32) for k, v in sd.items()}, strict=True)
    assert not missing and not unexpected
    return m.eval()


def model_input(inp, sigma=3.0):
    """what the loop feeds the networks at a mid-schedule sigma: [2, T, 8, h, w] (uncond half first, pipeline.py:451-457)"""
    lat = inp["latents"] * 5.0
    x = torch.cat([lat] * 2) / (sigma ** 2 + 1) ** 0.5
    il = torch.cat([torch.zeros_like(inp["image_latents"]), inp["image_latents"]]).unsqueeze(1).repeat(1, T, 1, 1, 1)
    return torch.cat([x, il], dim=2)


@pytest.fixture(scope="module")
def inp():
    return bench.synthetic_inputs(torch.device(DEV))


@pytest.fixture(scope="module")
def sds():
    """seeded fp16 state dicts in the reference checkpoint layout, on the GPU, shared by both sides (bench.py's seeds)"""
    from mofa_video_amd import schema
    mk = lambda sch, seed: schema.synthetic_state_dict(sch, seed=seed, device=DEV)   # noqa: E731
    return dict(unet=mk(schema.unet_schema(), 0), cn=mk(schema.controlnet_schema(), 1), vae=mk(schema.vae_decoder_schema(), 2),
                ldmk=mk(schema.ldmk_controlnet_schema(), 7))


@pytest.fixture(scope="module")
def hip(sds):
    from mofa_video_amd.adapter import FlowControlNet, LandmarkFlowControlNet
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    return dict(unet=UNetSpatioTemporalConditionControlNetModel(sds["unet"], None, DEV), cn=FlowControlNet(sds["cn"], None, DEV),
                vae=AutoencoderKLTemporalDecoder(sds["vae"], None, DEV), ldmk=LandmarkFlowControlNet(sds["ldmk"], None, DEV))


def _emb2(inp):
    return torch.cat([torch.zeros_like(inp["image_embeddings"]), inp["image_embeddings"]])


# ---------------------------------------------------------------------------------------------------------------------
def test_config2_one_denoise_step_at_25f_576x1024(inp, sds, hip):
    """(a) MOFA-Video-Traj/pipeline/pipeline.py:447-511, one iteration: FlowControlNet.forward then the UNet forward on its
    residuals, both CFG halves, 50 frames of 72x128 latents"""
    from oracle.controlnet import FlowControlNet as OCn
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    x, emb = model_input(inp), _emb2(inp)
    t = torch.tensor(0.8)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, device=DEV)
    cond2, flow2 = torch.cat([inp["cond"]] * 2), torch.cat([inp["flow"]] * 2)
    with exact_fp32_gpu():
        oc = gpu_oracle(OCn, sds["cn"])
        rd, rm, _, _ = oc(x, t.to(DEV), emb, ids, controlnet_cond=cond2, controlnet_flow=flow2, return_dict=False,
                          conditioning_scale=1.0)
        del oc
        ou = gpu_oracle(OUnet, sds["unet"])
        assert sum(p.numel() for p in ou.parameters()) == 1524623082
        ref = ou(x, t.to(DEV), emb, down_block_additional_residuals=rd, mid_block_additional_residual=rm, return_dict=False,
                 added_time_ids=ids)[0]
        del ou
    gd, gm, _, _ = hip["cn"](x, t, emb, ids, controlnet_cond=cond2, controlnet_flow=flow2, return_dict=False,
                             conditioning_scale=1.0)
    assert len(gd) == len(rd) == 12
    worst = 0.0
    for i, (r, g) in enumerate(zip(list(rd) + [rm], list(gd) + [gm])):
        e = rel(g, r)
        worst = max(worst, e)
        print(f"config 2 @ 25f 576x1024, ControlNet residual {i} {tuple(r.shape)}: rel-L2 {e:.3e}")
        assert e < 1e-2, (i, e)
    # the UNet on the ORACLE's residuals (isolates it), then the chain as the loop runs it (its own adapter's residuals)
    got_iso = hip["unet"](x, t, emb, down_block_additional_residuals=rd, mid_block_additional_residual=rm, return_dict=False,
                          added_time_ids=ids)[0]
    e_iso = rel(got_iso, ref)
    got = hip["unet"](x, t, emb, down_block_additional_residuals=gd, mid_block_additional_residual=gm, return_dict=False,
                      added_time_ids=ids)[0]
    e = rel(got, ref)
    print(f"config 2 @ 25f 576x1024, UNet noise prediction {tuple(ref.shape)}: rel-L2 {e_iso:.3e} (oracle residuals), "
          f"{e:.3e} (adapter -> UNet chain); worst residual {worst:.3e}")
    assert tuple(got.shape) == (2, T, 4, H // 8, W // 8)
    assert e_iso < 1e-2 and e < 1e-2, (e_iso, e)


def test_vae_chunk_8_frames_at_576x1024(inp, sds, hip):
    """(b) decode_latents' unit of work (pipeline.py:194-220): one chunk of 8 frames through the temporal decoder"""
    from oracle.vae import AutoencoderKLTemporalDecoder as OVae
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(bench.CHUNK, 4, H // 8, W // 8, generator=g) * 0.18215 * 4.0).to(DEV)
    with exact_fp32_gpu():
        ov = gpu_oracle(OVae, sds["vae"])
        ref = ov.decode(z / ov.scaling_factor, num_frames=bench.CHUNK)
        del ov
    got = hip["vae"].decode(z, num_frames=bench.CHUNK, _prescale=1.0 / 0.18215)
    e = rel(got, ref)
    print(f"VAE chunk {tuple(ref.shape)}: decoded frames rel-L2 {e:.3e}")
    assert tuple(got.shape) == (bench.CHUNK, 3, H, W)
    assert e < 2e-2, e


def test_config3_landmark_adapter_full_width(inp, sds, hip):
    """(c) MOFA-Video-Hybrid/models/ldmk_ctrlnet.py:291-320 (warp -> ForegroundMatting -> zero_out) and :387-451 (forward) at
    the default (full) widths, 25 f 576x1024: residuals and the occlusion masks of every scale"""
    from oracle.ldmk import LandmarkFlowControlNet as OLdmk
    lmk = bench.synthetic_landmark_inputs(torch.device(DEV))
    x, emb = model_input(inp), _emb2(inp)
    t = torch.tensor(0.8)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, device=DEV)
    cond2, flow2, lm2 = torch.cat([inp["cond"]] * 2), torch.cat([lmk["flow"]] * 2), torch.cat([lmk["landmarks"]] * 2)
    with exact_fp32_gpu():
        of = gpu_oracle(OLdmk, sds["ldmk"])
        assert sum(p.numel() for p in of.parameters()) == 722453685
        rd, rm, _, rocc = of(x, t.to(DEV), emb, ids, controlnet_cond=cond2, controlnet_flow=flow2, landmarks=lm2,
                             return_dict=False, conditioning_scale=0.9)
        del of
    gd, gm, _, gocc = hip["ldmk"](x, t, emb, ids, controlnet_cond=cond2, controlnet_flow=flow2, landmarks=lm2, return_dict=False,
                                  conditioning_scale=0.9)
    for i, (r, g) in enumerate(zip(list(rd) + [rm], list(gd) + [gm])):
        e = rel(g, r)
        print(f"config 3 full width @ 25f 576x1024, landmark-adapter residual {i} {tuple(r.shape)}: rel-L2 {e:.3e}")
        assert e < 1e-2, (i, e)
    assert len(gocc) == len(rocc) >= 3
    for lvl, (r, g) in enumerate(zip(rocc, gocc)):
        e = rel(g, r)
        print(f"config 3 full width, occlusion mask level {lvl} {tuple(r.shape)}: rel-L2 {e:.3e}")
        assert e < 1e-2, (lvl, e)


def test_config4_hybrid_blend_full_width_one_step(inp, sds, hip):
    """(d) MOFA-Video-Hybrid/pipeline/pipeline.py:443-507: face + drag adapters, residuals blended by the user mask at every
    scale (1280 channels included), UNet, CFG + Euler -- one step of the whole call at 25 f 576x1024"""
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.controlnet import FlowControlNet as OCn
    from oracle.ldmk import LandmarkFlowControlNet as OLdmk
    from oracle.pipeline import denoise_hybrid
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    lmk = bench.synthetic_landmark_inputs(torch.device(DEV))
    mask = bench.synthetic_mask(torch.device(DEV))
    il2, emb = torch.cat([torch.zeros_like(inp["image_latents"]), inp["image_latents"]]), _emb2(inp)
    with exact_fp32_gpu():
        ou, of, od = gpu_oracle(OUnet, sds["unet"]), gpu_oracle(OLdmk, sds["ldmk"]), gpu_oracle(OCn, sds["cn"])
        ref = denoise_hybrid(ou, of, od, OSch(), inp["latents"], il2, emb, inp["cond"], lmk["flow"], lmk["landmarks"],
                             inp["flow"], mask, num_inference_steps=1, ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1)
        del ou, of, od
    pipe = HybridFlowControlNetPipeline(unet=hip["unet"], face_controlnet=hip["ldmk"], drag_controlnet=hip["cn"],
                                        scheduler=EulerDiscreteScheduler())
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=lmk["flow"], landmarks=lmk["landmarks"],
               drag_flow=inp["flow"], mask=mask, height=H, width=W, num_frames=T, num_inference_steps=1,
               latents=inp["latents"], output_type="latent", ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1,
               image_embeddings=emb, image_latents=il2).frames
    e = rel(out, ref)
    print(f"config 4 full width @ 25f 576x1024: Hybrid latents after one step {tuple(ref.shape)}: rel-L2 {e:.3e}")
    assert e < 2e-2, e


# ---- (e) kernels at bench size -----------------------------------------------------------------------------------------
def test_group_norm_temporal_vae_size():
    """TemporalResnetBlock statistics of the VAE's last level: one group = 8 frames x 589 824 positions x 4 channels =
    18.9 M values (fp32 partial sums, fp64 combine) against float64 on the GPU"""
    from mofa_video_amd import ops
    frames, HW, C = 8, 576 * 1024, 128
    g = torch.Generator(device=DEV).manual_seed(25)
    x = (torch.randn(frames * HW, C, generator=g, device=DEV) * 1.5 + 0.75).half()
    gam = torch.randn(C, generator=g, device=DEV)
    bet = torch.randn(C, generator=g, device=DEV)
    for silu in (False, True):
        out = ops.group_norm(x, gam, bet, frames, HW, 1e-6, frames_per_stat=frames, silu=silu)
        ref = torch.empty(frames * HW, C, dtype=torch.float32, device=DEV)
        for gi in range(32):                                    # float64, one group at a time (memory)
            cs = slice(gi * 4, gi * 4 + 4)
            xg = x[:, cs].double()
            m, v = xg.mean(), xg.var(unbiased=False)
            y = (xg - m) / torch.sqrt(v + 1e-6) * gam[cs].double() + bet[cs].double()
            ref[:, cs] = (F.silu(y) if silu else y).float()
        e = rel(out, ref)
        err = (out.float() - ref).abs().max().item()
        print(f"GroupNorm 8 x 589824 x 128 (silu={silu}): rel-L2 {e:.3e}, max abs err {err:.3e}")
        assert e < 2e-3 and err < 2e-2, (e, err)


def test_softsplat_bench_size(inp):
    """the level-0 warp of the bench clip: 24 flows of up to 64 px at 72x128, 320 channels, against the oracle (on the GPU)"""
    from mofa_video_amd import ops
    from oracle.softsplat import softsplat
    h, w, C = H // 8, W // 8, 320
    g = torch.Generator().manual_seed(40)
    feat = torch.randn(1, C, h, w, generator=g).half().to(DEV)
    flow = ops.flow_downscale(inp["flow"][0].contiguous(), 8)               # [24, 2, 72, 128], values / 8 (the adapter's own)
    assert flow.abs().max().item() > 7.0
    tok = feat[0].permute(1, 2, 0).reshape(h * w, C).contiguous()
    out = ops.softsplat_avg_tokens(tok, flow, h, w)
    ref = torch.stack([softsplat(feat.float(), flow[i:i + 1], None, "avg")[0] for i in range(flow.shape[0])])
    ref = ref.permute(0, 2, 3, 1).reshape(-1, C)
    e = rel(out, ref)
    err = (out.float() - ref).abs().max().item()
    print(f"softsplat 24 x 72x128x320, bench flow: rel-L2 {e:.3e}, max abs err {err:.3e}")
    assert e < 2e-3 and err < 2e-2, (e, err)
    assert torch.equal(out, ops.softsplat_avg_tokens(tok, flow, h, w))


def test_attn_temporal_bench_size():
    from mofa_video_amd import ops
    clips, Tt, HW, heads, hd = 2, 25, 9216, 5, 64
    Cc = heads * hd
    g = torch.Generator(device=DEV).manual_seed(22)
    qkv = torch.randn(clips * Tt * HW, 3 * Cc, generator=g, device=DEV).half()
    out = ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], clips, Tt, HW, heads, head_dim=hd)
    q, k, v = [t.float().reshape(clips, Tt, HW, heads, hd).permute(0, 2, 3, 1, 4) for t in qkv.split(Cc, dim=1)]
    ref = chunked_sdpa(q.reshape(-1, 1, Tt, hd), k.reshape(-1, 1, Tt, hd), v.reshape(-1, 1, Tt, hd))
    ref = ref.reshape(clips, HW, heads, Tt, hd).permute(0, 3, 1, 2, 4).reshape(clips * Tt * HW, Cc)
    e = rel(out, ref)
    err = (out.float() - ref).abs().max().item()
    print(f"temporal attention T=25 HW=9216 heads=5: rel-L2 {e:.3e}, max abs err {err:.3e}")
    assert e < 2e-3 and err < 4e-3 * ref.abs().max().item() * 2, (e, err)


@pytest.mark.parametrize("S", [9216, 9176])
@pytest.mark.parametrize("qb", [2, 1])
def test_attn_spatial_bench_size(S, qb):
    """the level-0 shape (5 heads x 64, S = 72 x 128) and a ragged S whose last 256-query block and last key tile are partial,
    with 64 (qb = 2, the bench's kernel) and 32 queries per wave"""
    from mofa_video_amd import ops
    frames, heads, hd = 8, 5, 64
    Cc = heads * hd
    g = torch.Generator(device=DEV).manual_seed(20)
    qkv = torch.randn(frames * S, 3 * Cc, generator=g, device=DEV).half()
    out = ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], frames, heads, S, head_dim=hd, query_blocks=qb)
    q, k, v = [t.float().reshape(frames, S, heads, hd).transpose(1, 2) for t in qkv.split(Cc, dim=1)]
    ref = chunked_sdpa(q, k, v).transpose(1, 2).reshape(frames * S, Cc)
    e = rel(out, ref)
    err = (out.float() - ref).abs().max().item()
    print(f"spatial attention S={S} qb={qb}: rel-L2 {e:.3e}, max abs err {err:.3e}")
    assert e < 4e-3 and err < 4e-3 * (ref.abs().max().item() + 1.0), (e, err)


@pytest.mark.parametrize("S,hd", [(40, 64), (200, 64), (9176, 64), (40, 128)])
def test_attn_spatial_tail_rows_with_poisoned_lds_and_neighbours(S, hd):
    """round-2 advice: the d = 64 LDS-DMA path relies on the buffer descriptor's bounds check zero-filling K / V rows past S.
    Here the rows after every frame's tensor are NaN in memory (the frames sit in a NaN-filled buffer with a gap between
    them) and the CU's LDS was filled with NaN by the preceding launch (an implicit GEMM on NaN operands: its K-tile ring
    covers the attention kernel's K / V buffers); any lane that skipped its zero write would surface as NaN (0 * NaN)"""
    from mofa_video_amd import ops
    frames, heads = 3, 5
    Cc = heads * hd
    gap = 77
    g = torch.Generator(device=DEV).manual_seed(33)
    buf = torch.full((frames * (S + gap) + gap, 3 * Cc), float("nan"), dtype=torch.float16, device=DEV)
    data = torch.randn(frames, S, 3 * Cc, generator=g, device=DEV).half()
    outs = []
    poison_x = torch.full((65536, 256), float("nan"), dtype=torch.float16, device=DEV)
    poison_w = torch.full((256, 256), float("nan"), dtype=torch.float16, device=DEV)
    for f in range(frames):                                                  # one launch per frame: each frame's tensor ends in NaN rows
        r0 = gap + f * (S + gap)
        buf[r0:r0 + S] = data[f]
        ops.igemm(poison_x, poison_w)
        v = buf[r0:r0 + S]
        outs.append(ops.attn_spatial(v[:, :Cc], v[:, Cc:2 * Cc], v[:, 2 * Cc:], 1, heads, S, head_dim=hd))
    out = torch.cat(outs, 0)
    q, k, v = [t.float().reshape(frames, S, heads, hd).transpose(1, 2) for t in data.reshape(frames * S, 3 * Cc).split(Cc, dim=1)]
    ref = chunked_sdpa(q, k, v).transpose(1, 2).reshape(frames * S, Cc)
    e = rel(out, ref)
    print(f"spatial attention S={S} d={hd}, NaN-poisoned LDS and neighbours: rel-L2 {e:.3e}")
    assert e < 4e-3, e
