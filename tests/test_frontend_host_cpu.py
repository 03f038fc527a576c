"""Host logic of the image-conditioning front end without a GPU: the launch sequencing and weight repacking
of mofa_video_amd/{clip,vae,frontend,pipeline}.py run against tests/emu_ops.py (torch stand-ins for the HIP entry points,
test infrastructure) and must reproduce the oracle / the reference fixture.  What this pins: the 128-column head slots
and the masked key padding of the CLIP tower, quant_conv folded into conv_out, the trailing-pad downsample geometry, the
RNG order of the noise augmentation.  The kernels themselves are checked in tests/test_frontend_gpu.py."""
import os

import pytest
import torch

import emu_ops
from helpers import rel_l2
from mofa_video_amd import schema

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch)


def test_clip_host_logic_matches_transformers_fixture(emu):
    from mofa_video_amd.clip import CLIPVisionModelWithProjection
    from mofa_video_amd.frontend import encode_image
    G = torch.load(os.path.join(GD, "reference_golden_frontend.pt"), weights_only=False)["encode_image"]
    sd = schema.synthetic_state_dict(schema.clip_vision_schema(G["cfg"]), seed=G["seed"], dtype=torch.float32)
    enc = CLIPVisionModelWithProjection(sd, G["cfg"], "cpu")
    assert enc.qkv.shape == (264, 3 * 4 * 128)
    emb = encode_image(enc, G["image"])
    assert tuple(emb.shape) == (2, 1, 64) and torch.equal(emb[0], torch.zeros(1, 64))
    e = rel_l2(emb[1], G["image_embeddings"][1])
    assert e < 5e-3, e


def test_clip_key_padding_mask_is_exact(emu):
    """the padded key rows must not leak: junk in the padded rows of the hidden state leaves the result unchanged"""
    from mofa_video_amd.clip import CLIPVisionModelWithProjection
    cfg = dict(hidden_size=320, intermediate_size=640, num_hidden_layers=1, num_attention_heads=4, projection_dim=64)
    sd = schema.synthetic_state_dict(schema.clip_vision_schema(cfg), seed=5, dtype=torch.float32)
    enc = CLIPVisionModelWithProjection(sd, cfg, "cpu")
    pv = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    a = enc(pv).image_embeds.clone()
    C = 4 * 128
    enc.qkv[257:, 2 * C:] = 7.0                                # junk values in the padded V rows
    enc.qkv[257:, :C] = -3.0                                   # and in the padded Q rows
    b = enc(pv).image_embeds
    assert torch.equal(a, b)


def test_vae_encoder_host_logic_matches_oracle(emu):
    from mofa_video_amd.frontend import encode_vae_image
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    from oracle.frontend import encode_vae_image as oracle_encode
    from oracle.vae import AutoencoderKLTemporalDecoder as Oracle
    cfg = dict(block_out_channels=(64, 64, 128, 128))
    sd = schema.synthetic_state_dict(schema.vae_decoder_schema(**cfg), seed=40)
    sd.update(schema.synthetic_state_dict(schema.vae_encoder_schema(**cfg), seed=41))
    ref = Oracle(with_encoder=True, **cfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()})
    vae = AutoencoderKLTemporalDecoder(sd, cfg, "cpu")
    g = torch.Generator().manual_seed(8)
    img = torch.rand(1, 3, 64, 128, generator=g)
    noise = torch.randn(1, 3, 64, 128, generator=g)
    want = oracle_encode(ref, img * 2 - 1, noise)
    got = encode_vae_image(vae, img, noise=noise)
    assert tuple(got.shape) == tuple(want.shape) == (2, 4, 8, 16)
    assert rel_l2(got[1], want[1]) < 5e-3
    a = encode_vae_image(vae, img, generator=torch.Generator().manual_seed(9))
    b = encode_vae_image(vae, img, noise=torch.randn(1, 3, 64, 128, generator=torch.Generator().manual_seed(9)))
    assert torch.equal(a, b)


def test_resize_host_logic_matches_reference_fixture(emu):
    from mofa_video_amd.frontend import _resize_with_antialiasing
    G = torch.load(os.path.join(GD, "reference_golden_frontend.pt"), weights_only=False)["resize"]
    for name, c in G.items():
        out = _resize_with_antialiasing(c["x"], c["size"])
        assert (out - c["out"]).abs().max().item() < 2e-6, name


def test_pipeline_conditioning_rng_order_and_pil(emu):
    """the noise draw precedes the latent draw on the caller's generator (pipeline.py:340 before :379)"""
    import numpy as np
    from PIL import Image
    from mofa_video_amd.clip import CLIPVisionModelWithProjection
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    ccfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, projection_dim=64)
    vcfg = dict(block_out_channels=(64, 64))
    sdc = schema.synthetic_state_dict(schema.clip_vision_schema(ccfg), seed=60)
    sdv = schema.synthetic_state_dict(schema.vae_decoder_schema(**vcfg), seed=61)
    sdv.update(schema.synthetic_state_dict(schema.vae_encoder_schema(**vcfg), seed=62))
    pipe = FlowControlNetPipeline(vae=AutoencoderKLTemporalDecoder(sdv, vcfg, "cpu"),
                                  image_encoder=CLIPVisionModelWithProjection(sdc, ccfg, "cpu"),
                                  unet=type("U", (), {"device": torch.device("cpu")})())
    H, W = 64, 128
    img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(12))
    g = torch.Generator().manual_seed(13)
    emb, il = pipe._conditioning(img, None, None, H, W, 0.02, g)
    assert tuple(emb.shape) == (2, 1, 64) and tuple(il.shape) == (2, 4, H // 2, W // 2)
    after = torch.randn(3, generator=g)
    g2 = torch.Generator().manual_seed(13)
    torch.randn(1, 3, H, W, generator=g2)
    assert torch.equal(after, torch.randn(3, generator=g2))
    pil = Image.fromarray((img[0].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8))
    emb2, il2 = pipe._conditioning([pil], None, None, H, W, 0.02, torch.Generator().manual_seed(13))
    assert rel_l2(emb2, emb) < 2e-2 and rel_l2(il2, il) < 2e-2
    with pytest.raises(ValueError):
        pipe._conditioning(None, None, None, H, W, 0.02, None)
    # a tensor of another size is resized by nearest neighbour, as VaeImageProcessor.preprocess does for tensors
    big = torch.nn.functional.interpolate(img, size=(2 * H, 2 * W), mode="nearest")
    emb3, il3 = pipe._conditioning(big, None, None, H, W, 0.02, torch.Generator().manual_seed(13))
    assert rel_l2(il3, il) < 1e-6


def test_temb_batch_row_vector_addressing(emu):
    """TembBatch: all time_emb_proj layers in one GEMM; a layer's [B, Cout] slice of the [B, total] result is read as the
    igemm row vector through rv_mul = total / Cout (blocks.py).  Same numbers as one GEMM per layer."""
    from mofa_video_amd import ops
    from mofa_video_amd.blocks import BIG, Linear, Sub, TembBatch
    g = torch.Generator().manual_seed(0)
    sd = {}
    couts = [64, 128, 64, 320, 128]
    for i, n in enumerate(couts):
        sd[f"l{i}.weight"] = torch.randn(n, 256, generator=g) / 16
        sd[f"l{i}.bias"] = torch.randn(n, generator=g)
    ref = [Linear(Sub(sd, f"l{i}.", "cpu")) for i in range(len(couts))]
    with TembBatch() as tb:
        lins = [Linear(Sub(sd, f"l{i}.", "cpu")) for i in range(len(couts))]
        offs = [tb.add(l) for l in lins]
    assert offs == [0, 64, 192, 256, 576] and tb.w.shape[0] % 640 == 0 and tb.w.shape[0] >= sum(couts)   # lcm(64,128,320)
    assert all(l.w is None for l in lins)
    B, T, HW = 2, 3, 8
    temb = torch.randn(B, 256, generator=g).half()
    allv = tb.run(temb)
    assert allv.dtype == torch.float32 and tuple(allv.shape) == (B, tb.w.shape[0])
    x = torch.randn(B * T * HW, 64, generator=g).half()
    for lin, n, off in zip(ref, couts, offs):
        w = (torch.randn(n, 64, generator=g) / 8).half()
        want = ops.igemm(x, w, rowvec=ops.cast_f16_to_f32(lin(temb)), rv=(T * HW, 1, 1, BIG))
        got = ops.igemm(x, w, rowvec=allv[:, off:off + n], rv=(T * HW, allv.shape[1] // n, 1, BIG))
        assert torch.equal(want, got), n
