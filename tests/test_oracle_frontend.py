"""CPU tests of the front-end oracle: antialiased resize and _encode_image against the fixture produced by
the reference's own functions + transformers' CLIP class (tests/golden/make_golden_frontend.py); the VAE encoder
restatement against the published parameter count and its state_dict schema (diffusers is not available: unpinned)."""
import math
import os

import torch

from mofa_video_amd import schema

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _golden():
    return torch.load(os.path.join(GD, "reference_golden_frontend.pt"), weights_only=False)


def test_resize_with_antialiasing_matches_reference_fixture():
    from oracle.frontend import blur_geometry, resize_with_antialiasing
    G = _golden()["resize"]
    assert len(G) == 5
    for name, c in G.items():
        out = resize_with_antialiasing(c["x"], c["size"])
        assert tuple(out.shape) == tuple(c["out"].shape), name
        assert (out - c["out"]).abs().max().item() < 2e-6, name
    assert blur_geometry(576, 1024, (224, 224)) == ((3, 7), ((576 / 224 - 1) / 2, (1024 / 224 - 1) / 2))
    assert blur_geometry(100, 100, (224, 224)) == ((3, 3), (0.001, 0.001))


def test_encode_image_matches_reference_fixture():
    from oracle.clip import CLIPVisionModelWithProjection
    from oracle.frontend import encode_image, resize_with_antialiasing
    G = _golden()["encode_image"]
    enc = CLIPVisionModelWithProjection(G["cfg"]).eval()
    sd = schema.synthetic_state_dict(schema.clip_vision_schema(G["cfg"]), seed=G["seed"], dtype=torch.float32)
    enc.load_state_dict(sd, strict=True)
    emb = encode_image(enc, G["image"])
    assert tuple(emb.shape) == tuple(G["image_embeddings"].shape) == (2, 1, G["cfg"]["projection_dim"])
    assert torch.equal(emb[0], torch.zeros_like(emb[0]))
    assert (emb - G["image_embeddings"]).abs().max().item() < 2e-5 * G["image_embeddings"].abs().max().item() + 1e-6
    with torch.no_grad():                                            # the trunk before pooling, two token rows
        x = enc.vision_model.pre_layrnorm(enc.vision_model.embeddings(resize_with_antialiasing(G["image"], (224, 224))))
        for layer in enc.vision_model.encoder.layers:
            x = layer(x)
    assert (x[:, 0] - G["last_hidden_state_cls"]).abs().max().item() < 1e-4
    assert (x[:, 200] - G["last_hidden_state_tok200"]).abs().max().item() < 1e-4


def test_clip_schema_is_vit_h():
    s = schema.clip_vision_schema()
    assert sum(math.prod(v) for v in s.values()) == 632_076_800          # OpenCLIP ViT-H/14 vision tower + projection
    assert s["vision_model.embeddings.position_embedding.weight"] == (257, 1280)
    assert "vision_model.pre_layrnorm.weight" in s and s["visual_projection.weight"] == (1024, 1280)


def test_vae_encoder_structure():
    from oracle.frontend import encode_vae_image
    from oracle.vae import AutoencoderKLTemporalDecoder
    vae = AutoencoderKLTemporalDecoder(with_encoder=True).eval()
    got = {k: tuple(v.shape) for k, v in vae.state_dict().items() if not k.startswith("decoder.")}
    assert got == schema.vae_encoder_schema()
    assert sum(math.prod(v) for v in got.values()) == 34_163_664       # SD / SVD VAE encoder 34 163 592 + quant_conv 72
    small = AutoencoderKLTemporalDecoder(block_out_channels=(32, 64), layers_per_block=1, with_encoder=True).eval()
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 3, 18, 22, generator=g) * 2 - 1
    noise = torch.randn(1, 3, 18, 22, generator=g)
    lat = encode_vae_image(small, x, noise)
    assert tuple(lat.shape) == (2, 4, 9, 11) and torch.equal(lat[0], torch.zeros_like(lat[0]))
    with torch.no_grad():                                               # mode() is the mean half of quant_conv's output
        m = small.quant_conv(small.encoder(x + 0.02 * noise))
    assert torch.equal(lat[1], m[0, :4])
    # Downsample2D(padding=0) pads only after the last row / column
    conv = small.encoder.down_blocks[0].downsamplers[0].conv
    t = torch.rand(1, 32, 6, 8, generator=g)
    with torch.no_grad():
        a = small.encoder.down_blocks[0].downsamplers[0].conv(torch.nn.functional.pad(t, (0, 1, 0, 1)))
        b = torch.nn.functional.conv2d(torch.nn.functional.pad(t, (1, 1, 1, 1)), conv.weight, conv.bias, stride=2)
    assert tuple(a.shape) == (1, 32, 3, 4) and not torch.allclose(a, b)
