"""Host side of the fused level-0 feed-forward (mofa_ff320_f16, csrc/ff320.hip) without a GPU: the packed operand layout
that include/mofa_hip.h documents against weights.pack_ff320, and the wiring in blocks.TransformerSpatioTemporal (which norm is
folded into which feed-forward, the frame-position vector, the AlphaBlender residual, the second LayerNorm output) against the
unfused chain -- both on the torch stand-ins of tests/emu_ops.py (test infrastructure; the product has no CPU path).

Reference: diffusers 0.24.0 BasicTransformerBlock / TemporalBasicTransformerBlock feed-forward legs as the reference builds them
(MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:169-232), restated in oracle/blocks.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import emu_ops  # noqa: E402


def test_pack_ff320_matches_documented_layout():
    from mofa_video_amd.weights import pack_ff320
    g = torch.Generator().manual_seed(1)
    w1, b1, w2 = torch.randn(2560, 320, generator=g), torch.randn(2560, generator=g), torch.randn(320, 1280, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    w1p, b1f, w2p = pack_ff320(w1, b1, w2, gamma, beta)
    assert w1p.dtype == torch.float16 and w1p.numel() == 2560 * 320 and w2p.numel() == 320 * 1280 and b1f.dtype == torch.float32
    u1, u2 = emu_ops.unpack_ff320(w1p, w2p)
    assert torch.equal(u1, (w1 * gamma[None]).half()) and torch.equal(u2, w2.half())
    assert torch.allclose(b1f, b1 + w1 @ beta, atol=1e-4)


def test_pack_lin320_matches_documented_layout():
    from mofa_video_amd.weights import pack_lin320
    g = torch.Generator().manual_seed(4)
    w, b = torch.randn(960, 320, generator=g), torch.randn(960, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(320, generator=g), 0.3 * torch.randn(320, generator=g)
    wp, bp = pack_lin320(w, b, gamma, beta)
    assert wp.dtype == torch.float16 and tuple(wp.shape) == (15, 2, 20, 64, 8)
    assert torch.equal(emu_ops.unpack_lin320(wp), (w * gamma[None]).half())
    assert torch.allclose(bp, b + w @ beta, atol=1e-4)
    wp2, bp2 = pack_lin320(w[:320])
    assert bp2 is None and torch.equal(emu_ops.unpack_lin320(wp2), w[:320].half())


def test_transformer_block_fused_ff_equals_unfused(monkeypatch):
    """C = 320 transformer layer: level-0 feed-forwards fused (three ops.ff320 calls) against LayerNorm + two GEMMs each"""
    from mofa_video_amd import blocks, ops, schema
    emu_ops.install(monkeypatch)
    cfg = dict(block_out_channels=(320, 128, 256, 256), num_attention_heads=(5, 2, 4, 4), cross_attention_dim=128)
    sd = schema.synthetic_state_dict(schema.unet_schema(cfg), seed=11)
    pre = "down_blocks.0.attentions.0."
    sub = blocks.Sub({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, "", "cpu")
    xf = blocks.TransformerSpatioTemporal(sub, heads=5)
    assert xf.ff.pk is not None and xf.ff_in.pk is not None and xf.tff.pk is not None
    B, T, H, W = 2, 3, 4, 4
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(B * T * H * W, 320, generator=g) * 0.7).half()
    outs, calls = [], []
    real = ops.ff320
    monkeypatch.setattr(ops, "ff320", lambda *a, **k: (calls.append(sorted(k)), real(*a, **k))[1])
    for fused in (True, False):
        monkeypatch.setattr(ops, "FF_FUSED", fused)
        monkeypatch.setattr(ops, "LIN320", fused)                   # (the 320-channel projections with their norms folded, too)
        c = blocks.Ctx(B, T)
        c.ctx16 = (torch.randn(B, 128, generator=torch.Generator().manual_seed(3))).half()
        outs.append(xf(x, c, H, W).float())
    assert len(calls) == 3 and any("pos" in k and "ln_out" in k for k in calls) and any("r2" in k for k in calls), calls
    e = ((outs[0] - outs[1]).norm() / outs[1].norm()).item()
    assert e < 2e-3, e
