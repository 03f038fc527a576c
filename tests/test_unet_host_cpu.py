"""The host-side block graph of the UNet and the ControlNet trunk (mofa_video_amd/{blocks,unet,adapter}.py: launch sequencing, weight
repacking, epilogue fusions, the residual quirk F8, the copy-free decoder concat buffers) WITHOUT a GPU: every HIP entry point is
replaced by its torch stand-in (tests/emu_ops.py, test infrastructure) and the result is compared with the oracle on the reduced
configuration.  The product has no CPU path: this checks the Python that decides WHAT is launched, the GPU suite checks the kernels.

Reference: MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:356-504 (forward, residual re-zip :434-459).
"""
import pytest
import torch

import emu_ops
from helpers import TINY, TINY_CN, oracle_models, rel_l2, synthetic_inputs

T, H, W = 3, 256, 256


@pytest.fixture
def emu(monkeypatch):
    emu_ops.install(monkeypatch)
    from mofa_video_amd import ops
    # (the stand-ins have no launch to time and no scratch to hand over)
    monkeypatch.setattr(ops, "TIMER", None)
    return ops


@pytest.fixture(scope="module")
def oracle():
    torch.manual_seed(0)
    return oracle_models(TINY, seed=3, cn_cfg=TINY_CN)


def _model_input(inp):
    lat = torch.cat([inp["latents"]] * 2)
    il = inp["image_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)
    return torch.cat([lat, il], dim=2)


def test_unet_host_graph_matches_oracle(emu, oracle):
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    ou, oc, ov, sdu, sdc, sdv = oracle
    hu = UNetSpatioTemporalConditionControlNetModel(sdu, config=TINY, device="cpu")
    inp = synthetic_inputs(T, H, W, cross_dim=TINY["cross_attention_dim"], seed=7)
    x = _model_input(inp)
    t = torch.tensor(0.8)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    boc = TINY["block_out_channels"]
    h, w = H // 8, W // 8
    shapes = [(boc[0], h, w)] * 3 + [(boc[0], h // 2, w // 2)] + [(boc[1], h // 2, w // 2)] * 2 + \
             [(boc[1], h // 4, w // 4)] + [(boc[2], h // 4, w // 4)] * 2 + [(boc[2], h // 8, w // 8)] + \
             [(boc[3], h // 8, w // 8)] * 2
    g = torch.Generator().manual_seed(11)
    res = [(torch.randn(2 * T, *s, generator=g) * 0.3).half().float() for s in shapes]
    mid = (torch.randn(2 * T, boc[3], h // 8, w // 8, generator=g) * 0.3).half().float()
    with torch.no_grad():
        ref = ou(x, t, inp["image_embeddings"], down_block_additional_residuals=res, mid_block_additional_residual=mid,
                 return_dict=False, added_time_ids=ids)[0]
    keep = [r.clone() for r in res]
    got = hu(x, t, inp["image_embeddings"], down_block_additional_residuals=res, mid_block_additional_residual=mid,
             return_dict=False, added_time_ids=ids)[0]
    e = rel_l2(got, ref)
    assert tuple(got.shape) == tuple(ref.shape) == (2, T, 4, h, w)
    assert e < 1e-2, e
    assert all(torch.equal(a, b) for a, b in zip(keep, res)), "the caller's residual tensors must stay untouched"


def test_decoder_leaves_encoder_tensors_untouched_and_is_repeatable(emu, oracle):
    """decode_tokens writes skip + multiplicity x residual into the concat buffers: the encoder's outputs (sample, skips) are read
    only, so two decodes of one encoder result agree (the split-decoder order of the pipeline relies on it)"""
    from mofa_video_amd import ops
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    ou, oc, ov, sdu, sdc, sdv = oracle
    hu = UNetSpatioTemporalConditionControlNetModel(sdu, config=TINY, device="cpu")
    inp = synthetic_inputs(T, H, W, cross_dim=TINY["cross_attention_dim"], seed=8)
    x = _model_input(inp)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    h, w = H // 8, W // 8
    c = hu.make_ctx(0.5, inp["image_embeddings"], ids, 2, T)
    xt = ops.nchw_to_tokens(x.reshape(2 * T, 8, h, w), ld=hu.in_ld)
    enc = hu.encode_tokens(xt, c, h, w)
    sample, skips, counts, Hm, Wm = enc
    g = torch.Generator().manual_seed(12)
    down = [(torch.randn(k.shape, generator=g) * 0.3).half() for k in skips]
    mid = (torch.randn(sample.shape, generator=g) * 0.3).half()
    before = [k.clone() for k in skips] + [sample.clone()]
    a = hu.decode_tokens(enc, c, down, mid)
    b = hu.decode_tokens(enc, c, down, mid)
    assert torch.equal(a, b)
    assert all(torch.equal(p, q) for p, q in zip(before, list(skips) + [sample]))
    # the multiplicities of the reference's re-zip: skip i gets residual i once per remaining down block
    from mofa_video_amd.unet import residual_multiplicity
    assert residual_multiplicity(counts, len(down)) == [4, 4, 4, 4, 3, 3, 3, 2, 2, 2, 1, 1]


def test_controlnet_trunk_host_graph_matches_oracle(emu, oracle):
    from mofa_video_amd import ops
    from mofa_video_amd.adapter import FlowControlNet
    ou, oc, ov, sdu, sdc, sdv = oracle
    hc = FlowControlNet(sdc, config=TINY_CN, device="cpu")
    inp = synthetic_inputs(T, H, W, cross_dim=TINY["cross_attention_dim"], seed=9)
    x = _model_input(inp)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    h, w = H // 8, W // 8
    cond2, flow2 = torch.cat([inp["cond"]] * 2), torch.cat([inp["flow"]] * 2)
    with torch.no_grad():
        rd, rm, _, _ = oc(x, torch.tensor(0.8), inp["image_embeddings"], ids, controlnet_cond=cond2, controlnet_flow=flow2,
                          return_dict=False)
        # the adapter's timestep-invariant state (condition CNN, pyramids, forward-splat warps) from the oracle's own modules:
        # the warp kernels have no stand-in, and this test is about the trunk's launch graph
        warped_ref = oc.warped_cond_features(inp["cond"], inp["flow"])          # 4 x [T, C_l, h_l, w_l] (one CFG half: shared)
    warped = [wr.permute(0, 2, 3, 1).reshape(-1, wr.shape[1]).half() for wr in warped_ref]
    c = hc.make_ctx(0.8, inp["image_embeddings"], ids, 2, T)
    xt = ops.nchw_to_tokens(x.reshape(2 * T, 8, h, w), ld=hc.in_ld)
    gd, gm = hc.forward_tokens(xt, c, h, w, warped, 1.0)
    r = rm.permute(0, 2, 3, 1).reshape(gm.shape[0], -1)
    assert rel_l2(gm, r) < 1e-2
    assert len(gd) == len(rd) == 12
    for got, ref in zip(gd, rd):
        assert rel_l2(got, ref.permute(0, 2, 3, 1).reshape(got.shape[0], -1)) < 1e-2
