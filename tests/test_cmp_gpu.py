"""GPU parity of the CMP sparse-to-dense motion encoder through the C ABI against the CPU oracle on the same
seeded weights (the oracle itself is pinned to the reference's classes by tests/test_oracle_cmp.py).
Stated fp16 tolerance: rel-L2 <= 2e-2 on the 198-bin logits and on the flow (fp16 storage, BatchNorm folded into fp16
weights, 50+ layers); the elementwise kernels are checked against torch within fp16 rounding."""
import pytest
import torch
import torch.nn.functional as F

from helpers import rel_l2
from test_oracle_cmp import cmp_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tok(x):                       # [n,C,H,W] fp32 -> token-major fp16 on the device
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).half().to(DEV).contiguous()


def _untok(t, n, c, h, w):
    return t.float().cpu().reshape(n, h, w, -1)[..., :c].permute(0, 3, 1, 2)


@pytest.mark.parametrize("k,stride,pad,mode", [(3, 2, 1, "max"), (2, 2, 0, "max"), (8, 8, 0, "max"), (2, 2, 0, "avg")])
def test_pool2d(k, stride, pad, mode):
    from mofa_video_amd import ops
    x = torch.randn(2, 64, 24, 40).half().float()
    y, ho, wo = ops.pool2d(_tok(x), 2, 24, 40, 64, k, stride, pad=pad, mode=mode)
    ref = F.max_pool2d(x, k, stride, pad) if mode == "max" else F.avg_pool2d(x, k, stride)
    assert (ho, wo) == tuple(ref.shape[2:])
    assert torch.allclose(_untok(y, 2, 64, ho, wo), ref.half().float(), atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("hw", [((6, 10), (12, 20)), ((2, 3), (16, 24)), ((16, 24), (32, 48))])
def test_resize_bilinear_align_corners(hw):
    from mofa_video_amd import ops
    (h, w), (ho, wo) = hw
    x = torch.randn(3, 128, h, w).half().float()
    y = ops.resize_bilinear_ac(_tok(x), 3, h, w, 128, ho, wo)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True)
    assert torch.allclose(_untok(y, 3, 128, ho, wo), ref, atol=2e-3, rtol=2e-3)
    f = torch.randn(2, 2, h, w)
    assert torch.allclose(ops.resize_bilinear_ac_f32(f.to(DEV), ho, wo).cpu(),
                          F.interpolate(f, size=(ho, wo), mode="bilinear", align_corners=True), atol=1e-5, rtol=1e-5)


def test_flow_expectation():
    from mofa_video_amd import ops
    from oracle.cmp import Fuser
    logits = (torch.randn(2, 198, 12, 20) * 3).half().float()
    buf = torch.zeros(2 * 12 * 20, 256, dtype=torch.float16, device=DEV)
    buf[:, :198] = _tok(logits)
    got = ops.flow_expectation(buf, 2, 12, 20, 99, 50).cpu()
    assert torch.allclose(got, Fuser(99, 50).convert_flow(logits), atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("ksize,stride,dil", [(3, 1, 2), (3, 1, 4), (5, 2, 1), (1, 2, 1), (7, 2, 1)])
def test_conv_dilation_and_kernel_sizes(ksize, stride, dil):
    from mofa_video_amd import lib as L
    from mofa_video_amd import ops
    from mofa_video_amd import weights as Wt
    n, c, h, w, nout = 2, 64, 24, 40, 128
    x = torch.randn(n, c, h, w).half().float()
    wt = (torch.randn(nout, c, ksize, ksize) * 0.05).half().float()
    b = torch.randn(nout)
    g = ops.conv3x3_geom(h, w, stride=stride, ksize=ksize, dil=dil)
    y = ops.igemm(_tok(x), Wt.pack_conv3x3(wt).to(DEV), bias=b.to(DEV), geom=g, act=L.ACT_RELU)
    ref = F.relu(F.conv2d(x, wt, b, stride=stride, padding=dil * (ksize // 2), dilation=dil))
    assert (g.Hout, g.Wout) == tuple(ref.shape[2:])
    assert rel_l2(_untok(y, n, nout, g.Hout, g.Wout), ref) < 2e-3


@pytest.fixture(scope="module")
def models():
    from mofa_video_amd import schema
    from mofa_video_amd.cmp import CMP_demo
    from oracle.cmp import CMPDemo
    sd = schema.synthetic_state_dict(schema.cmp_schema(), seed=21, gain=2.0)
    o = CMPDemo()
    o.model.load_state_dict({k: (t if t.dtype == torch.long else t.float()) for k, t in sd.items()})
    return o, CMP_demo(sd, DEV)


def test_cmp_run_vs_oracle(models):
    o, hm = models
    image, sparse, mask = cmp_inputs(2, 128, 160, seed=5)
    with torch.no_grad():
        ref_logits = o.model(image * 2 - 1, torch.cat([sparse, mask], dim=1))
    logits, h, w = hm.model.forward((image * 2 - 1).to(DEV), torch.cat([sparse, mask], dim=1).to(DEV))
    e = rel_l2(_untok(logits, 2, 198, h, w), ref_logits)
    print(f"CMP logits: rel-L2 {e:.3e}")
    assert (h, w) == (64, 80) and e < 2e-2
    ref = o.run(image, sparse, mask)
    got = hm.run(image.to(DEV), sparse.to(DEV), mask.to(DEV)).cpu()
    e = rel_l2(got, ref)
    print(f"CMP flow: rel-L2 {e:.3e}  (|flow| max {ref.abs().max():.2f})")
    assert tuple(got.shape) == tuple(ref.shape) and e < 2e-2


def test_get_flow_vs_oracle(models):
    from mofa_video_amd.cmp import get_flow
    from oracle.cmp import get_flow as oget_flow
    o, hm = models
    fb, fl, hs, ws, H, W = 1, 3, 96, 96, 64, 112
    image, sparse, mask = cmp_inputs(fl, hs, ws, seed=6)
    brush = ((torch.rand(hs, ws, generator=torch.Generator().manual_seed(7)) > 0.3).numpy().astype("uint8") * 255)
    ref = oget_flow(o, image.reshape(fb, fl, 3, hs, ws), sparse.unsqueeze(0), mask.unsqueeze(0), H, W, motion_brush_mask=brush)
    got = get_flow(hm, image.reshape(fb, fl, 3, hs, ws).to(DEV), sparse.unsqueeze(0).to(DEV), mask.unsqueeze(0).to(DEV), H, W,
                   motion_brush_mask=brush).cpu()
    e = rel_l2(got, ref)
    print(f"get_flow: rel-L2 {e:.3e}")
    assert tuple(got.shape) == (fb, fl, 2, H, W) and e < 2e-2


def test_tracks_to_controlnet_flow_end_to_end(models):
    """user tracks -> sparse drags (control.py) -> CMP (HIP) -> in/out-brush merge, against the same chain on the oracle"""
    from mofa_video_amd import control
    from oracle.cmp import get_flow as oget_flow
    o, hm = models
    work, H, W, T = 96, 64, 112, 5
    tracks = [[(10, 12), (40, 30), (70, 80)], [(80, 20), (60, 50)]]
    brush = torch.zeros(work, work, dtype=torch.uint8).numpy()
    brush[:40, :40] = 255
    d = control.tracking_points_to_drags(tracks, work, work, T, brush, work=work)
    assert d["in_flag"] and d["out_flag"]
    first = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(9))
    got = control.controlnet_flow_from_drags(hm, first.to(DEV), d, H, W, motion_brush_mask=brush, work=work).cpu()
    ff = F.interpolate(first, (work, work)).repeat(T - 1, 1, 1, 1).unsqueeze(0)
    fin = oget_flow(o, ff, d["drag_in"].permute(0, 1, 4, 2, 3).float(), d["mask_in"].unsqueeze(2).repeat(1, 1, 2, 1, 1).float(),
                    H, W, motion_brush_mask=brush)
    fout = oget_flow(o, ff, d["drag_out"].permute(0, 1, 4, 2, 3).float(), d["mask_out"].unsqueeze(2).repeat(1, 1, 2, 1, 1).float(),
                     H, W)
    ref = control.merge_inmask_outmask(fin, fout)
    e = rel_l2(got, ref)
    print(f"tracks -> controlnet_flow: rel-L2 {e:.3e}")
    assert tuple(got.shape) == (1, T - 1, 2, H, W) and e < 2e-2
