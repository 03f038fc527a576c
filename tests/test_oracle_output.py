"""CPU tests of the output-stage oracle: flow_to_image against the fixture produced by the reference's own
flow_viz.py (tests/golden/make_golden_output.py); postprocess against its published definition (diffusers 0.24.0
VaeImageProcessor.postprocess is not in the reference tree: parity unpinned beyond that)."""
import os

import numpy as np
import torch

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_flow_to_image_matches_reference_fixture():
    from oracle.output import flow_to_image
    G = torch.load(os.path.join(GD, "reference_golden_output.pt"), weights_only=False)
    assert set(G) == {"smooth", "noise", "small", "unknown", "zero"}
    for name, case in G.items():
        img = flow_to_image(case["flow"])
        assert img.dtype == np.uint8 and img.shape == tuple(case["image"].shape)
        assert np.array_equal(img, case["image"].numpy()), name


def test_postprocess_definition():
    from oracle.output import tensor2vid
    g = torch.Generator().manual_seed(3)
    video = torch.randn(1, 3, 4, 8, 10, generator=g) * 1.5
    pt = tensor2vid(video, "pt")[0]
    assert tuple(pt.shape) == (4, 3, 8, 10) and pt.min() >= 0 and pt.max() <= 1
    npv = tensor2vid(video, "np")[0]
    assert npv.shape == (4, 8, 10, 3) and npv.dtype == np.float32
    assert np.array_equal(npv, pt.permute(0, 2, 3, 1).numpy())
    pil = tensor2vid(video, "pil")[0]
    assert len(pil) == 4 and pil[0].size == (10, 8)
    assert np.array_equal(np.asarray(pil[2]), (npv[2] * 255).round().astype("uint8"))
    # round-half-even at an exact .5: 0.5/255*... pick x so that x/2+0.5 == 2.5/255 is not representable exactly; check both neighbours
    x = torch.tensor([[[[(2 * 0.5 / 255.0 - 1.0)]]]]).reshape(1, 1, 1, 1, 1).repeat(1, 3, 1, 1, 1)
    assert np.asarray(tensor2vid(x, "pil")[0][0]).item(0) in (0, 1)
