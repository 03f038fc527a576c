"""CPU tests: the oracle's CMP sparse-to-dense motion encoder against the fixture produced by the
reference's own CMP classes (tests/golden/make_golden_cmp.py), and the product's key inventory."""
import os

import pytest
import torch

from helpers import rel_l2
from mofa_video_amd import schema

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cmp_inputs(n, h, w, seed):
    """the generator of tests/golden/make_golden_cmp.py"""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(n, 3, h, w, generator=g)
    sparse = torch.zeros(n, 2, h, w)
    mask = torch.zeros(n, 2, h, w)
    for i in range(n):
        for _ in range(3):
            y, x = int(torch.randint(0, h, (1,), generator=g)), int(torch.randint(0, w, (1,), generator=g))
            sparse[i, :, y, x] = (torch.rand(2, generator=g) * 2 - 1) * 40 * (i + 1) / n
            mask[i, :, y, x] = 1
    return image, sparse, mask


@pytest.fixture(scope="module")
def cmp():
    from oracle.cmp import CMPDemo
    m = CMPDemo()
    sd = schema.synthetic_state_dict(schema.cmp_schema(), seed=21, gain=2.0)
    m.model.load_state_dict({k: (t if t.dtype == torch.long else t.float()) for k, t in sd.items()})
    return m


def test_cmp_inventory_and_run(cmp):
    G = torch.load(os.path.join(GD, "reference_golden_cmp.pt"), weights_only=False)
    assert schema.cmp_schema() == G["inventory"]
    assert {k: tuple(v.shape) for k, v in cmp.model.state_dict().items()} == G["inventory"]
    r = G["run"]
    image, sparse, mask = cmp_inputs(r["n"], r["h"], r["w"], r["seed"])
    with torch.no_grad():
        logits = cmp.model(image * 2 - 1, torch.cat([sparse, mask], dim=1))
    assert rel_l2(logits[:, :, ::4, ::4], r["logits_stride4"]) < 1e-4
    flow = cmp.run(image, sparse, mask)
    assert tuple(flow.shape) == tuple(r["flow"].shape) and rel_l2(flow, r["flow"]) < 1e-4


def test_get_flow(cmp):
    from oracle.cmp import get_flow
    p = torch.load(os.path.join(GD, "reference_golden_cmp.pt"), weights_only=False)["get_flow"]
    image, sparse, mask = cmp_inputs(p["fl"], p["hs"], p["ws"], p["seed"])
    flow = get_flow(cmp, image.reshape(p["fb"], p["fl"], 3, p["hs"], p["ws"]), sparse.unsqueeze(0), mask.unsqueeze(0),
                    p["H"], p["W"], motion_brush_mask=p["brush"].numpy())
    assert tuple(flow.shape) == tuple(p["flow"].shape) and rel_l2(flow, p["flow"]) < 1e-4
