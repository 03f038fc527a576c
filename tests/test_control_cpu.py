"""CPU tests of the control-signal rasterisers against vectors produced by the reference's own functions
(tests/golden/make_golden_control.py): exact for the integer rasterisation, 1e-12 for the PCHIP resampling."""
import os

import numpy as np
import torch

from mofa_video_amd import control

G = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden_control.pt"),
               weights_only=False)


def test_interpolate_trajectory():
    g = G["interpolate"]
    for tr, ref in zip(g["tracks"], g["out"]):
        got = np.array(control.interpolate_trajectory(tr, g["n_points"]))
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12


def test_divide_points():
    g = G["divide"]
    inm, outm = control.divide_points_afterinterpolate(g["points"], g["brush"])
    assert np.array_equal(inm, g["inmask"]) and np.array_equal(outm, g["outmask"])
    assert len(inm) + len(outm) == len(g["points"]) and len(inm) > 0 and len(outm) > 0


def test_sparseflow_and_mask():
    g = G["sparseflow"]
    flow, mask = control.get_sparseflow_and_mask_forward(g["points"], g["n_steps"], g["H"], g["W"])
    assert np.array_equal(np.argwhere(mask > 0), g["nz"]) and mask.sum() == g["mask_sum"]
    assert np.array_equal(flow[mask > 0], g["flow_at"]) and np.array_equal(flow.sum(axis=(1, 2)), g["flow_sum"])
    fb, _ = control.get_sparseflow_and_mask_forward(g["points"], g["n_steps"], g["H"], g["W"], is_backward_flow=True)
    assert np.array_equal(fb.sum(axis=(1, 2)), g["backward_flow_sum"])


def test_keypoint_sparse_flow():
    g = G["keypoint"]
    flow, mask = control.get_sparse_flow(g["landmarks"].clone(), g["h"], g["w"], g["t"])
    assert torch.equal(flow, g["flow"].to_dense()) and torch.equal(mask, g["mask"].to_dense())


def test_drags_and_merge():
    tracks = G["interpolate"]["tracks"]
    brush = G["divide"]["brush"]
    d = control.tracking_points_to_drags(tracks, 384, 384, 14, brush)
    assert d["in_flag"] and d["out_flag"]
    assert tuple(d["drag_in"].shape) == (1, 13, 384, 384, 2) and tuple(d["mask_out"].shape) == (1, 13, 384, 384)
    total = d["mask_in"].sum() + d["mask_out"].sum()
    assert total == G["sparseflow"]["mask_sum"]
    a = torch.tensor([[[[[1.0, 0.0]], [[2.0, 3.0]]]]]).permute(0, 1, 4, 2, 3)     # [1,2,2,1,1]... components on dim 2
    fin = torch.zeros(1, 2, 2, 1, 2)
    fin[0, 0, :, 0, 0] = torch.tensor([1.0, 2.0])          # both components non-zero -> kept
    fin[0, 0, :, 0, 1] = torch.tensor([1.0, 0.0])          # one zero component -> replaced
    fout = torch.full((1, 2, 2, 1, 2), 7.0)
    m = control.merge_inmask_outmask(fin, fout)
    assert m[0, 0, :, 0, 0].tolist() == [1.0, 2.0] and m[0, 0, :, 0, 1].tolist() == [7.0, 7.0] and (m[0, 1] == 7).all()
    del a


def test_edge_cases_of_the_rasterisers():
    """ragged / degenerate inputs the reference functions accept (Traj/run_gradio.py:41-86, :162-177): two tracks that
    start on the same pixel accumulate, an empty brush group stays an empty array, a two-point track is a straight line,
    the backward flag only flips the sign."""
    import numpy as np
    from mofa_video_amd import control
    line = control.interpolate_trajectory([(10, 20), (30, 60)], 5)
    assert np.allclose(np.array(line), np.stack([np.linspace(10, 30, 5), np.linspace(20, 60, 5)], 1))
    a = np.array(control.interpolate_trajectory([(5, 5), (9, 7), (20, 8)], 4))
    b = np.array(control.interpolate_trajectory([(5, 5), (6, 11), (7, 30)], 4))
    pts = np.stack([a, b])                                            # both start at pixel (5, 5)
    flow, mask = control.get_sparseflow_and_mask_forward(pts, 3, 32, 32)
    assert mask.sum() == 2 * 3 and (mask[:, 5, 5] == 2).all()
    want = np.int64(a[1:] - a[0]) + np.int64(b[1:] - b[0])
    assert np.array_equal(flow[:, 5, 5], want) and np.count_nonzero(flow) == np.count_nonzero(want)
    back, _ = control.get_sparseflow_and_mask_forward(pts, 3, 32, 32, is_backward_flow=True)
    assert np.array_equal(back, -flow)
    brush = np.zeros((32, 32), dtype=np.uint8)                       # nothing brushed: every track is "outside"
    inm, outm = control.divide_points_afterinterpolate(pts, brush)
    assert inm.size == 0 and outm.shape == pts.shape
    brush[:] = 255
    inm, outm = control.divide_points_afterinterpolate(pts, brush)
    assert outm.size == 0 and inm.shape == pts.shape
