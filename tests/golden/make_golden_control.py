"""Golden vectors for the control-signal rasterisers, produced BY THE REFERENCE'S OWN FUNCTIONS.  Their
modules import gradio / cv2 / diffusers at import time, so the FunctionDef nodes are taken from the source files in
place (ast) and executed in a namespace that only holds numpy / scipy / torch -- no reference text enters the repo:
  Traj/run_gradio.py: divide_points_afterinterpolate (:41-58), get_sparseflow_and_mask_forward (:61-86),
                      interpolate_trajectory (:162-177)
  Keypoint/utils/utils.py: sample_optical_flow (:81-103), get_sparse_flow (:106-119)
    python tests/golden/make_golden_control.py"""
import ast
import os

import numpy as np
import torch
from scipy.interpolate import PchipInterpolator

HERE = os.path.dirname(os.path.abspath(__file__))


def take(path, names, ns):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            exec(compile(ast.Module([node], []), path, "exec"), ns)
    missing = [n for n in names if n not in ns]
    assert not missing, missing


traj, kp = {"np": np, "PchipInterpolator": PchipInterpolator}, {"torch": torch}
take("/root/reference/MOFA-Video-Traj/run_gradio.py",
     ["divide_points_afterinterpolate", "get_sparseflow_and_mask_forward", "interpolate_trajectory"], traj)
take("/root/reference/MOFA-Video-Keypoint/utils/utils.py", ["sample_optical_flow", "get_sparse_flow"], kp)

G = {}
rng = np.random.RandomState(7)
tracks = [[(30, 40), (80, 60), (150, 90), (170, 200)], [(300, 300), (280, 250)], [(10, 370), (60, 330), (200, 350)]]
T = 14
interp = [traj["interpolate_trajectory"](t, T) for t in tracks]
G["interpolate"] = dict(tracks=tracks, n_points=T, out=[np.array(p) for p in interp])
pts = np.array(interp)                                            # [K, T, 2]
brush = np.zeros((384, 384), dtype=np.uint8)
brush[20:120, 10:200] = 255
inm, outm = traj["divide_points_afterinterpolate"](pts, brush)
G["divide"] = dict(points=pts, brush=brush, inmask=inm, outmask=outm)
flow, mask = traj["get_sparseflow_and_mask_forward"](pts, T - 1, 384, 384)
G["sparseflow"] = dict(points=pts, n_steps=T - 1, H=384, W=384, nz=np.argwhere(mask > 0), flow_at=flow[mask > 0], mask_sum=mask.sum(),
                       flow_sum=flow.sum(axis=(1, 2)))
flow_b, _ = traj["get_sparseflow_and_mask_forward"](pts, T - 1, 384, 384, is_backward_flow=True)
G["sparseflow"]["backward_flow_sum"] = flow_b.sum(axis=(1, 2))
g = torch.Generator().manual_seed(3)
lm = torch.rand(2, 5, 68, 2, generator=g) * torch.tensor([95.0, 63.0])      # (x, y) pixel coordinates, h=64, w=96
sflow, smask = kp["get_sparse_flow"](lm.clone(), 64, 96, 5)
G["keypoint"] = dict(landmarks=lm, h=64, w=96, t=5, flow=sflow.to_sparse(), mask=smask.to_sparse())
torch.save(G, os.path.join(HERE, "reference_golden_control.pt"))
print({k: list(v) for k, v in G.items()}, os.path.getsize(os.path.join(HERE, "reference_golden_control.pt")))
