"""Golden vectors for the output stage: runs the reference's OWN flow visualisation
(/root/reference/MOFA-Video-Traj/utils/flow_viz.py, imported in place) on seeded flows and stores flow + image.
    python tests/golden/make_golden_output.py      (needs /root/reference; the fixture is committed)"""
import importlib.util
import os

import torch

REF = "/root/reference/MOFA-Video-Traj/utils/flow_viz.py"
spec = importlib.util.spec_from_file_location("ref_flow_viz", REF)
fv = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fv)

g = torch.Generator().manual_seed(1234)
cases = {}
H, W = 48, 64
smooth = torch.stack(torch.meshgrid(torch.linspace(-3, 3, H), torch.linspace(-2, 5, W), indexing="ij"), -1)
cases["smooth"] = smooth
cases["noise"] = torch.randn(H, W, 2, generator=g) * 4
small = torch.randn(H, W, 2, generator=g) * 0.01
cases["small"] = small
unk = torch.randn(H, W, 2, generator=g)
unk[5:9, 7:30, 0] = 2e7
unk[20, 3, 1] = -3e9
cases["unknown"] = unk
cases["zero"] = torch.zeros(H, W, 2)
out = {}
for k, f in cases.items():
    img = fv.flow_to_image(f.clone())
    out[k] = dict(flow=f, image=torch.from_numpy(img))
torch.save(out, os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_golden_output.pt"))
print({k: tuple(v["image"].shape) for k, v in out.items()})
