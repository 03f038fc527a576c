"""a few launches of the fused level-0 feed-forward alone (for rocprofv3 --pmc passes): python tools/ff320_prof.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mofa_video_amd import blocks, lib as L  # noqa: E402

L.load()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0)
sd = {"ff.net.0.proj.weight": (torch.randn(2560, 320, generator=g) * 320 ** -0.5).half(), "ff.net.0.proj.bias": torch.randn(2560, generator=g) * 0.1,
      "ff.net.2.weight": (torch.randn(320, 1280, generator=g) * 1280 ** -0.5).half(), "ff.net.2.bias": torch.randn(320, generator=g) * 0.1,
      "n.weight": 1 + 0.1 * torch.randn(320, generator=g), "n.bias": 0.1 * torch.randn(320, generator=g)}
s = blocks.Sub(sd, "", "cuda")
ff = blocks.GegluFF(s.sub("ff"), norm=s.sub("n"))
x = torch.randn(frames * 9216, 320, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").half()
for _ in range(5):
    y = ff.fused(x)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
