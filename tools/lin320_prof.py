"""a few launches of mofa_lin320_f16 alone (N = 960, no norm) for rocprofv3 --pmc passes"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mofa_video_amd import lib as L, ops  # noqa: E402
from mofa_video_amd.weights import pack_lin320  # noqa: E402

L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
w = (torch.randn(N, 320) * 0.05).half()
wp, _ = pack_lin320(w)
wp = wp.cuda()
x = torch.randn(50 * 9216, 320, device="cuda").half()
for _ in range(5):
    y = ops.lin320(x, wp)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
