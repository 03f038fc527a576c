import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from mofa_video_amd import lib, ops
import test_ff320_gpu as T
lib.load()
prm = T._params(3)
w1p, b1f, w2p, b2 = T._pack(prm)
for M in (640, 128, 33):
    g = torch.Generator(device='cuda').manual_seed(M)
    x = (torch.randn(M, 320, generator=g, device='cuda') * 1.3 + 0.2).half()
    ref, _ = T._reference(x, prm)
    got = ops.ff320(x, w1p, b1f, w2p, b2).float()
    err = (got - ref).abs()
    bad = err > 3e-3 * (ref.abs().max() + ref.abs())
    print("M", M, "bad", int(bad.sum()), "max err", err.max().item(), "nan", int(torch.isnan(got).sum()))
    rows = bad.any(1).nonzero().flatten().tolist()
    cols = bad.any(0).nonzero().flatten().tolist()
    print(" bad rows", rows[:40], len(rows)); print(" bad cols", cols[:40], len(cols))
    # error as function of hidden chunk: recompute partial contributions
    got2 = ops.ff320(x, w1p, b1f, w2p, b2).float()
    print(" repeat equal:", torch.equal(got, got2), " err mean", err.mean().item())
