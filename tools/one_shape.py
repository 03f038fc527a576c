"""Run ONE igemm shape repeatedly (PMC profiling target).  usage: one_shape.py conv|gemm M K N [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops
kind = sys.argv[1]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
lib.load()
if kind == "conv":      # M = n*H*W given as n H W packed: args n H W C
    n, H, W, C = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]); iters = 5
    x = (torch.randn(n * H * W, C, device="cuda")).half(); w = (torch.randn(C, 9 * C, device="cuda") * 0.02).half()
    g = ops.conv3x3_geom(H, W)
    for _ in range(iters): ops.igemm(x, w, geom=g)
else:
    M, K, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    for _ in range(iters): ops.igemm(x, w)
torch.cuda.synchronize()
