import sys, os
sys.path.insert(0, "/root/repo")
import torch
from mofa_video_amd import lib, ops
lib.load()
for (B, T, HW, heads, hd, tag) in [(2, 25, 9216, 5, 64, "L0"), (2, 25, 2304, 10, 64, "L1"), (2, 25, 576, 20, 64, "L2"), (2, 25, 576, 10, 128, "CN L2 d128")]:
    Cc = heads * hd
    qkv = torch.randn(B * T * HW, 3 * Cc, device="cuda").half()
    run = lambda: ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], B, T, HW, heads, head_dim=hd)
    out = run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e-3)
    t = sorted(ts)[1]
    print(f"attn temporal {tag:12s} {B}x{T}x{HW} {heads}h d{hd}: {t*1e6:8.1f} us  {4 * B * T * HW * Cc * 2 / t / 1e9:7.0f} GB/s (q, k, v read + out written)")
