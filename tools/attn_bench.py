"""Spatial-attention kernel timing on the denoise step's shapes (25 f x CFG 2 = 50 frames; L0 S = 9216 / 5 heads,
L1 S = 2304 / 10 heads), random data, HIP-event timed, plus an output checksum so that two builds / environment switches
(--qb 1|2 forces the workgroup size) can be compared for bit identity from separate processes.

    python tools/attn_bench.py [--iters 5]
"""
import argparse
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--qb", type=int, default=0, help="0 = the launcher's rule, 1 / 2 = 128- / 256-row workgroups")
    ap.add_argument("--lib", default="", help="A/B: load this build of libmofa_hip.so instead of the in-tree one")
    args = ap.parse_args()
    if args.lib:
        lib.LIB_PATH = os.path.abspath(args.lib)
        print("library:", lib.LIB_PATH)
    lib.load()
    torch.manual_seed(0)
    for (fr, heads, S, tag) in [(50, 5, 9216, "L0"), (50, 10, 2304, "L1"), (50, 5, 9216 - 40, "L0 ragged"), (4, 5, 1000, "small ragged")]:
        Cc = heads * 64
        qkv = (torch.randn(fr * S, 3 * Cc, device="cuda")).half()
        run = lambda: ops.attn_spatial(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], fr, heads, S, query_blocks=args.qb)   # noqa: E731
        out = run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.iters * 1e-3)
        t = sorted(ts)[1]
        # reference on a slice (frame 0, head 0) in fp32
        q, k, v = (qkv[:S, i * Cc:i * Cc + 64].float() for i in range(3))
        ref = torch.softmax(q @ k.t() * 0.125, dim=1) @ v
        err = (out[:S, :64].float() - ref).abs().max().item()
        digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"attn spatial {tag:14s} {fr}x{heads}h S={S:5d}  {t * 1e3:8.3f} ms {4.0 * S * S * Cc * fr / t / 1e12:7.1f} TF/s"
              f"   max|err| vs fp32 {err:.2e}   sha1 {digest}")


if __name__ == "__main__":
    main()
