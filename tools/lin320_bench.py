"""mofa_lin320_f16 against the launches it replaces at the level-0 shape (M = 50 x 9216 tokens) and a rank-of-8's (7 frames):
norm1 + to_q|k|v (N = 960), to_out + vector + residual, proj_in (N = 320).   python tools/lin320_bench.py [--frames 50 7]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ff320_bench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="*", default=[50, 7])
    args = ap.parse_args()
    from mofa_video_amd import blocks, lib as L, ops
    L.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    sd = {f"a.{k}.weight": (torch.randn(320, 320, generator=g) * 0.05).half() for k in ("to_q", "to_k", "to_v", "to_out.0")}
    sd.update({"a.to_out.0.bias": torch.randn(320, generator=g) * 0.1, "n.weight": 1 + 0.1 * torch.randn(320, generator=g),
               "n.bias": 0.1 * torch.randn(320, generator=g), "p.weight": (torch.randn(320, 320, generator=g) * 0.05).half(),
               "p.bias": torch.randn(320, generator=g) * 0.1})
    s = blocks.Sub(sd, "", dev)
    att, norm, proj = blocks.SelfAttn(s.sub("a"), 5, fold_q_scale=True, norm=s.sub("n")), blocks.LayerNorm(s.sub("n")), blocks.Linear(s.sub("p"))
    for frames in args.frames:
        HW = 9216
        M = frames * HW
        gg = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(M, 320, generator=gg, device=dev).half()
        a = torch.randn(M, 320, generator=gg, device=dev).half()
        vec = torch.randn(2, 320, generator=gg, device=dev)
        rv = (max(frames // 2, 1) * HW, 1, 1, 1 << 30)

        from mofa_video_amd.weights import pack_lin320
        wo, bo = pack_lin320(sd["a.to_out.0.weight"], sd["a.to_out.0.bias"])
        wpi, bpi = pack_lin320(sd["p.weight"], sd["p.bias"])
        wo, bo, wpi, bpi = wo.to(dev), bo.to(dev), wpi.to(dev), bpi.to(dev)
        rows = [("norm1 + to_q|k|v (N = 960)", 960, lambda: att.qkv_normed(x), lambda: att.qkv(norm(x))),
                ("to_q|k|v on normed tokens", 960, lambda: ops.lin320(x, att.qkv_pk[0], att.qkv_pk[1]), lambda: att.qkv(x)),
                ("to_out + vector + residual", 320, lambda: ops.lin320(a, wo, bo, r1=x, s1=1.0, rowvec=vec, rv=rv), lambda: att.to_out(a, r1=x, s1=1.0, rowvec=vec, rv=rv)),
                ("proj_in (bias)", 320, lambda: ops.lin320(x, wpi, bpi), lambda: proj(x))]
        print(f"M = {frames} x {HW} = {M} tokens")
        for name, N, new, old in rows:
            tn, tnm = timeit(new)
            to, tom = timeit(old)
            fl = 2.0 * M * N * 320
            gb = M * (320 + N + (320 if "residual" in name else 0)) * 2.0
            print(f"  {name:30s} lin320 {tn:8.1f} us = {fl / tn / 1e6:7.1f} TF/s, {gb / tn / 1e3:6.0f} GB/s | before {to:8.1f} us = {fl / to / 1e6:7.1f} TF/s | x{to / tn:.3f}")


if __name__ == "__main__":
    main()
