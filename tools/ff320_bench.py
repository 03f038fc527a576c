"""Fused level-0 feed-forward (mofa_ff320_f16) against the three launches it replaces, at the level-0 shape of config 2
(M = 50 x 9216 = 460 800 tokens) and at the rank-of-8 shape (7 frames): device time per layer (HIP events, 20 runs, random data).

    python tools/ff320_bench.py [--lib path] [--frames 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--frames", type=int, nargs="*", default=[50, 7])
    ap.add_argument("--fused-only", action="store_true", help="time only the fused launches (A/B of kernel builds with --lib)")
    args = ap.parse_args()
    from mofa_video_amd import blocks, lib as L, ops
    if args.lib:
        L.LIB_PATH = os.path.abspath(args.lib)
    L.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    sd = {"ff.net.0.proj.weight": (torch.randn(2560, 320, generator=g) * 320 ** -0.5).half(), "ff.net.0.proj.bias": torch.randn(2560, generator=g) * 0.1,
          "ff.net.2.weight": (torch.randn(320, 1280, generator=g) * 1280 ** -0.5).half(), "ff.net.2.bias": torch.randn(320, generator=g) * 0.1,
          "n.weight": 1 + 0.1 * torch.randn(320, generator=g), "n.bias": 0.1 * torch.randn(320, generator=g)}
    s = blocks.Sub(sd, "", dev)
    ff, norm = blocks.GegluFF(s.sub("ff"), norm=s.sub("n")), blocks.LayerNorm(s.sub("n"))
    for frames in args.frames:
        HW = 9216
        M = frames * HW
        T = frames // 2 if frames % 2 == 0 else frames
        gg = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn(M, 320, generator=gg, device=dev)).half()
        h = torch.randn(M, 320, generator=gg, device=dev).half()
        pos = torch.randn(T, 320, generator=gg, device=dev) * 0.3
        lng, lnb = torch.ones(320, device=dev), torch.zeros(320, device=dev)
        flop = 2.0 * M * (2560 * 320 + 320 * 1280)
        rows = [
            ("spatial ff", lambda: ff.fused(x), lambda: ff(norm(x), r1=x, s1=1.0)),
            ("ff_in (+pos, +norm1 out)", lambda: ff.fused(x, pos=pos, HW=HW, T=T, ln_out=(lng, lnb)),
             lambda: norm(ff(norm(x, rowvec=pos, rv_div=HW, rv_mod=T), r1=x, s1=1.0, rowvec=pos, rv=(HW, 1, 1, T)))),
            ("ff_in (+pos), norm1 stand-alone", lambda: norm(ff.fused(x, pos=pos, HW=HW, T=T)),
             lambda: norm(ff(norm(x, rowvec=pos, rv_div=HW, rv_mod=T), r1=x, s1=1.0, rowvec=pos, rv=(HW, 1, 1, T)))),
            ("temporal ff (AlphaBlender)", lambda: ff.fused(x, s_acc=0.7, s1=0.7, r2=h, s2=0.3),
             lambda: ff(norm(x), s_acc=0.7, r1=x, s1=0.7, r2=h, s2=0.3)),
        ]
        print(f"M = {frames} x {HW} = {M} tokens, {flop / 1e12:.3f} TFLOP per layer")
        for name, fu, un in rows:
            tf, tfm = timeit(fu)
            if args.fused_only:
                print(f"  {name:28s} fused {tf:8.1f} us (min {tfm:8.1f}) = {flop / tf / 1e6:7.1f} TF/s   [{os.path.basename(args.lib) or 'in-tree'}]")
                continue
            tu, tum = timeit(un)
            a_ = fu()
            if isinstance(a_, tuple):                              # (out, LayerNorm(out)): compare the LayerNorm outputs, as the unfused row returns
                a_ = a_[1]
            e = (a_.float() - un().float()).norm().item()
            print(f"  {name:28s} fused {tf:8.1f} us (min {tfm:8.1f}) = {flop / tf / 1e6:7.1f} TF/s | unfused {tu:8.1f} us (min {tum:8.1f}) = "
                  f"{flop / tu / 1e6:7.1f} TF/s | x{tu / tf:.3f}   (diff norm {e:.3e})")


if __name__ == "__main__":
    main()
