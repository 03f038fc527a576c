"""Which implicit-GEMM launches of one clip run far below the mix average, and what they cost: one bench clip (config 2 by
default) with the per-launch HIP-event timer, grouped by (mode, stride, up, M, N, K, act), everything below --below TF/s
sorted by its share of the clip's igemm time.

    python tools/slow_launches.py [--config 2] [--below 500]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd import lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--below", type=float, default=500.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib.load()
    pipe = bench.build_config_pipeline(dev, args.config, seed=0)
    inp = bench.config_inputs(dev, args.config, seed=42)
    bench.run_config(pipe, inp, args.config)
    torch.cuda.synchronize()
    timer = ops.LaunchTimer()
    ops.TIMER = timer
    bench.run_config(pipe, inp, args.config)
    ops.TIMER = None
    tags = timer.by_tag()
    tot = sum(v[1] for v in tags.values())
    print(f"igemm: {sum(v[0] for v in tags.values())} launches, {tot * 1e3:.1f} ms, "
          f"{sum(v[2] for v in tags.values()) / tot / 1e12:.0f} TF/s")
    print(f"{'mode':>4s} {'s':>1s} {'u':>1s} {'M':>9s} {'N':>6s} {'K':>6s} {'act':>3s} {'n':>5s} {'ms':>8s} {'%':>5s} {'TF/s':>6s} {'GB/s':>6s}")
    slow = 0.0
    for tag, (n, sec, fl) in sorted(tags.items(), key=lambda kv: -kv[1][1]):
        tf = fl / sec / 1e12
        if tf >= args.below:
            continue
        mode, stride, up, M, N, K, act = tag
        gbs = n * M * (K / (9 if mode == 1 else (3 if mode == 2 else 1)) + N) * 2.0 / sec / 1e9   # one read of X + one write
        slow += sec
        print(f"{mode:4d} {stride:1d} {up:1d} {M:9d} {N:6d} {K:6d} {act:3d} {n:5d} {sec * 1e3:8.2f} {100 * sec / tot:5.2f} {tf:6.0f} {gbs:6.0f}")
    print(f"below {args.below:.0f} TF/s: {slow * 1e3:.1f} ms = {100 * slow / tot:.1f} % of the igemm time")


if __name__ == "__main__":
    main()
