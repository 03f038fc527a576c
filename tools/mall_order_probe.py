"""Does the ORDER in which a consumer walks a tensor its producer has just written matter on MI355X (256 MB Infinity Cache in front
of HBM)?  A level-0 activation is 295 MB: read front to back right after being written front to back, an LRU memory-side cache
holds its tail and misses on every line; read back to front it would hit on most of it.  The probe runs producer -> consumer pairs
on [460800, 320] fp16 tensors: the PRODUCER writes its output in K row chunks front-to-back or back-to-front (untimed), the consumer
is one ordinary front-to-back launch, timed alone: after a back-to-front producer the consumer meets the most recently written rows
first, which is what a consumer walking in the reverse of its producer's order would see.

    python tools/mall_order_probe.py [--chunks 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402


def timed(fn, prep, iters=6):
    ts = []
    for _ in range(iters):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=460800)
    args = ap.parse_args()
    lib.load()
    M, C, K = args.rows, 320, args.chunks
    dev = "cuda"
    x = torch.randn(M, C, device=dev).half()
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    wq = (torch.randn(960, C, device=dev) * 0.05).half()
    q = torch.empty(M, 960, dtype=torch.float16, device=dev)
    big = torch.empty(1 << 29, dtype=torch.float16, device=dev)        # 1 GB: flushes the cache between a producer and a "cold" consumer
    step = M // K
    order_f = [(i * step, (i + 1) * step if i + 1 < K else M) for i in range(K)]
    order_b = order_f[::-1]

    def produce(order):
        for r0, r1 in order:
            ops.layer_norm(x[r0:r1], g, b, out=y[r0:r1])              # writes y chunk by chunk in the given order

    def consume_ln():
        ops.layer_norm(y, g, b, out=z)                                # ONE launch, walks y front to back

    def consume_gemm():
        ops.igemm(y, wq, out=q)

    def consume_gn():
        ops.group_norm(y, g, b, M // 9216 if M % 9216 == 0 else 1, 9216 if M % 9216 == 0 else M, 1e-5, out=z)

    print(f"tensor {M} x {C} fp16 = {M * C * 2 / 1e6:.0f} MB; producer = LayerNorm in {K} row chunks, consumer = ONE front-to-back "
          f"launch (timed alone); median of 6, us")
    for name, cons in (("LayerNorm (read + write)", consume_ln), ("GroupNorm (2 reads + write)", consume_gn),
                       ("QKV GEMM N = 960 (read X, write 3x)", consume_gemm)):
        t_f = timed(cons, lambda: produce(order_f))
        t_b = timed(cons, lambda: produce(order_b))
        t_cold = timed(cons, lambda: (produce(order_f), big.zero_()))
        print(f"{name:38s} producer front-to-back (tail hot) {t_f:8.1f} | producer back-to-front (head hot) {t_b:8.1f} | "
              f"after a 1 GB flush {t_cold:8.1f}")


if __name__ == "__main__":
    main()
