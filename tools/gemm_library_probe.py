"""Calibration: the plain GEMM shapes of one denoise step through libmofa_hip's implicit-GEMM kernels and through the vendor library
(torch.matmul in fp16 = hipBLASLt / rocBLAS, whichever torch prefers and the other one), same box, same data, HIP-event timed.

The product never calls the vendor library (every launch of the path carries a fused prologue / epilogue the library has no form of);
this tool only answers "how far from a tuned library GEMM are the hand-written tiles on these shapes".
Usage: python tools/gemm_library_probe.py [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402

# (M, N, K, what) -- the step's gemm rows of profiles/r06_shape_breakdown.log
SHAPES = [
    (115200, 5120, 640, "L1 feed-forward in (GEGLU in the product)"),
    (28800, 10240, 1280, "L2 feed-forward in (GEGLU in the product)"),
    (115200, 640, 2560, "L1 feed-forward out"),
    (28800, 1280, 5120, "L2 feed-forward out"),
    (460800, 320, 320, "L0 to_out / proj"),
    (460800, 960, 320, "L0 q|k|v"),
    (115200, 640, 640, "L1 to_out / proj"),
    (115200, 1920, 640, "L1 q|k|v"),
    (28800, 1280, 1280, "L2 to_out / proj"),
    (28800, 3840, 1280, "L2 q|k|v"),
    (7200, 10240, 1280, "L3 feed-forward in"),
    (460800, 2560, 320, "L0 feed-forward in (fused into ff320 in the product)"),
    (460800, 320, 1280, "L0 feed-forward out (fused into ff320 in the product)"),
]


def t_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    lib.load()
    print(f"{'M':>7s} {'N':>6s} {'K':>5s}  {'libmofa us':>10s} {'TF/s':>6s} | {'hipblaslt us':>12s} {'TF/s':>6s} | {'rocblas us':>10s} {'TF/s':>6s} | ours/best lib")
    for M, N, K, what in SHAPES:
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * 0.05).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        fl = 2.0 * M * N * K
        ours = t_us(lambda: ops.igemm(x, w, out=out), args.iters)
        ref = {}
        for name in ("hipblaslt", "cublas"):
            try:
                torch.backends.cuda.preferred_blas_library(name)
                wt = w.t()
                ref[name] = t_us(lambda: torch.matmul(x, wt, out=out), args.iters)
            except Exception as e:  # noqa: BLE001
                ref[name] = float("nan")
                print("  (", name, "unavailable:", str(e)[:80], ")")
        best = min(v for v in ref.values() if v == v)
        print(f"{M:7d} {N:6d} {K:5d}  {ours:10.1f} {fl / ours * 1e-6:6.0f} | {ref['hipblaslt']:12.1f} {fl / ref['hipblaslt'] * 1e-6:6.0f} | "
              f"{ref['cublas']:10.1f} {fl / ref['cublas'] * 1e-6:6.0f} | x{best / ours:5.2f}   {what}")


if __name__ == "__main__":
    main()
