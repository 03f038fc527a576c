"""K-loop probe of the phase-pipelined 256x256 implicit-GEMM tile (csrc/igemm8.hip): times the diagnostic build variants
(template parameter VAR, selected through the hook mofa_igemm8_set_probe) on plain GEMMs with random data, interleaved rounds
in one process, and prints the per-segment cycle trace of variant 64.  The variants and the hook are NOT in the product
library: this tool builds and loads tools/libmofa_hip_probe.so (the same sources with -DMOFA_PROBE; `_build.build(probe=True)`).

    python tools/igemm8_probe.py [--shapes 4096x4096x4096,...] [--vars 0,1,2,...]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import _build, lib, ops  # noqa: E402

lib.LIB_PATH = os.path.abspath(_build.build(probe=True))   # the probe library stands in for libmofa_hip.so in this process

NAMES = {0: "shipped", 1: "no stagger", 2: "no setprio", 3: "no stagger, no setprio", 4: "DMA issue before reads", 8: "no vmcnt wait (wrong)",
         16: "no DMA in loop (wrong)", 32: "no fragment reads (wrong)", 48: "MFMA + barriers only (wrong)", 64: "traced",
         68: "traced, DMA first", 128: "barrier 2 MFMAs early", 256: "barrier 3 MFMAs early", 512: "barrier 4 MFMAs early",
         192: "traced, barrier 2 early", 176: "2 early, MFMA + barriers only (wrong)", 560: "4 early, MFMA + barriers only (wrong)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096x4096,8192x8192x8192,460800x2560x320")
    ap.add_argument("--vars", default="0,64,16,32,48")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    l = lib.load()
    hook = l.mofa_igemm8_set_probe
    hook.argtypes = [C.c_int, C.c_void_p]
    variants = [int(v) for v in args.vars.split(",")]
    for shp in args.shapes.split(","):
        M, N, K = (int(v) for v in shp.split("x"))
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        fl = 2.0 * M * N * K
        ref = None
        times = {v: [] for v in variants}
        for r in range(args.rounds + 1):
            for v in variants:
                hook(v, None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    ops.igemm(x, w, tile=lib.TILE_256X256, out=out)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[v].append(e0.elapsed_time(e1) / args.iters * 1e-3)
        hook(0, None)
        ops.igemm(x, w, tile=lib.TILE_192X128, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.igemm(x, w, tile=lib.TILE_192X128, out=out)
        e1.record()
        torch.cuda.synchronize()
        print(f"== {M} x {N} x {K}   (192x128 tile: {fl / (e0.elapsed_time(e1) / args.iters * 1e-3) / 1e12:.0f} TF/s)")
        for v in variants:
            t = sorted(times[v])
            print(f"   var {v:3d} {NAMES.get(v, ''):34s} median {fl / t[len(t) // 2] / 1e12:7.0f}  best {fl / t[0] / 1e12:7.0f} TF/s")
        # cycle trace
        for v in (64,):
            grid = 256
            tr = torch.zeros(grid * 2 * 16, dtype=torch.int64, device="cuda")
            hook(v, tr.data_ptr())
            ops.igemm(x, w, tile=lib.TILE_256X256, out=out)
            torch.cuda.synchronize()
            hook(0, None)
            t = tr.view(grid, 2, 16).double()
            used = t[:, 0, 12] > 0
            kt = t[used][:, :, 12:13]
            per = (t[used][:, :, :6] / kt).mean(0)             # [group][6] cycles per K tile
            print(f"   trace var {v} ({NAMES[v]}): cycles per K tile, mean over {int(used.sum())} workgroups"
                  f" ({kt.mean().item():.0f} K tiles each); ideal MFMA issue = 512 per phase")
            for g in range(2):
                row = per[g].tolist()
                print(f"      group {g}: " + "  ".join(f"P{p + 1}[load+wait {row[3 * p]:5.0f} mfma {row[3 * p + 1]:4.0f} bar2 {row[3 * p + 2]:4.0f}]"
                                                      for p in range(2)) + f"   sum {sum(row):6.0f}")


def tile_trace():
    """per-tile anatomy of the traced kernel on the bench's own shapes and epilogue kinds (plain, r1, row vector, GEGLU)"""
    import igemm_tiles_bench as tb
    l = lib.load()
    hook = l.mofa_igemm8_set_probe
    hook.argtypes = [C.c_int, C.c_void_p]
    print(f"{'shape':28s} {'epi':>6s} {'TF/s':>6s} {'traced':>6s} {'GHz':>5s} | cycles per tile: {'K loop':>7s} ({'per K tile':>10s}) {'resync':>6s} "
          f"{'epilogue':>8s} {'setup':>6s} | {'K tiles':>7s} {'tiles/WG':>8s}")
    for (mode, Mg, N, Cin, epi, weight, tag) in tb.SHAPES:
        if epi == "none":
            epi = "plain"
        call, fl = tb.make_call(mode, Mg, N, Cin, "none" if epi == "plain" else epi)
        def timed(var, tr=None):
            hook(var, tr)
            call(lib.TILE_256X256)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                call(lib.TILE_256X256)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 3 * 1e-3
        t0 = timed(0)
        tr = torch.zeros(256 * 2 * 16, dtype=torch.int64, device="cuda")
        t64 = timed(64, tr.data_ptr())
        tr.zero_()
        hook(64, tr.data_ptr())
        call(lib.TILE_256X256)
        torch.cuda.synchronize()
        hook(0, None)
        t = tr.view(256, 2, 16).double()
        used = t[:, 0, 8] > 0
        g = t[used].mean(1)                                      # mean of the two groups, [wg][16]
        tiles = g[:, 8].mean().item()
        kt = g[:, 12].mean().item() / tiles
        kloop = (g[:, :6].sum(1) / g[:, 8]).mean().item()
        resync = (g[:, 6] / g[:, 8]).mean().item()
        epil = (g[:, 7] / g[:, 8]).mean().item()
        setup = (g[:, 9] / g[:, 8]).mean().item()
        total = (g[:, :10].sum(1)).mean().item()                # cycles of the whole kernel per workgroup
        ghz = total / t64 * 1e-9
        print(f"{tag:28s} {epi:>6s} {fl / t0 / 1e12:6.0f} {fl / t64 / 1e12:6.0f} {ghz:5.2f} | {'':17s}{kloop:7.0f} ({kloop / kt:10.0f}) {resync:6.0f} "
              f"{epil:8.0f} {setup:6.0f} | {kt:7.0f} {tiles:8.1f}")
        del call
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tiles":
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        tile_trace()
    else:
        main()
