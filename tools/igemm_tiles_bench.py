"""Implicit-GEMM tile comparison on the denoise step's own problem shapes (BASELINE config 2: 25 f, 576x1024, CFG 2).

For every shape of the per-step mix (profiles/../shape_breakdown: kind, M, N, K, epilogue, launches per step) time each
output tile forced through mofa_igemm_args.tile, interleaved rounds in one process (guide rule 24), random data (rule
25).  Prints TF/s per tile, the best tile per shape, the launch-weighted mix average for "always this tile" and for
"best per shape", and the auto choice of the launcher's cost model.

    python tools/igemm_tiles_bench.py [--rounds 5] [--quick]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402

DEV = "cuda"
TILES = [("128", lib.TILE_128X128), ("192", lib.TILE_192X128), ("256p", lib.TILE_256X256), ("320p", lib.TILE_256X320),
         ("auto", lib.TILE_AUTO), ("320s", -1)]

# (mode, M-or-(n,H,W), N, Cin, epilogue, launches per denoise step [UNet + ControlNet], tag)
SHAPES = [
    ("gemm", 460800, 2560, 320, "geglu", 21, "L0 GEGLU proj"),
    ("gemm", 115200, 5120, 640, "geglu", 21, "L1 GEGLU proj"),
    ("gemm", 28800, 10240, 1280, "geglu", 21, "L2 GEGLU proj"),
    ("gemm", 460800, 320, 1280, "r1", 21, "L0 ff out"),
    ("conv", (50, 72, 128), 320, 320, "rv", 11, "L0 conv3x3 320"),
    ("gemm", 460800, 320, 320, "r1", 31, "L0 proj / attn out"),
    ("gemm", 460800, 320, 320, "none", 0, "L0 proj_in (no residual; weight in the row above)"),
    ("gemm", 115200, 640, 2560, "r1", 21, "L1 ff out"),
    ("gemm", 28800, 1280, 5120, "r1", 21, "L2 ff out"),
    ("conv", (50, 36, 64), 640, 640, "rv", 9, "L1 conv3x3 640"),
    ("conv", (50, 18, 32), 1280, 1280, "rv", 9, "L2 conv3x3 1280"),
    ("convt", (2, 25, 9216), 320, 320, "r1", 14, "L0 conv(3,1,1)"),
    ("gemm", 460800, 960, 320, "none", 14, "L0 qkv"),
    ("gemm", 115200, 640, 640, "r1", 30, "L1 proj / attn out"),
    ("conv", (50, 9, 16), 1280, 1280, "r1", 19, "L3 conv3x3 1280"),
    ("gemm", 115200, 1920, 640, "none", 14, "L1 qkv"),
    ("convt", (2, 25, 2304), 640, 640, "r1", 14, "L1 conv(3,1,1)"),
    ("gemm", 28800, 1280, 1280, "r1", 30, "L2 proj / attn out"),
    ("convt", (2, 25, 576), 1280, 1280, "r1", 14, "L2 conv(3,1,1)"),
    ("gemm", 28800, 3840, 1280, "none", 14, "L2 qkv"),
    ("conv", (50, 72, 128), 320, 640, "rv", 2, "L0 conv3x3 640->320"),
    ("convt", (2, 25, 144), 1280, 1280, "r1", 22, "L3 conv(3,1,1)"),
    ("conv", (50, 18, 32), 1280, 2560, "rv", 2, "L2 conv3x3 2560->1280"),
    ("gemm", 7200, 10240, 1280, "geglu", 6, "L3 GEGLU proj"),
    # temporal VAE decoder (per chunk of 8 frames; 25 steps share 4 chunks -> weight ~ 4 / 25 per step)
    ("conv", (8, 576, 1024), 128, 128, "r1", 0.8, "VAE conv3x3 128 @576x1024"),
    ("conv", (8, 288, 512), 256, 256, "r1", 0.8, "VAE conv3x3 256 @288x512"),
    ("conv", (8, 144, 256), 512, 512, "r1", 1.0, "VAE conv3x3 512 @144x256"),
    ("convt", (1, 8, 589824), 128, 128, "r1", 1.0, "VAE conv(3,1,1) 128"),
]


def h(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


def make_call(mode, Mg, N, Cin, epi):
    if mode == "gemm":
        M, K, geom = Mg, Cin, ops.PLAIN
        x = h(M, Cin)
    elif mode == "conv":
        n, H, W = Mg
        M, K, geom = n * H * W, 9 * Cin, ops.conv3x3_geom(H, W)
        x = h(M, Cin)
    else:
        B, T, HW = Mg
        M, K, geom = B * T * HW, 3 * Cin, ops.convt3_geom(T, HW)
        x = h(M, Cin)
    w = h(N, K, scale=K ** -0.5)
    bias = torch.randn(N, device=DEV)
    kw = dict(geom=geom)
    if epi == "geglu":
        kw.update(act=lib.ACT_GEGLU_PAIR)
        out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
    else:
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        if epi == "r1":
            kw.update(r1=h(M, N), s1=1.0)
        elif epi == "rv":
            kw.update(rowvec=torch.randn(64, N, device=DEV), rv=(max(M // 50, 1), 1, 1, 64))
    kw_s = dict(kw)                                          # "320s": the 256x320 tile emitting GroupNorm pair sums (fresh output)
    kw.update(out=out)

    def call(tile):
        if tile == -1:
            y = ops.igemm(x, w, bias, tile=lib.TILE_256X320, stats=True, **kw_s)
            if getattr(y, "gn_stats", None) is None:
                raise lib.MofaHipError("no stats path for this shape")
            return y
        return ops.igemm(x, w, bias, tile=tile, **kw)
    return call, 2.0 * M * N * K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--quick", action="store_true", help="first 10 shapes only")
    ap.add_argument("--tiles", default="192,256p,320p,auto")
    ap.add_argument("--only", default="", help="comma-separated substrings of the shape tags to run")
    ap.add_argument("--lib", default="", help="A/B: load this build of libmofa_hip.so instead of the in-tree one")
    ap.add_argument("--div", type=int, default=1, help="per-rank shapes of an N-GPU run: rows / N (N = 8: one CFG half x 4 frame shards)")
    args = ap.parse_args()
    if args.lib:
        lib.LIB_PATH = os.path.abspath(args.lib)
        print("library:", lib.LIB_PATH)
    lib.load()
    tiles = [t for t in TILES if t[0] in args.tiles.split(",")]
    shapes = SHAPES[:10] if args.quick else SHAPES
    if args.only:
        shapes = [sh for sh in shapes if any(o in sh[6] for o in args.only.split(","))]
    print(f"{'shape':30s} {'M':>8s} {'N':>6s} {'K':>6s} {'epi':>6s} " + " ".join(f"{n:>7s}" for n, _ in tiles) + "   best")
    tot = {n: 0.0 for n, _ in tiles}
    tot_best, tot_fl = 0.0, 0.0
    for (mode, Mg, N, Cin, epi, weight, tag) in shapes:
        if args.div > 1:
            if mode == "gemm":
                Mg = Mg // args.div
            elif mode == "conv":
                Mg = (max(Mg[0] // args.div, 1), Mg[1], Mg[2])
            else:
                Mg = (1, max(Mg[1] * Mg[0] // args.div, 1), Mg[2])      # (clips, frames, HW): a shard of one clip
        call, fl = make_call(mode, Mg, N, Cin, epi)
        times = {n: [] for n, _ in tiles}
        ok = {}
        for n, t in tiles:                                   # warm-up; a forced tile may refuse the launch (GEGLU on 256x320)
            try:
                call(t)
                ok[n] = True
            except lib.MofaHipError:
                ok[n] = False
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for n, t in tiles:
                if not ok[n]:
                    times[n].append(float("inf"))
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    call(t)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / args.iters * 1e-3)
        med = {n: sorted(v)[len(v) // 2] for n, v in times.items()}
        best = min((n for n in med if n != "auto"), key=lambda n: med[n])
        M = Mg if mode == "gemm" else (Mg[0] * Mg[1] * Mg[2])
        K = Cin * (1 if mode == "gemm" else (9 if mode == "conv" else 3))
        print(f"{tag:30s} {M:8d} {N:6d} {K:6d} {epi:>6s} " + " ".join(f"{fl / med[n] / 1e12:7.0f}" for n, _ in tiles) + f"   {best}")
        for n in med:
            tot[n] += weight * (med[n] if ok[n] else med["256p" if "256p" in med else best])
        tot_best += weight * med[best]
        tot_fl += weight * fl
        del call
        torch.cuda.empty_cache()
    print("launch-weighted mix, TF/s: " + "  ".join(f"{n} {tot_fl / tot[n] / 1e12:.0f}" for n in tot) +
          f"  best-per-shape {tot_fl / tot_best / 1e12:.0f}")


if __name__ == "__main__":
    main()
