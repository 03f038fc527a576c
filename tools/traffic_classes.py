"""L2 <-> fabric traffic of the implicit GEMM PER SHAPE CLASS (round-2 verdict, "find the 2.7 x read re-fetch").

    rocprofv3 --pmc FETCH_SIZE  --kernel-trace --output-format csv -d /tmp/tc_f -- python tools/traffic_classes.py run
    rocprofv3 --pmc WRITE_SIZE  --kernel-trace --output-format csv -d /tmp/tc_w -- python tools/traffic_classes.py run
    python tools/traffic_classes.py report /tmp/tc_f /tmp/tc_w > profiles/archive/r03_traffic_classes.md

``run`` launches every (class, tile) case ITERS times in a fixed order and nothing else from libmofa_hip.so; ``report``
matches the per-dispatch counter rows to the cases by that order and prints, per case, FETCH_SIZE and WRITE_SIZE in bytes
next to the algorithmic operand bytes (X once, W once, residual once; output once).  FETCH_SIZE counts the L2's fabric-side
read requests: Infinity-Cache (256 MB) hits are INCLUDED (MI355X_MICROARCH.md), so it is an upper bound of the HBM reads, and
its unit is request-size dependent: 128-byte requests are tallied as 64 B (hence the x2 of the guide for whole-line
streaming), 64-byte requests as 64 B.  The first case of each tile (a K = 64 "copy" GEMM whose X is read exactly once
and is larger than the Infinity Cache) calibrates that factor for the tile's access pattern."""
import csv
import glob
import os
import sys

ITERS = 3
# (label, mode, M or (n, H, W), N, Cin, epilogue, tile)
CASES = [
    ("calibration: X 590 MB read once (K = 64)", "gemm", 4608000, 64, 64, "none", "256p"),
    ("calibration: X 590 MB read once (K = 64)", "gemm", 4608000, 64, 64, "none", "320p"),
    ("calibration: X 590 MB read once (K = 64)", "gemm", 4608000, 64, 64, "none", "192"),
    ("plain GEMM, shallow K (L0 attn out)", "gemm", 460800, 320, 320, "r1", "192"),
    ("plain GEMM, shallow K (L0 attn out)", "gemm", 460800, 320, 320, "r1", "320p"),
    ("plain GEMM, K = 1280 (L0 ff out)", "gemm", 460800, 320, 1280, "r1", "256p"),
    ("plain GEMM, K = 1280 (L0 ff out)", "gemm", 460800, 320, 1280, "r1", "320p"),
    ("plain GEMM, K = 5120 (L2 ff out)", "gemm", 28800, 1280, 5120, "r1", "256p"),
    ("plain GEMM, K = 5120 (L2 ff out)", "gemm", 28800, 1280, 5120, "r1", "320p"),
    ("GEGLU projection L0", "gemm", 460800, 2560, 320, "geglu", "256p"),
    ("GEGLU projection L2", "gemm", 28800, 10240, 1280, "geglu", "256p"),
    ("conv3x3 L0 (320 ch)", "conv", (50, 72, 128), 320, 320, "rv", "256p"),
    ("conv3x3 L0 (320 ch)", "conv", (50, 72, 128), 320, 320, "rv", "320p"),
    ("conv3x3 L2 (1280 ch, W 29 MB)", "conv", (50, 18, 32), 1280, 1280, "rv", "256p"),
    ("conv3x3 L2 (1280 ch, W 29 MB)", "conv", (50, 18, 32), 1280, 1280, "rv", "320p"),
    ("conv(3,1,1) L0", "convt", (2, 25, 9216), 320, 320, "r1", "320p"),
    ("conv3x3 VAE 128 ch @576x1024", "conv", (8, 576, 1024), 128, 128, "r1", "192"),
]


def algorithmic(mode, Mg, N, Cin, epi):
    M = Mg if mode == "gemm" else Mg[0] * Mg[1] * Mg[2]
    K = Cin * (1 if mode == "gemm" else (9 if mode == "conv" else 3))
    n_out = N // 2 if epi == "geglu" else N
    reads = M * Cin * 2 + N * K * 2 + (M * N * 2 if epi == "r1" else 0)
    return M, K, reads, M * n_out * 2


def run():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import igemm_tiles_bench as tb
    from mofa_video_amd import lib
    lib.load()
    tiles = dict(tb.TILES)
    for (label, mode, Mg, N, Cin, epi, tile) in CASES:
        call, _ = tb.make_call(mode, Mg, N, Cin, epi)
        for _ in range(ITERS):
            call(tiles[tile])
        torch.cuda.synchronize()
        del call
        torch.cuda.empty_cache()


def _dispatches(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name", r.get("counter_name")) == counter]
    key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else "dispatch_id"
    rows.sort(key=lambda r: int(r[key]))
    out = []
    for r in rows:
        name = r.get("Kernel_Name", r.get("kernel_name", ""))
        if "igemm" in name and "fixup" not in name:
            out.append((name, float(r.get("Counter_Value", r.get("counter_value")))))
    return out


def report(df, dw):
    fe, wr = _dispatches(df, "FETCH_SIZE"), _dispatches(dw, "WRITE_SIZE")
    assert len(fe) == len(wr) == ITERS * len(CASES), (len(fe), len(wr), ITERS * len(CASES))
    print("| case | tile | kernel | algorithmic reads MB | FETCH_SIZE MB (as counted) | ratio | algorithmic writes MB | WRITE_SIZE MB | ratio |")
    print("|---|---|---|---:|---:|---:|---:|---:|---:|")
    for ci, (label, mode, Mg, N, Cin, epi, tile) in enumerate(CASES):
        M, K, rd, wt = algorithmic(mode, Mg, N, Cin, epi)
        fs = [v for _, v in fe[ci * ITERS + 1:(ci + 1) * ITERS]]             # (skip each case's first launch)
        ws = [v for _, v in wr[ci * ITERS + 1:(ci + 1) * ITERS]]
        f_mb, w_mb = sum(fs) / len(fs) * 1024 / 1e6, sum(ws) / len(ws) * 1024 / 1e6
        kern = fe[ci * ITERS][0].split("(")[0].split("::")[-1][:34]
        print(f"| {label} {M}x{N}x{K} | {tile} | `{kern}` | {rd / 1e6:.0f} | {f_mb:.0f} | {f_mb / (rd / 1e6):.2f} | {wt / 1e6:.0f} | {w_mb:.0f} | "
              f"{w_mb / (wt / 1e6):.2f} |")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
