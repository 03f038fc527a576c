"""Does running the GEGLU feed-forward (projection -> net.2 + residual) in ROW CHUNKS keep the 4C-wide hidden tensor in the 256 MB
Infinity Cache between the two GEMMs?  Level 0 of the bench: M = 460800, C = 320 (hidden 1280: 1.18 GB unchunked).
    python tools/ff_chunk_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402
from mofa_video_amd.weights import interleave_geglu  # noqa: E402

lib.load()
DEV = "cuda"


def run_case(M, C, chunks):
    x = (torch.randn(M, C, device=DEV)).half()
    w1 = (torch.randn(8 * C, C, device=DEV) * C ** -0.5).half()
    b1 = torch.randn(8 * C, device=DEV)
    w1, b1 = interleave_geglu(w1, b1)
    w2 = (torch.randn(C, 4 * C, device=DEV) * (4 * C) ** -0.5).half()
    b2 = torch.randn(C, device=DEV)
    r1 = torch.randn(M, C, device=DEV).half()
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    res = {}
    for rows in chunks:
        rows_ = M if rows == 0 else rows

        merge = rows < 0                       # negative: fixed chunks of -rows, a remainder < 1/4 chunk joins the last one
        rows_ = -rows if merge else rows_
        bounds = list(range(0, M, rows_)) + [M]
        if merge and len(bounds) > 2 and bounds[-1] - bounds[-2] < rows_ // 4:
            del bounds[-2]
        h = torch.empty(max(b - a for a, b in zip(bounds, bounds[1:])), 4 * C, dtype=torch.float16, device=DEV)

        def ff():
            for m0, m1 in zip(bounds, bounds[1:]):
                hh = ops.igemm(x[m0:m1], w1, b1, act=lib.ACT_GEGLU_PAIR, out=h[:m1 - m0])
                ops.igemm(hh, w2, b2, r1=r1[m0:m1], s1=1.0, out=out[m0:m1])
        for _ in range(2):
            ff()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ff()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        res[rows] = sorted(ts)[1]
        chk = out.float().abs().mean().item()
        print(f"M {M} C {C}  chunk rows {rows_:7d}{' (remainder merged)' if merge else '':19s}: {res[rows] * 1e3:8.1f} us per feed-forward   (hidden chunk {h.shape[0] * 4 * C * 2 / 1e6:6.0f} MB; |out| {chk:.4f})")
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "merge":
        run_case(460800, 320, [65536, -65536, 65536, -65536])
    else:
        run_case(460800, 320, [0, 230400, 131072, 65536, 57600, 32768])
        run_case(115200, 640, [0, 57600, 38400, 28800])
