import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch
import igemm_tiles_bench as tb
from mofa_video_amd import lib
lib.load()
tiles = dict(tb.TILES)
for (M, N, K) in [(460800, 2560, 320), (115200, 5120, 640), (28800, 10240, 1280), (28800, 2560, 1280), (28800, 1280, 1280), (115200, 1280, 2560), (115200, 2560, 2560)]:
    call, fl = tb.make_call("gemm", M, N, K, "none")
    res = {}
    for name in ("256p", "320p"):
        for _ in range(2): call(tiles[name])
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): call(tiles[name])
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4 * 1e-3)
        res[name] = fl / sorted(ts)[1] / 1e12
    print(f"plain GEMM {M:7d} x {N:6d} x {K:5d}:  256p {res['256p']:6.0f}   320p {res['320p']:6.0f} TF/s   ratio {res['320p'] / res['256p']:.3f}")
    del call; torch.cuda.empty_cache()
