"""gn_apply / gn_partial / layernorm alone on the denoise step's shapes: microseconds and effective TB/s (read + write bytes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops
l = lib.load()


def timeit(f, iters=10):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return sorted(ts)[1]


for (fr, HW, C, fps) in [(50, 9216, 320, 1), (50, 9216, 320, 25), (50, 9216, 640, 1), (50, 2304, 640, 1), (50, 2304, 640, 25), (50, 2304, 1280, 1), (50, 576, 1280, 1)]:
    M = fr * HW
    x = torch.randn(M, C, device="cuda").half()
    y = torch.empty_like(x)
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    nparts = l.mofa_gn_nparts(HW, C)
    part = torch.empty((fr, nparts, 32, 2), dtype=torch.float32, device="cuda")
    st = lib.stream_ptr()
    tp = timeit(lambda: lib.check(l.mofa_gn_partial_f16(lib.ptr(x), lib.ptr(part), fr, HW, C, C, st), "p"))
    if fps * nparts <= 512:
        ta = timeit(lambda: lib.check(l.mofa_gn_apply_f16(lib.ptr(x), lib.ptr(part), lib.ptr(g), lib.ptr(b), lib.ptr(y), fr, HW, C, C, C, fps, 1e-5, 1, st), "a"))
    else:
        ta = float("nan")
    tl = timeit(lambda: ops.layer_norm(x, g, b, out=y)) if C <= 1280 else float("nan")
    nb = M * C * 2
    print(f"{fr} x {HW} x {C} fps {fps:2d} ({nb / 1e6:5.0f} MB): gn_partial {tp * 1e6:7.1f} us {nb / tp / 1e12:5.2f} TB/s | gn_apply+silu {ta * 1e6:7.1f} us "
          f"{2 * nb / ta / 1e12:5.2f} TB/s | layernorm {tl * 1e6:7.1f} us {2 * nb / tl / 1e12:5.2f} TB/s")
