import sys, os
sys.path.insert(0, "/root/repo")
import torch
from mofa_video_amd import lib, ops
lib.load()
DEV = "cuda"
def t(fn, it=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / it)
    return sorted(ts)[1]
for (M, N, K) in [(460800, 960, 320), (460800, 320, 320), (115200, 640, 640), (460800, 320, 1280)]:
    x = torch.randn(M, K, device=DEV).half(); w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.randn(N, device=DEV); out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    r1 = torch.randn(M, N, device=DEV).half()
    for tile, name in ((lib.TILE_256X320, "320p"), (lib.TILE_256X256, "256p")):
        t(lambda: ops.igemm(x, w, None, out=out, tile=tile))          # (the first measurement after fresh allocations runs slower:
        c = t(lambda: ops.igemm(x, w, None, out=out, tile=tile))      #  discarded; an earlier version of this probe mistook that for a bias cost)
        a = t(lambda: ops.igemm(x, w, b, out=out, tile=tile))
        d = t(lambda: ops.igemm(x, w, b, r1=r1, s1=1.0, out=out, tile=tile))
        e = t(lambda: ops.igemm(x, w, None, r1=r1, s1=1.0, out=out, tile=tile))
        print(f"{M}x{N}x{K} {name}: plain bias {a*1e3:7.1f} us  no bias {c*1e3:7.1f} us | r1 bias {d*1e3:7.1f}  r1 no bias {e*1e3:7.1f}")
