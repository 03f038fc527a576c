import sys
sys.path.insert(0, "/root/repo")
import torch
from mofa_video_amd import lib, ops
lib.load()
def timeit(f, iters=10):
    f(); torch.cuda.synchronize()
    ts=[]
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/iters*1e-3)
    return sorted(ts)[1]
for (fr, HW, C) in [(50, 9216, 320), (50, 2304, 640), (50, 9216, 640)]:
    M = fr*HW
    x = torch.randn(M, C, device="cuda").half()
    g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    sc = torch.randn(fr, C, device="cuda"); sh = torch.randn(fr, C, device="cuda")
    y = torch.empty_like(x)
    l = lib.load()
    for silu in (0, 1):
        t = timeit(lambda: lib.check(l.mofa_affine_act_f16(lib.ptr(x), lib.ptr(sc), lib.ptr(sh), lib.ptr(y), fr, HW, C, C, C, silu, lib.stream_ptr()), "aa"))
        print(f"affine_act silu={silu} {M}x{C}: {t*1e6:8.1f} us {2*M*C*2/t/1e9:7.0f} GB/s")
    t = timeit(lambda: ops.layer_norm(x, g, b))
    print(f"layer_norm {M}x{C}: {t*1e6:8.1f} us {2*M*C*2/t/1e9:7.0f} GB/s")
    t = timeit(lambda: ops.group_norm(x, g, b, fr, HW, 1e-5, silu=True))
    print(f"group_norm+silu {M}x{C}: {t*1e6:8.1f} us {3*M*C*2/t/1e9:7.0f} GB/s")
