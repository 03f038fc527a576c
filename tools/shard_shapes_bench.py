import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch
import igemm_tiles_bench as tb
from mofa_video_amd import lib
lib.load()
DIV = int(sys.argv[1]) if len(sys.argv) > 1 else 8     # 8 GPUs: one CFG half x 4 frame shards -> rows / 8
tot_t = tot_f = 0.0
print(f"per-rank shapes at {DIV} GPUs (rows / {DIV}); auto tile")
for (mode, Mg, N, Cin, epi, weight, tag) in tb.SHAPES[:23]:
    if mode == "gemm":
        Mg2 = Mg // DIV
    elif mode == "conv":
        Mg2 = (max(Mg[0] // DIV, 1), Mg[1], Mg[2])
    else:
        Mg2 = (1, max(Mg[1] * Mg[0] // DIV, 1), Mg[2])      # (clips, frames, HW): a shard of one clip
    call, fl = tb.make_call(mode, Mg2, N, Cin, epi)
    for _ in range(2): call(0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call(0)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e-3)
    t = sorted(ts)[1]
    tot_t += weight * t; tot_f += weight * fl
    print(f"{tag:28s} {str(Mg2):>18s} {N:6d} {fl / t / 1e12:7.0f} TF/s  {t*1e6:8.1f} us")
print(f"launch-weighted mix: {tot_f / tot_t / 1e12:.0f} TF/s")
