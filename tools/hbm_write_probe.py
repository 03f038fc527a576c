"""HBM bandwidth by read : write mix on one MI355X -- what bounds a GEMM whose output is three times its input (level-0 q|k|v).

torch's own fill / copy / cat kernels, timed with events: pure write (fill_), 1 : 1 (copy_), 1 : 3 (one 320-channel read, three
320-channel writes = expand + copy), pure read (sum).  Usage: python tools/hbm_write_probe.py
"""
import torch

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

def main():
    M = 460800
    x = torch.randn(M, 320, device="cuda", dtype=torch.float16)
    y = torch.empty(M, 960, device="cuda", dtype=torch.float16)
    z = torch.empty(M, 320, device="cuda", dtype=torch.float16)
    rows = [
        ("write only  (fill 885 MB)", lambda: y.fill_(1.0), y.numel() * 2),
        ("write only  (fill 295 MB)", lambda: z.fill_(1.0), z.numel() * 2),
        ("1 : 1       (copy 295 MB)", lambda: z.copy_(x), 2 * x.numel() * 2),
        ("1 : 3       (read 295, write 885)", lambda: y.view(M, 3, 320).copy_(x[:, None, :].expand(M, 3, 320)), 4 * x.numel() * 2),
        ("read only   (sum 885 MB)", lambda: y.sum(dtype=torch.float32), y.numel() * 2),
    ]
    for name, fn, nbytes in rows:
        us = t_us(fn)
        print(f"  {name:38s} {us:8.1f} us  {nbytes / us * 1e-6:7.2f} TB/s")

if __name__ == "__main__":
    main()
