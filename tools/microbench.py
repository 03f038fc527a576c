"""Kernel micro-benchmarks on the problem shapes of SURVEY.md Appendix A (config 2: 25 f, 576x1024, N = 50).
Prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM kernels) per shape; HIP-event timed."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofa_video_amd import lib, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def h(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


def main():
    lib.load()
    print(f"{'kernel':40s} {'shape':34s} {'ms':>9s} {'TF/s|GB/s':>10s}")
    # plain GEMMs
    for (M, K, N, tag) in [(460800, 320, 320, "L0 linear"), (460800, 320, 960, "L0 qkv"), (460800, 320, 2560, "L0 geglu proj"),
                           (460800, 1280, 320, "L0 ff out"), (115200, 640, 5120, "L1 geglu proj"), (115200, 2560, 640, "L1 ff out"),
                           (28800, 1280, 10240, "L2 geglu proj"), (28800, 5120, 1280, "L2 ff out"), (7200, 1280, 1280, "L3 linear")]:
        x, w = h(M, K), h(N, K, scale=0.05)
        t = timeit(lambda: ops.igemm(x, w))
        print(f"{'igemm plain ' + tag:40s} {f'{M}x{K}x{N}':34s} {t * 1e3:9.3f} {2 * M * K * N / t / 1e12:10.1f}")
    M, Cc = 460800, 320
    x, w = h(M, Cc), h(8 * Cc, Cc, scale=0.05)
    t = timeit(lambda: ops.igemm(x, w, act=lib.ACT_GEGLU_PAIR))
    print(f"{'igemm geglu-pair':40s} {f'{M}x{Cc}x{8 * Cc}':34s} {t * 1e3:9.3f} {2 * M * Cc * 8 * Cc / t / 1e12:10.1f}")
    # conv3x3
    for (n, H, W, Ci, Co, tag) in [(50, 72, 128, 320, 320, "L0"), (50, 36, 64, 640, 640, "L1"), (50, 18, 32, 1280, 1280, "L2"),
                                   (50, 9, 16, 1280, 1280, "L3"), (50, 72, 128, 960, 320, "L0 up"),
                                   (8, 576, 1024, 128, 128, "VAE 128@576x1024"), (8, 288, 512, 256, 256, "VAE 256@288x512")]:
        x, w = h(n * H * W, Ci), h(Co, 9 * Ci, scale=0.02)
        g = ops.conv3x3_geom(H, W)
        t = timeit(lambda: ops.igemm(x, w, geom=g), iters=5)
        print(f"{'igemm conv3x3 ' + tag:40s} {f'{n}x{H}x{W} {Ci}->{Co}':34s} {t * 1e3:9.3f} {2 * n * H * W * 9 * Ci * Co / t / 1e12:10.1f}")
    for (B, T, HW, Cc, tag) in [(2, 25, 9216, 320, "L0"), (2, 25, 2304, 640, "L1")]:
        x, w = h(B * T * HW, Cc), h(Cc, 3 * Cc, scale=0.03)
        g = ops.convt3_geom(T, HW)
        t = timeit(lambda: ops.igemm(x, w, geom=g), iters=5)
        print(f"{'igemm conv(3,1,1) ' + tag:40s} {f'{B}x{T}x{HW} {Cc}':34s} {t * 1e3:9.3f} {2 * B * T * HW * 3 * Cc * Cc / t / 1e12:10.1f}")
    # attention
    for (fr, heads, S, tag) in [(50, 5, 9216, "L0"), (50, 10, 2304, "L1"), (50, 20, 576, "L2"), (50, 20, 144, "L3")]:
        Cc = heads * 64
        qkv = h(fr * S, 3 * Cc)
        vt = ops.transpose_v(qkv[:, 2 * Cc:], fr, heads, S)
        out = torch.empty(fr * S, Cc, dtype=torch.float16, device=DEV)
        l = lib.load()

        def run():
            lib.check(l.mofa_attn_spatial_f16(lib.ptr(qkv), lib.ptr(qkv[:, Cc:]), lib.ptr(qkv[:, 2 * Cc:]), lib.ptr(out), fr, heads, 64,
                                              S, 3 * Cc, 3 * Cc, 3 * Cc, Cc, 0.125, lib.stream_ptr()), "attn")
        t = timeit(run, iters=5)
        print(f"{'attn spatial ' + tag:40s} {f'{fr}x{heads}h S={S}':34s} {t * 1e3:9.3f} {4 * S * S * 64 * heads * fr / t / 1e12:10.1f}")
        t = timeit(lambda: ops.transpose_v(qkv[:, 2 * Cc:], fr, heads, S), iters=5)
        print(f"{'transpose_v ' + tag:40s} {'':34s} {t * 1e3:9.3f} {2 * fr * S * Cc * 2 / t / 1e9:10.0f}")
    for (B, T, HW, heads, tag) in [(2, 25, 9216, 5, "L0"), (2, 25, 2304, 10, "L1")]:
        Cc = heads * 64
        qkv = h(B * T * HW, 3 * Cc)
        t = timeit(lambda: ops.attn_temporal(qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], B, T, HW, heads), iters=5)
        print(f"{'attn temporal ' + tag:40s} {f'{B}x{T}x{HW} {heads}h':34s} {t * 1e3:9.3f} {4 * B * T * HW * Cc * 2 / t / 1e9:10.0f}")
    # HBM-bound kernels
    M, Cc = 460800, 320
    x = h(M, Cc)
    g_, b_ = torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    t = timeit(lambda: ops.group_norm(x, g_, b_, 50, 9216, 1e-5, silu=True))
    print(f"{'group_norm+silu (3 launches)':40s} {f'{M}x{Cc}':34s} {t * 1e3:9.3f} {3 * M * Cc * 2 / t / 1e9:10.0f}")
    t = timeit(lambda: ops.layer_norm(x, g_, b_))
    print(f"{'layer_norm':40s} {f'{M}x{Cc}':34s} {t * 1e3:9.3f} {2 * M * Cc * 2 / t / 1e9:10.0f}")
    y = h(M, Cc)
    t = timeit(lambda: ops.axpby_(x, y, 1.0, 1.0))
    print(f"{'axpby':40s} {f'{M}x{Cc}':34s} {t * 1e3:9.3f} {3 * M * Cc * 2 / t / 1e9:10.0f}")
    a, b = h(M, 640), h(M, 320)
    t = timeit(lambda: ops.concat_channels(a, b))
    print(f"{'concat 640+320':40s} {f'{M}':34s} {t * 1e3:9.3f} {2 * M * 960 * 2 / t / 1e9:10.0f}")
    # softsplat (one adapter level-0 warp set: 24 flows, 72x128, C=320)
    feat = h(9216, 320)
    flow = torch.randn(24, 2, 72, 128, device=DEV) * 3
    t = timeit(lambda: ops.softsplat_avg_tokens(feat, flow, 72, 128), iters=5)
    alg = 24 * 9216 * 4 * ((320 + 1) + 2 + 2 * (320 + 1) + (320 + 1) + 320)
    print(f"{'softsplat gather (24 flows)':40s} {'72x128 C=320':34s} {t * 1e3:9.3f} {alg / t / 1e9:10.0f}")
    xin = torch.randn(24, 321, 72, 128, device=DEV)
    t = timeit(lambda: ops.softsplat_scatter_f32(xin, flow), iters=5)
    print(f"{'softsplat scatter fp32 (24 flows)':40s} {'72x128 C=321':34s} {t * 1e3:9.3f} {alg / t / 1e9:10.0f}")


if __name__ == "__main__":
    main()
