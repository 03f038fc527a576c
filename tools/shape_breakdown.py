"""Per-shape time breakdown of the MFMA implicit-GEMM launches in ONE full-size denoise step (+ one VAE chunk),
HIP-event timed per launch.  Directs kernel tuning: prints shapes sorted by total time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev)
    pipe.overlap_adapter = False                     # single-stream order: a launch's event duration is then the kernel's own
    inp = bench.synthetic_inputs(dev)

    def run(vae=True):
        out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=bench.H, width=bench.W,
                   num_frames=bench.T, num_inference_steps=1, decode_chunk_size=8, latents=inp["latents"],
                   output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
        if vae:
            pipe.vae.decode(out.frames[0, :8], num_frames=8, _prescale=1.0 / 0.18215)
    run()
    run()
    t = ops.LaunchTimer()
    t.tags = []
    ops.TIMER = t
    run()
    ops.TIMER = None
    bt = t.by_tag()
    tot = sum(v[1] for v in bt.values())
    names = {0: "gemm", 1: "conv3x3", 2: "convT3", 9: "ff320", 10: "lin320"}
    print(f"total igemm time {tot * 1e3:.1f} ms over {sum(v[0] for v in bt.values())} launches "
          f"({sum(v[2] for v in bt.values()) / tot / 1e12:.0f} TF/s)")
    print(f"{'kind':8s} {'s':>1s} {'u':>1s} {'M':>9s} {'N':>6s} {'K':>6s} {'act':>3s} {'n':>4s} {'ms':>8s} {'%':>5s} {'TF/s':>6s}")
    for tag, (n, sec, fl) in sorted(bt.items(), key=lambda kv: -kv[1][1])[:60]:
        mode, stride, up, M, N, K, act = tag
        print(f"{names.get(mode, str(mode)):8s} {stride:1d} {up:1d} {M:9d} {N:6d} {K:6d} {act:3d} {n:4d} {sec * 1e3:8.2f} "
              f"{sec / tot * 100:5.1f} {fl / sec / 1e12:6.0f}")


if __name__ == "__main__":
    main()
