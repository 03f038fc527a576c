"""Where a clip's time goes outside the denoise loop: image conditioning (antialiased resize + CLIP + VAE encode), the adapter's
per-clip preparation (condition CNN, pyramids, 96 forward-splat warps), one denoise step, the VAE decode.
    python tools/phase_times.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd import lib  # noqa: E402
from mofa_video_amd.vae import decode_latents  # noqa: E402


def timed(fn, n=3):
    ts = []
    out = None
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], out


def main():
    dev = torch.device("cuda", 0)
    lib.load()
    pipe = bench.build_pipeline(dev, frontend=True)
    inp = bench.synthetic_inputs(dev)
    H, W, T = bench.H, bench.W, bench.T
    t_cond, _ = timed(lambda: pipe._conditioning(inp["image"], None, None, H, W, 0.02, torch.Generator().manual_seed(1)))
    cond = pipe._condition_image(inp["cond"], H, W)
    t_prep, _ = timed(lambda: pipe.controlnet.prepare_condition(cond[:1], inp["flow"][:1]))
    lat = inp["latents"].to(dev)
    t_dec, _ = timed(lambda: decode_latents(pipe.vae, lat, T, bench.CHUNK))

    def run(n):
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W, num_frames=T,
                    num_inference_steps=n, decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type="latent",
                    image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
    t1, _ = timed(lambda: run(1))
    t3, _ = timed(lambda: run(3))
    print(f"image conditioning (resize + CLIP + VAE encode) {t_cond:8.1f} ms")
    print(f"adapter prepare_condition                       {t_prep:8.1f} ms")
    print(f"denoise step (two-stream default)               {(t3 - t1) / 2:8.1f} ms   x 25 = {(t3 - t1) / 2 * 25:8.1f}")
    print(f"VAE decode (chunks of {bench.CHUNK})                         {t_dec:8.1f} ms")


if __name__ == "__main__":
    main()
