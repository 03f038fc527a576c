"""Profiling target for rocprofv3: the full-size hot path for ONE denoise step + ONE 8-frame VAE chunk (config 2
geometry: 25 f, 576x1024).  Used for --kernel-trace --stats and for the separate --pmc passes; never for reported
throughput (bench.py times whole clips)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd import ops  # noqa: E402
from mofa_video_amd.blocks import Ctx  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev)
    pipe.overlap_adapter = False          # single-stream order: per-kernel counters / durations with nothing running beside
    inp = bench.synthetic_inputs(dev)
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=bench.H, width=bench.W,
               num_frames=bench.T, num_inference_steps=steps, decode_chunk_size=8, latents=inp["latents"],
               output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
    lat = out.frames
    z = lat[0, :8]
    frames = pipe.vae.decode(z, num_frames=8, _prescale=1.0 / 0.18215)
    torch.cuda.synchronize()
    print("profiled", steps, "denoise step(s) + one 8-frame VAE chunk; finite:", bool(torch.isfinite(frames).all()))


if __name__ == "__main__":
    main()
