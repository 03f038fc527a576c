"""Host (Python + ctypes launch) time of the denoise loop against its GPU time: the loop is enqueued without any
synchronisation, so `enqueue` = what the host needs to issue N steps, `gpu` = until the device has finished them.  On one
GPU the host runs far ahead; per-rank GPU time shrinks with the frame shards while the host time does not -- this is the
strong-scaling limit to watch.

    python tools/host_time.py [steps] [HxWxT] [graph]

``HxWxT`` (e.g. 64x64x4) keeps every layer and launch but makes the device work negligible: the enqueue time then IS the host
cost of a step (at full size the HIP queue fills and the host is throttled to the device's pace).  ``graph``: with
pipeline.graph_steps (one hipGraph launch per step from the third step on).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    if len(sys.argv) > 2 and "x" in sys.argv[2]:           # python tools/host_time.py 4 64x64x4 : same layers, negligible device work
        bench.H, bench.W, bench.T = (int(v) for v in sys.argv[2].split("x"))
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev)
    pipe.graph_steps = "graph" in sys.argv[2:]
    inp = bench.synthetic_inputs(dev)

    def run(n):
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=bench.H, width=bench.W,
                    num_frames=bench.T, num_inference_steps=n, decode_chunk_size=8, latents=inp["latents"],
                    output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
    run(1)
    torch.cuda.synchronize()
    res = {}
    base = 3 if pipe.graph_steps else 1                    # (graph mode: step 0 eager, step 1 = the capture, replays from then on)
    for n in (base, base + steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[n] = (t1 - t0, t2 - t0)
    enq = (res[base + steps][0] - res[base][0]) / steps
    gpu = (res[base + steps][1] - res[base][1]) / steps
    mode = "hipGraph replay" if pipe.graph_steps else "eager launches"
    print(f"per denoise step ({bench.T} f {bench.H}x{bench.W}, CFG 2, {mode}): host enqueue {enq * 1e3:.1f} ms, device {gpu * 1e3:.1f} ms "
          f"(host / device = {enq / gpu:.2f}); {base}-step run: enqueue {res[base][0] * 1e3:.0f} ms, total {res[base][1] * 1e3:.0f} ms")


if __name__ == "__main__":
    main()
