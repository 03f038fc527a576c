"""One rank of an N-GPU frame-sharded run, alone on ONE GPU: what a rank of the 2-way CFG x N/2 frame-shard layout computes per
denoise step (its CFG half, its frames, per-rank kernel shapes, the two networks enqueued in lockstep on two streams) with every exchange answered locally by a LOOPBACK transport -- peers' data = copies of this rank's own, zero latency.
Device time per step of this proxy is a LOWER bound of a real rank's step time (no wire time, no waiting for peers), so

    t(1 GPU) / (N * t_proxy)   is an UPPER bound of the strong-scaling efficiency at N GPUs,

and the host enqueue time beside it says whether the host keeps up with the per-rank device work.  No scaling curve exists
for this build (no multi-GPU node was ever available to it); this is the 1-GPU evidence that stands in.

    python tools/shard_proxy.py [--world 8] [--rank 0] [--steps 4] [--single-stream]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mofa_video_amd.parallel import FrameParallel, Layout, _Done, _HaloWork  # noqa: E402


class LoopbackComm:
    """every peer holds what this rank holds (tools only: a rank of an N-GPU layout timed without its peers)"""

    def __init__(self, rank):
        self.rank = rank

    def all_reduce_sum(self, t, ranks):
        return t.mul_(len(ranks))

    def all_gather(self, t, ranks):
        return [t] + [t.clone() for _ in ranks[1:]]

    def all_gather_world(self, t):
        return [t]

    def all_gather_into(self, buf, slot_rows, ranks, lane=0):
        i = list(ranks).index(self.rank)
        for j in range(len(ranks)):
            if j != i:
                buf[j * slot_rows:(j + 1) * slot_rows].copy_(buf[i * slot_rows:(i + 1) * slot_rows])
        return _Done()

    def gather_small_into(self, buf, slot_rows, ranks, lane=0):
        self.all_gather_into(buf, slot_rows, ranks)
        return buf

    def halo_begin(self, first, last, prev_rank, next_rank, ranks, lane=0):
        return _HaloWork([], last.clone() if prev_rank is not None else None, first.clone() if next_rank is not None else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--single-stream", action="store_true", help="trunk and encoder one after the other (FrameParallel.two_streams = False)")
    ap.add_argument("--split-convs", action="store_true", help="(3,1,1) convolutions as interior + boundary launches")
    ap.add_argument("--geometry", default="", help="HxWxT, e.g. 64x64x8: same launches, negligible device work -> the enqueue time IS the host cost")
    args = ap.parse_args()
    if args.geometry:
        bench.H, bench.W, bench.T = (int(v) for v in args.geometry.split("x"))
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev)
    inp = bench.synthetic_inputs(dev)
    lay = Layout(args.world, args.rank, bench.T)
    par = FrameParallel(lay, LoopbackComm(args.rank))
    par.two_streams = not args.single_stream
    par.split_convs = args.split_convs
    pipe.parallel = par if args.world > 1 else None

    def run(n):
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=bench.H, width=bench.W,
                    num_frames=bench.T, num_inference_steps=n, decode_chunk_size=8, latents=inp["latents"],
                    output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
    run(1)
    torch.cuda.synchronize()
    res = {}
    for n in (1, 1 + args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[n] = (t1 - t0, time.perf_counter() - t0)
    enq = (res[1 + args.steps][0] - res[1][0]) / args.steps
    gpu = (res[1 + args.steps][1] - res[1][1]) / args.steps
    print(f"rank {args.rank} of {args.world} (CFG half {lay.half}, frames {lay.f0}..{lay.f1 - 1} of {bench.T}, {bench.H}x{bench.W}; "
          f"{'two streams, lockstep' if par.two_streams else 'one stream'}, convs {'interior + boundary' if par.split_convs else 'whole'}; "
          f"loopback transport): per denoise step host enqueue {enq * 1e3:.1f} ms, device {gpu * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
