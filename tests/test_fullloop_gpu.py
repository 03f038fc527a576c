"""The WHOLE denoise loop of the headline configuration against the oracle (round-3 verdict, "What's missing" 1): BASELINE
config 2 -- 25 frames, 576x1024, full SVD-XT widths, ``STEPS`` = 25 Euler steps (MOFA-Video-Traj/pipeline/pipeline.py:447-511,
utils/scheduling_euler_discrete_karras_fix.py:418-528) -- then ``decode_latents`` in chunks (8, 8, 8, 1) (pipeline.py:194-220).

The checker is ``oracle.pipeline.denoise`` + ``oracle.vae.decode_latents`` moved to the GPU in fp32 (the harness of
tests/test_fullgeom_gpu.py: MIOpen off, exact chunked attention), on the same seeded weights and the bench's inputs.  One
oracle run (module fixture, latents kept after EVERY step) is compared with the product in every order it ships:

  (a) the default order: adapter trunk || UNet encoder and the decoder's CFG halves on two HIP streams;
  (b) single-stream order (``overlap_adapter = False``; what ``bench.py --single-stream`` and the roofline leg run);
  (c) ``round_latents_to_fp16=True`` (the reference's fp16 run rounds the latents after every step, scheduling_...:520)
      -- against the fp32 oracle (and, with MOFA_FULLLOOP_ORACLE_FP16=1, against the oracle with the same rounding: a second
      oracle run, profiles/r04_fullloop.log);
  (d) the frame-sharded layout, world 4 (2-way CFG x 2 frame shards) as virtual ranks on this GPU, every rank's shard after
      every step and the gathered clip at the end: sharded-vs-single, kernel-vs-oracle and 25 steps of error growth are spent
      against ONE tolerance here.

Stated fp16 tolerance (fp16 storage / fp32 accumulate against fp32): rel-L2 <= 2e-2 for the latents after any number of
steps and for the decoded frames.  Every case prints rel-L2 after steps 1, 5, 10, 25 (``profiles/r04_fullloop.log``).
``MOFA_FULLLOOP_STEPS`` shortens the loop (development); the suite runs all 25.
"""
import os
import threading

import pytest
import torch

import bench
from test_fullgeom_gpu import DEV, exact_fp32_gpu, gpu_oracle

pytestmark = pytest.mark.gpu
T, H, W = bench.T, bench.H, bench.W
STEPS = int(os.environ.get("MOFA_FULLLOOP_STEPS", "25"))
MARKS = sorted({s for s in (1, 5, 10, 15, 20, 25) if s <= STEPS} | {STEPS})
TOL = 2e-2


def rel(a, b):
    a, b = a.to(DEV, torch.float32), b.to(DEV, torch.float32)
    assert tuple(a.shape) == tuple(b.shape), (a.shape, b.shape)
    assert bool(torch.isfinite(a).all()), "non-finite product output"
    return float((a - b).norm() / (b.norm() + 1e-12))


class _RoundingScheduler:
    """the oracle scheduler with the reference's fp16 rounding of ``prev_sample`` (scheduling_...:520 casts to the model's dtype)"""

    def __init__(self, inner):
        self._s = inner

    def __getattr__(self, k):
        return getattr(self._s, k)

    def step(self, model_output, timestep, sample):
        return self._s.step(model_output, timestep, sample).half().float()


@pytest.fixture(scope="module")
def world():
    """inputs, shared fp16-valued state dicts, the product modules, the oracle's trace of every step and its decoded frames"""
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import FlowControlNet
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    from oracle.controlnet import FlowControlNet as OCn
    from oracle.pipeline import denoise
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    from oracle.vae import AutoencoderKLTemporalDecoder as OVae
    from oracle.vae import decode_latents as odecode
    inp = bench.synthetic_inputs(torch.device(DEV))
    mk = lambda sch, seed: schema.synthetic_state_dict(sch, seed=seed, device=DEV)   # noqa: E731
    sds = dict(unet=mk(schema.unet_schema(), 0), cn=mk(schema.controlnet_schema(), 1), vae=mk(schema.vae_decoder_schema(), 2))
    il2 = torch.cat([torch.zeros_like(inp["image_latents"]), inp["image_latents"]])
    emb2 = torch.cat([torch.zeros_like(inp["image_embeddings"]), inp["image_embeddings"]])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with exact_fp32_gpu():
        ou, oc = gpu_oracle(OUnet, sds["unet"]), gpu_oracle(OCn, sds["cn"])
        ev0.record()
        _, trace = denoise(ou, oc, OSch(), inp["latents"], il2, emb2, inp["cond"], inp["flow"], num_inference_steps=STEPS,
                           return_trace=True)
        ev1.record()
        trace16 = None
        if os.environ.get("MOFA_FULLLOOP_ORACLE_FP16", "0") == "1":   # (a second 90 s oracle run: opt-in since r05, recorded in profiles/r04_fullloop.log)
            _, trace16 = denoise(ou, oc, _RoundingScheduler(OSch()), inp["latents"], il2, emb2, inp["cond"], inp["flow"],
                                 num_inference_steps=STEPS, return_trace=True)
        del ou, oc
        ov = gpu_oracle(OVae, sds["vae"])
        frames = odecode(ov, trace[-1], T, bench.CHUNK)                      # fp32 [1,3,T,H,W]
        del ov
    torch.cuda.synchronize()
    print(f"oracle on the GPU in fp32: {STEPS} steps at {T} f {H}x{W} in {ev0.elapsed_time(ev1) * 1e-3:.1f} s")
    hip = dict(unet=UNetSpatioTemporalConditionControlNetModel(sds["unet"], None, DEV), cn=FlowControlNet(sds["cn"], None, DEV),
               vae=AutoencoderKLTemporalDecoder(sds["vae"], None, DEV))
    del sds
    torch.cuda.empty_cache()
    return dict(inp=inp, il2=il2, emb2=emb2, hip=hip, trace=trace, trace16=trace16, frames=frames)


def _run(world, output_type="raw", parallel=None, keep=None, single_stream=False, **ctor):
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    hip, inp = world["hip"], world["inp"]
    pipe = FlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], controlnet=hip["cn"], scheduler=EulerDiscreteScheduler(),
                                  parallel=parallel, **ctor)
    if single_stream:
        pipe.overlap_adapter = False

    def cb(p, i, t, kw):
        if keep is not None and (i + 1) in MARKS:
            keep[i + 1] = kw["latents"].clone()
        return {}
    return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W, num_frames=T,
                num_inference_steps=STEPS, decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type=output_type,
                callback_on_step_end=cb, image_embeddings=world["emb2"], image_latents=world["il2"]).frames


def _report(name, keep, trace, sl=slice(None)):
    errs = {s: rel(keep[s], trace[s - 1][:, sl]) for s in MARKS}
    print(f"{name}: latents rel-L2 vs oracle after step " + ", ".join(f"{s}: {e:.3e}" for s, e in errs.items()))
    return errs


@pytest.mark.parametrize("mode", ["default two-stream + split decoder", "single-stream", "round_latents_to_fp16"])
def test_full_loop_vs_oracle(world, mode):
    keep = {}
    kw = dict(single_stream=(mode == "single-stream"))
    if mode == "round_latents_to_fp16":
        kw["round_latents_to_fp16"] = True
    frames = _run(world, keep=keep, **kw)
    errs = _report(f"config 2, {STEPS} steps @ {T}f {H}x{W}, {mode}", keep, world["trace"])
    ef = rel(frames, world["frames"])
    print(f"config 2, {STEPS} steps, {mode}: decoded frames {tuple(frames.shape)} rel-L2 vs oracle {ef:.3e}")
    if mode == "round_latents_to_fp16" and world["trace16"] is not None:
        _report(f"config 2, {STEPS} steps, {mode} vs the oracle WITH the same rounding", keep, world["trace16"])
        e16 = rel(world["trace16"][-1], world["trace"][-1])
        print(f"(the oracle's own fp16-rounded loop vs its fp32 loop after {STEPS} steps: {e16:.3e})")
    assert max(errs.values()) < TOL, errs
    assert ef < TOL, ef


def test_full_loop_frame_sharded_world4_vs_oracle(world):
    """2-way CFG x 2 frame shards (13 + 12 frames) as four virtual ranks: each rank's frames after every marked step, and the
    clip gathered after the loop, against the oracle (and the gathered clip against the single-rank default order)"""
    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm, ThreadWorld
    nranks = 4
    tw = ThreadWorld(nranks)
    outs, keeps, errors = [None] * nranks, [dict() for _ in range(nranks)], []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            outs[r] = _run(world, output_type="latent", parallel=FrameParallel(Layout(nranks, r, T), ThreadComm(tw, r)),
                           keep=keeps[r])
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            for b in tw.barriers.values():
                b.abort()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(1800)
    assert not errors, errors
    single = _run(world, output_type="latent")
    worst = 0.0
    for r in range(nranks):
        lay = Layout(nranks, r, T)
        errs = _report(f"world {nranks} rank {r} (half {lay.half}, frames {lay.f0}..{lay.f1 - 1})", keeps[r], world["trace"],
                       slice(lay.f0, lay.f1))
        e_all, e_single = rel(outs[r], world["trace"][-1]), rel(outs[r], single)
        print(f"world {nranks} rank {r}: gathered clip after {STEPS} steps rel-L2 {e_all:.3e} vs oracle, {e_single:.3e} vs single rank")
        worst = max(worst, e_all, *errs.values())
        assert e_single < TOL, (r, e_single)
    assert worst < TOL, worst
