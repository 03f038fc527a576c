"""Multi-step, full-geometry, full-width parity for BASELINE configs 3, 4 and 5 (round-4 verdict, "What's missing" 1) and the
8-rank layout of config 2 (item 8): the product's loops against the oracle's loops run on the GPU in fp32 (the harness of
tests/test_fullgeom_gpu.py: MIOpen off, exact chunked attention), same seeded weights, the bench's inputs.

  config 4  ``HybridFlowControlNetPipeline``: 25 steps at 25 f 576x1024, landmark + trajectory adapters blended by the 25 %
            mask at every residual scale (MOFA-Video-Hybrid/pipeline/pipeline.py:445-507, blend :479-489) against
            ``oracle.pipeline.denoise_hybrid``; latents after steps 1/5/.../25, the per-step update, decoded frames.
  config 3  ``KeypointFlowControlNetPipeline`` with ONE window = the clip (25 f, window 25, stride 12: the reference builds
            the views (1,25),(1,25) and averages the two identical results, svdxt_pipeline_ctrlnet_loop.py:426-429, :500-511),
            25 steps, against ``oracle.pipeline.denoise_keypoint_loop``.
  config 5  the window loop proper at full width and 576x1024: 49 frames, window 25, stride 12 = views (1,25) (13,37) (25,49)
            (25,49), hybrid control in every window, ``LONG_STEPS`` = 8 steps with the ``_step_index`` rewind and the overlap
            average after every step, against ``denoise_keypoint_loop(drag_controlnet=...)``; merged latents after every
            step and the decoded frames of the first and the last chunk (the VAE is per-chunk; all 7 chunks in the product).
  world 8   config 2 on the BASELINE layout (2-way CFG x 4 frame shards of 7/6/6/6) as eight virtual ranks, 10 steps; config 4
            (Hybrid) on the same 8-rank layout for all 25 steps ("frames sharded over 8 x MI355X", BASELINE configs[3]).
  config 5  at its FULL length: 97 frames, 7 distinct windows + the repeated last view, 2 steps, on one GPU and with the windows
            dealt to 8 and 4 virtual ranks incl. the sharded / overlapped VAE decode (BASELINE configs[4]).

The oracle's repeated views are served from the first evaluation of the same view in the same step
(``reuse_identical_views``: the networks are deterministic functions of identical inputs; Euler step, rewind and merge still
run per view) -- checked against the literal loop in tests/test_oracle_structure.py (CPU).

Stated fp16 tolerance: rel-L2 <= 2e-2 for latents after any number of steps and for decoded frames.
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
LONG_FRAMES = int(os.environ.get("MOFA_FULLLOOP_LONG_FRAMES", "49"))
LONG_STEPS = int(os.environ.get("MOFA_FULLLOOP_LONG_STEPS", "8"))
W8_STEPS = min(int(os.environ.get("MOFA_FULLLOOP_W8_STEPS", "10")), STEPS)   # opt-in: 25 (profiles/r06_fullloop_full_steps.log)
TOL = 2e-2


def rel(a, b):
    a, b = a.to(DEV, torch.float32), b.to(DEV, torch.float32)
    assert tuple(a.shape) == tuple(b.shape), (a.shape, b.shape)
    assert bool(torch.isfinite(a).all()), "non-finite product output"
    return float((a - b).norm() / (b.norm() + 1e-12))


def marks(steps):
    return sorted({s for s in (1, 5, 10, 15, 20, 25) if s <= steps} | {steps})


def report(name, keep, trace, x0, steps, sl=slice(None)):
    """latents after the marked steps, and the worst per-step UPDATE error: (x_s - x_{s-1}) of the product against the
    oracle's -- the update is sigma-weighted model output (scheduling_euler_discrete_karras_fix.py:481-520), so unlike the
    latents themselves it does not hide behind the sigma = 700 noise both sides share in the early steps"""
    errs = {s: rel(keep[s], trace[s - 1][:, sl]) for s in marks(steps)}
    upd = {}
    for s in range(1, steps + 1):
        pa, pb = (keep[s - 1] if s > 1 else x0[:, sl]), (trace[s - 2][:, sl] if s > 1 else x0[:, sl])
        upd[s] = rel(keep[s].to(DEV) - pa.to(DEV), trace[s - 1][:, sl] - pb)
    print(f"{name}: latents rel-L2 vs oracle after step " + ", ".join(f"{s}: {e:.3e}" for s, e in errs.items()))
    print(f"{name}: per-step update rel-L2 " + ", ".join(f"{s}: {upd[s]:.2e}" for s in marks(steps)) +
          f"; worst {max(upd.values()):.3e} at step {max(upd, key=upd.get)}")
    return errs, upd


@pytest.fixture(scope="module")
def world():
    """inputs of configs 4 / 5, shared fp16-valued state dicts, product modules and the oracle's modules (fp32, on the GPU)"""
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import FlowControlNet, LandmarkFlowControlNet
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    from oracle.controlnet import FlowControlNet as OCn
    from oracle.ldmk import LandmarkFlowControlNet as OLdmk
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    dev = torch.device(DEV)
    mk = lambda sch, seed: schema.synthetic_state_dict(sch, seed=seed, device=DEV)   # noqa: E731
    sds = dict(unet=mk(schema.unet_schema(), 0), cn=mk(schema.controlnet_schema(), 1), vae=mk(schema.vae_decoder_schema(), 2),
               ldmk=mk(schema.ldmk_controlnet_schema(), 7))
    hip = dict(unet=UNetSpatioTemporalConditionControlNetModel(sds["unet"], None, DEV), cn=FlowControlNet(sds["cn"], None, DEV),
               vae=AutoencoderKLTemporalDecoder(sds["vae"], None, DEV), ldmk=LandmarkFlowControlNet(sds["ldmk"], None, DEV))
    with exact_fp32_gpu():
        ora = dict(unet=gpu_oracle(OUnet, sds["unet"]), ldmk=gpu_oracle(OLdmk, sds["ldmk"]), cn=gpu_oracle(OCn, sds["cn"]))
    vae_sd = sds["vae"]
    del sds
    torch.cuda.empty_cache()
    inp4 = bench.config_inputs(dev, 4)
    il2 = torch.cat([torch.zeros_like(inp4["image_latents"]), inp4["image_latents"]])
    emb2 = torch.cat([torch.zeros_like(inp4["image_embeddings"]), inp4["image_embeddings"]])
    yield dict(hip=hip, ora=ora, vae_sd=vae_sd, inp4=inp4, il2=il2, emb2=emb2)
    ora.clear()
    hip.clear()
    torch.cuda.empty_cache()


def _oracle_frames(world, latents, chunks):
    """the oracle's temporal VAE (fp32 on the GPU) on the given frame ranges of ``latents`` [1,N,4,h,w] -> {(s0, s1): [1,3,n,H,W]}"""
    from oracle.vae import AutoencoderKLTemporalDecoder as OVae
    from oracle.vae import decode_latents as odecode
    out = {}
    with exact_fp32_gpu():
        ov = gpu_oracle(OVae, world["vae_sd"])
        for (s0, s1) in chunks:
            out[(s0, s1)] = odecode(ov, latents[:, s0:s1], s1 - s0, s1 - s0)
        del ov
    return out


def _timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    return r, e0.elapsed_time(e1) * 1e-3


def _keeper(keep):
    def cb(p, i, t, kw):
        keep[i + 1] = kw["latents"].clone()
        return {}
    return cb


# ---------------------------------------------------------------------------------------------------------------------
HYBRID_SCALES = dict(ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1)


def _hybrid_trace(world):
    """the oracle's Hybrid loop (config 4), latents after every step; computed once per module"""
    if "trace4" not in world:
        from oracle.pipeline import denoise_hybrid
        from oracle.scheduler import EulerDiscreteScheduler as OSch
        inp, ora = world["inp4"], world["ora"]
        with exact_fp32_gpu():
            (_, trace), sec = _timed(lambda: denoise_hybrid(
                ora["unet"], ora["ldmk"], ora["cn"], OSch(), inp["latents"], world["il2"], world["emb2"], inp["cond"], inp["flow"],
                inp["landmarks"], inp["drag_flow"], inp["mask"], num_inference_steps=STEPS, return_trace=True, **HYBRID_SCALES))
        print(f"oracle (Hybrid) on the GPU in fp32: {STEPS} steps at {T} f {H}x{W} in {sec:.1f} s")
        world["trace4"] = trace
    return world["trace4"]


def test_config4_hybrid_full_loop_vs_oracle(world):
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    inp, hip = world["inp4"], world["hip"]
    scales = HYBRID_SCALES
    trace = _hybrid_trace(world)
    pipe = HybridFlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], face_controlnet=hip["ldmk"], drag_controlnet=hip["cn"],
                                        scheduler=EulerDiscreteScheduler())
    keep = {}
    frames = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=inp["landmarks"],
                  drag_flow=inp["drag_flow"], mask=inp["mask"], height=H, width=W, num_frames=T, num_inference_steps=STEPS,
                  decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type="raw", callback_on_step_end=_keeper(keep),
                  image_embeddings=world["emb2"], image_latents=world["il2"], **scales).frames
    errs, upd = report(f"config 4 (Hybrid), {STEPS} steps @ {T}f {H}x{W}", keep, trace, _x0(inp["latents"], STEPS), STEPS)
    chunks = [(s0, min(s0 + bench.CHUNK, T)) for s0 in range(0, T, bench.CHUNK)]
    ref = torch.cat([v for _, v in sorted(_oracle_frames(world, trace[-1], chunks).items())], dim=2)
    ef = rel(frames, ref)
    print(f"config 4 (Hybrid), {STEPS} steps: decoded frames {tuple(frames.shape)} rel-L2 vs oracle {ef:.3e}")
    assert max(errs.values()) < TOL and max(upd.values()) < TOL and ef < TOL, (errs, max(upd.values()), ef)


def _x0(latents, steps):
    """the loop's starting point: latents * init_noise_sigma (pipeline.py:272)"""
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    s = OSch()
    s.set_timesteps(steps)
    return latents.to(DEV, torch.float32) * float(s.init_noise_sigma)


def test_config3_keypoint_single_window_full_loop_vs_oracle(world):
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_keypoint_loop
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    inp, hip, ora = world["inp4"], world["hip"], world["ora"]
    with exact_fp32_gpu():
        (_, trace), sec = _timed(lambda: denoise_keypoint_loop(
            ora["unet"], ora["ldmk"], OSch(), inp["latents"], world["il2"], world["emb2"], inp["cond"], inp["flow"],
            inp["landmarks"], window_size=T, stride=T // 2, num_inference_steps=STEPS, return_trace=True,
            reuse_identical_views=True))
    print(f"oracle (Keypoint, one window) on the GPU in fp32: {STEPS} steps at {T} f {H}x{W} in {sec:.1f} s")
    pipe = KeypointFlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], controlnet=hip["ldmk"],
                                          scheduler=EulerDiscreteScheduler())
    keep = {}
    frames = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=inp["landmarks"],
                  window_size=T, stride=T // 2, height=H, width=W, num_frames=T, num_inference_steps=STEPS,
                  decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type="raw", callback_on_step_end=_keeper(keep),
                  image_embeddings=world["emb2"], image_latents=world["il2"]).frames
    errs, upd = report(f"config 3 (Keypoint, one window), {STEPS} steps @ {T}f {H}x{W}", keep, trace, _x0(inp["latents"], STEPS),
                       STEPS)
    ref = _oracle_frames(world, trace[-1], [(0, bench.CHUNK), (T - 1, T)])          # first chunk (8 f) and the 1-frame tail
    ef = max(rel(frames[:, :, s0:s1], v) for (s0, s1), v in ref.items())
    print(f"config 3 (Keypoint), {STEPS} steps: decoded frames 0..{bench.CHUNK - 1} and {T - 1} rel-L2 vs oracle {ef:.3e}")
    assert max(errs.values()) < TOL and max(upd.values()) < TOL and ef < TOL, (errs, max(upd.values()), ef)


def test_config5_window_loop_hybrid_control_vs_oracle(world):
    """49 frames, 3 + 1 views, hybrid control per window, ``LONG_STEPS`` steps: the rewind of ``_step_index`` between the
    views of a step and the overlap average after it (svdxt_pipeline_ctrlnet_loop.py:499-511) at full geometry"""
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline, window_views
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_keypoint_loop
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    hip, ora = world["hip"], world["ora"]
    N, steps = LONG_FRAMES, LONG_STEPS
    old = bench.LONG_FRAMES
    bench.LONG_FRAMES = N
    try:
        inp = bench.config_inputs(torch.device(DEV), 5)
    finally:
        bench.LONG_FRAMES = old
    views = window_views(N, T, T // 2)
    assert len(set(views)) >= min(3, (N - T) // (T // 2) + 1), views
    with exact_fp32_gpu():
        (_, trace), sec = _timed(lambda: denoise_keypoint_loop(
            ora["unet"], ora["ldmk"], OSch(), inp["latents"], world["il2"], world["emb2"], inp["cond"], inp["flow"],
            inp["landmarks"], window_size=T, stride=T // 2, num_inference_steps=steps, drag_controlnet=ora["cn"],
            drag_flow=inp["drag_flow"], mask=inp["mask"], ctrl_scale_traj=0.9, controlnet_cond_scale=1.05, return_trace=True,
            reuse_identical_views=True))
    print(f"oracle (window loop, hybrid control) on the GPU in fp32: views {views}, {steps} steps at {N} f {H}x{W} in {sec:.1f} s")
    pipe = KeypointFlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], controlnet=hip["ldmk"], drag_controlnet=hip["cn"],
                                          scheduler=EulerDiscreteScheduler())
    keep = {}
    frames = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=inp["landmarks"],
                  window_size=T, stride=T // 2, height=H, width=W, num_frames=N, num_inference_steps=steps,
                  decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type="raw", callback_on_step_end=_keeper(keep),
                  image_embeddings=world["emb2"], image_latents=world["il2"], drag_flow=inp["drag_flow"], mask=inp["mask"],
                  ctrl_scale_traj=0.9, controlnet_cond_scale=1.05).frames
    assert tuple(frames.shape) == (1, 3, N, H, W), frames.shape
    errs, upd = report(f"config 5 (window loop + hybrid control), {steps} steps @ {N}f {H}x{W}", keep, trace,
                       _x0(inp["latents"], steps), steps)
    # frames in the overlap of two windows are the ones the average touches: report them separately
    ov = [f for f in range(N) if sum((0 if i == 0 else t0) <= f < t1 for i, (t0, t1) in enumerate(views)) > 1]
    e_ov = rel(keep[steps][:, ov], trace[-1][:, ov])
    print(f"config 5: {len(ov)} overlap-averaged frames after {steps} steps rel-L2 {e_ov:.3e}")
    last0 = (N - 1) // bench.CHUNK * bench.CHUNK
    ref = _oracle_frames(world, trace[-1], [(0, bench.CHUNK), (last0, N)])
    ef = max(rel(frames[:, :, s0:s1], v) for (s0, s1), v in ref.items())
    print(f"config 5, {steps} steps: decoded frames 0..{bench.CHUNK - 1} and {last0}..{N - 1} rel-L2 vs oracle {ef:.3e}")
    assert max(errs.values()) < TOL and max(upd.values()) < TOL and e_ov < TOL and ef < TOL, (errs, max(upd.values()), e_ov, ef)


def test_world8_baseline_layout_10_steps_vs_oracle(world):
    """config 2 on 8 virtual ranks: 2-way CFG x 4 frame shards (7 / 6 / 6 / 6 frames), ``W8_STEPS`` steps against the oracle's
    plain loop (MOFA-Video-Traj/pipeline/pipeline.py:447-511)"""
    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm, ThreadWorld
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    inp, hip, ora = world["inp4"], world["hip"], world["ora"]
    steps, nranks = W8_STEPS, 8
    with exact_fp32_gpu():
        (_, trace), sec = _timed(lambda: denoise(ora["unet"], ora["cn"], OSch(), inp["latents"], world["il2"], world["emb2"],
                                                 inp["cond"], inp["drag_flow"], num_inference_steps=steps, return_trace=True))
    print(f"oracle (config 2) on the GPU in fp32: {steps} steps in {sec:.1f} s")
    tw = ThreadWorld(nranks)
    outs, keeps, errors = [None] * nranks, [dict() for _ in range(nranks)], []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            pipe = FlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], controlnet=hip["cn"], scheduler=EulerDiscreteScheduler(),
                                          parallel=FrameParallel(Layout(nranks, r, T), ThreadComm(tw, r)))
            outs[r] = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["drag_flow"], height=H, width=W,
                           num_frames=T, num_inference_steps=steps, decode_chunk_size=bench.CHUNK, latents=inp["latents"],
                           output_type="latent", callback_on_step_end=_keeper(keeps[r]), image_embeddings=world["emb2"],
                           image_latents=world["il2"]).frames
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
    x0 = _x0(inp["latents"], steps)
    worst, sizes = 0.0, []
    for r in range(nranks):
        lay = Layout(nranks, r, T)
        sizes.append(lay.f1 - lay.f0)
        errs, upd = report(f"world {nranks} rank {r} (half {lay.half}, frames {lay.f0}..{lay.f1 - 1})", keeps[r], trace, x0, steps,
                           slice(lay.f0, lay.f1))
        e_all = rel(outs[r], trace[-1])
        print(f"world {nranks} rank {r}: gathered clip after {steps} steps rel-L2 {e_all:.3e} vs oracle")
        worst = max(worst, e_all, *errs.values(), *upd.values())
    assert sizes[:4] == [7, 6, 6, 6], sizes
    assert worst < TOL, worst


def _virtual_ranks(nranks, fn):
    """fn(rank, thread_world) on ``nranks`` threads of this process sharing the GPU"""
    from mofa_video_amd.parallel import ThreadWorld
    tw = ThreadWorld(nranks)
    results, errors = [None] * nranks, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            results[r] = fn(r, tw)
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
    return results


def test_config4_hybrid_frames_sharded_over_8_ranks_vs_oracle(world):
    """BASELINE config 4 as it is worded: the Hybrid dual-adapter clip with its frames sharded over 8 ranks (2-way CFG x 4 frame
    shards of 7 / 6 / 6 / 6; both adapters warp and step only the rank's frames, the mask blend is per frame), all ``STEPS`` steps,
    every rank's shard after the marked steps and the gathered clip against the oracle's Hybrid loop"""
    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    inp, hip = world["inp4"], world["hip"]
    trace = _hybrid_trace(world)
    nranks = 8
    keeps = [dict() for _ in range(nranks)]

    def run(r, tw):
        pipe = HybridFlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], face_controlnet=hip["ldmk"], drag_controlnet=hip["cn"],
                                            scheduler=EulerDiscreteScheduler(),
                                            parallel=FrameParallel(Layout(nranks, r, T), ThreadComm(tw, r)))
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=inp["landmarks"],
                    drag_flow=inp["drag_flow"], mask=inp["mask"], height=H, width=W, num_frames=T, num_inference_steps=STEPS,
                    decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type="latent",
                    callback_on_step_end=_keeper(keeps[r]), image_embeddings=world["emb2"], image_latents=world["il2"],
                    **HYBRID_SCALES).frames
    outs = _virtual_ranks(nranks, run)
    x0 = _x0(inp["latents"], STEPS)
    worst = 0.0
    for r in range(nranks):
        lay = Layout(nranks, r, T)
        errs, upd = report(f"config 4 (Hybrid), world {nranks} rank {r} (half {lay.half}, frames {lay.f0}..{lay.f1 - 1})", keeps[r], trace,
                           x0, STEPS, slice(lay.f0, lay.f1))
        e_all = rel(outs[r], trace[-1])
        print(f"config 4 (Hybrid), world {nranks} rank {r}: gathered clip after {STEPS} steps rel-L2 {e_all:.3e} vs oracle")
        worst = max(worst, e_all, *errs.values(), *upd.values())
    assert worst < TOL, worst


def test_config5_full_length_97_frames_single_gpu_and_8_ranks_vs_oracle(world):
    """BASELINE config 5 at its full length: 97 frames = 7 distinct windows of 25 (stride 12) + the repeated last view, hybrid
    control in every window, 576x1024, full width, ``FULL_STEPS`` steps -- on one GPU, and with the windows dealt to 8 and to 4
    ranks (``parallel.WindowParallel``; 4 ranks = two rounds: the decode of finished chunks overlaps the second round of the last
    step on a second stream), latents and the first / last decoded chunks against the oracle"""
    from mofa_video_amd.parallel import ThreadComm, WindowParallel
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline, window_views
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_keypoint_loop
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    hip, ora = world["hip"], world["ora"]
    N, steps = bench.LONG_FRAMES, int(os.environ.get("MOFA_FULLLOOP_FULL_STEPS", "2"))
    inp = bench.config_inputs(torch.device(DEV), 5)
    views = window_views(N, T, T // 2)
    assert N == 97 and len(set(views)) == 7 and len(views) == 8, views
    kw = dict(ctrl_scale_traj=0.9, controlnet_cond_scale=1.05)
    with exact_fp32_gpu():
        (_, trace), sec = _timed(lambda: denoise_keypoint_loop(
            ora["unet"], ora["ldmk"], OSch(), inp["latents"], world["il2"], world["emb2"], inp["cond"], inp["flow"],
            inp["landmarks"], window_size=T, stride=T // 2, num_inference_steps=steps, drag_controlnet=ora["cn"],
            drag_flow=inp["drag_flow"], mask=inp["mask"], return_trace=True, reuse_identical_views=True, **kw))
    print(f"oracle (window loop, hybrid control) on the GPU in fp32: {len(views)} views, {steps} steps at {N} f {H}x{W} in {sec:.1f} s")
    last0 = (N - 1) // bench.CHUNK * bench.CHUNK
    ref = _oracle_frames(world, trace[-1], [(0, bench.CHUNK), (last0, N)])

    def run(parallel, output_type, keep=None):
        pipe = KeypointFlowControlNetPipeline(vae=hip["vae"], unet=hip["unet"], controlnet=hip["ldmk"], drag_controlnet=hip["cn"],
                                              scheduler=EulerDiscreteScheduler(), parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=inp["landmarks"],
                    window_size=T, stride=T // 2, height=H, width=W, num_frames=N, num_inference_steps=steps,
                    decode_chunk_size=bench.CHUNK, latents=inp["latents"], output_type=output_type,
                    callback_on_step_end=_keeper(keep) if keep is not None else None, image_embeddings=world["emb2"],
                    image_latents=world["il2"], drag_flow=inp["drag_flow"], mask=inp["mask"], **kw).frames
    # one GPU
    keep = {}
    frames = run(None, "raw", keep)
    errs, upd = report(f"config 5 at full length, {steps} steps @ {N}f {H}x{W}, one GPU", keep, trace, _x0(inp["latents"], steps), steps)
    ef = max(rel(frames[:, :, s0:s1], v) for (s0, s1), v in ref.items())
    print(f"config 5 at full length, one GPU: decoded frames 0..{bench.CHUNK - 1} and {last0}..{N - 1} rel-L2 vs oracle {ef:.3e}")
    assert max(errs.values()) < TOL and max(upd.values()) < TOL and ef < TOL, (errs, max(upd.values()), ef)
    del frames
    # windows dealt to 8 ranks (the BASELINE layout: one round, one rank idle in the loop) and to 4 (two rounds, overlapped decode)
    for nranks in (8, 4):
        lat = _virtual_ranks(nranks, lambda r, tw: run(WindowParallel(ThreadComm(tw, r), r, nranks), "latent"))
        e_lat = max(rel(o, trace[-1]) for o in lat)
        assert all(torch.equal(o, lat[0]) for o in lat), "ranks disagree on the merged latents"
        del lat
        chunks = _virtual_ranks(nranks, lambda r, tw: run(WindowParallel(ThreadComm(tw, r), r, nranks), "raw"))
        owner, ef = {}, 0.0
        for r, mine in enumerate(chunks):
            for s0, fr in mine:                                              # fr [n, 3, H, W]
                assert s0 not in owner, (s0, r, owner)
                owner[s0] = r
                for (a, b), v in ref.items():
                    if a == s0:
                        ef = max(ef, rel(fr.permute(1, 0, 2, 3).unsqueeze(0), v))
        assert sorted(owner) == list(range(0, N, bench.CHUNK)), owner
        print(f"config 5 at full length, windows over {nranks} ranks: latents rel-L2 vs oracle {e_lat:.3e}, decoded first / last chunk "
              f"{ef:.3e}; chunk -> rank {owner}")
        del chunks
        assert e_lat < TOL and 0.0 < ef < TOL, (nranks, e_lat, ef)
