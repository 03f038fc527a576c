"""The reference's pipeline surface beyond the tensor path (MOFA-Video-Traj/pipeline/pipeline.py:283-311 and the call sites
of run_gradio.py:98-116, :335-354): PIL inputs, ``output_type="pil"`` default, ``callback_on_step_end``, checkpoint-directory
loaders (``from_pretrained`` / ``from_unet``), the fp16-rounded-latents switch, and the long-video loop's extensions
(hybrid control inside the windows, VAE decode overlapped with the last denoise step)."""
import numpy as np
import pytest
import torch

from helpers import LDMK_CN, LDMK_UNET, TINY, TINY_CN, TINY_VAE, oracle_models, rel_l2, synthetic_inputs, synthetic_landmarks

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, H, W = 3, 256, 256


@pytest.fixture(scope="module")
def tiny():
    from mofa_video_amd.adapter import FlowControlNet
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    ou, oc, ov, sdu, sdc, sdv = oracle_models(TINY, seed=0, vae_cfg=TINY_VAE, cn_cfg=TINY_CN)
    mods = dict(vae=AutoencoderKLTemporalDecoder(sdv, TINY_VAE, DEV), unet=UNetSpatioTemporalConditionControlNetModel(sdu, TINY, DEV),
                controlnet=FlowControlNet(sdc, TINY_CN, DEV), scheduler=EulerDiscreteScheduler())
    return mods, (sdu, sdc, sdv), synthetic_inputs(T, H, W, cross_dim=TINY["cross_attention_dim"])


def _call(pipe, inp, **kw):
    args = dict(controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W, num_frames=T,
                num_inference_steps=2, decode_chunk_size=2, latents=inp["latents"],
                image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])
    args.update(kw)
    return pipe(None, **args)


def test_default_output_is_pil_and_pil_condition_is_accepted(tiny):
    """run_gradio.py:335-354: the same PIL image is `image` and `controlnet_condition`; frames[0][i] are PIL images"""
    from PIL import Image
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    mods, _, inp = tiny
    pipe = FlowControlNetPipeline(**mods)
    u8 = ((inp["cond"][0] * 0.5 + 0.5).clamp(0, 1) * 255).round().byte().permute(1, 2, 0).numpy()
    pil = Image.fromarray(u8)
    out = _call(pipe, inp, controlnet_condition=pil)
    assert isinstance(out.frames, list) and len(out.frames) == 1 and len(out.frames[0]) == T
    assert isinstance(out.frames[0][0], Image.Image) and out.frames[0][0].size == (W, H)
    # the PIL condition is the 8-bit quantisation of the tensor condition: same latents within that quantisation
    cond_q = torch.from_numpy(u8.astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0) * 2 - 1
    a = _call(pipe, inp, controlnet_condition=pil, output_type="latent").frames
    b = _call(pipe, inp, controlnet_condition=cond_q, output_type="latent").frames
    assert torch.equal(a, b)
    # a [0, 1] tensor is normalised to [-1, 1] as VaeImageProcessor.preprocess does (no negative values -> 2 x - 1)
    c = _call(pipe, inp, controlnet_condition=cond_q * 0.5 + 0.5, output_type="latent").frames
    assert rel_l2(c, b) < 1e-5
    # a PIL condition of another size is resized (lanczos) instead of rejected
    small = pil.resize((W // 2, H // 2))
    d = _call(pipe, inp, controlnet_condition=small, output_type="latent").frames
    assert torch.isfinite(d).all() and tuple(d.shape) == tuple(b.shape)


def test_call_under_inference_mode_equals_no_grad(tiny):
    """round-5 advice: a caller who wraps ``pipe(...)`` in ``torch.inference_mode()`` (tensors without a version counter) crashed on the
    first GroupNorm-producer convolution.  The whole call runs there and gives the bits of the ordinary (``no_grad``) call."""
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    mods, _, inp = tiny
    pipe = FlowControlNetPipeline(**mods)
    a = _call(pipe, inp, output_type="pt").frames
    with torch.inference_mode():
        b = _call(pipe, inp, output_type="pt").frames
    assert torch.equal(a[0], b[0])


def test_callback_on_step_end_and_fp16_rounding_switch(tiny):
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    mods, _, inp = tiny
    seen = []

    def cb(pipe, i, t, kw):
        seen.append((i, float(t), tuple(kw["latents"].shape)))
        return {"latents": kw["latents"] * 0.5} if i == 0 else {}
    pipe = FlowControlNetPipeline(**mods)
    base = _call(pipe, inp, output_type="latent").frames
    got = _call(pipe, inp, output_type="latent", callback_on_step_end=cb).frames
    assert [s[0] for s in seen] == [0, 1] and seen[0][2] == (1, T, 4, H // 8, W // 8)
    assert rel_l2(got, base) > 1e-2                                    # the returned latents were taken
    r16 = _call(FlowControlNetPipeline(**mods, round_latents_to_fp16=True), inp, output_type="latent").frames
    assert torch.equal(r16, r16.half().float())                        # the reference's per-step fp16 rounding
    assert 0 < rel_l2(r16, base) < 2e-3


def test_from_pretrained_directories_and_from_unet(tiny, tmp_path):
    """config.json + diffusion_pytorch_model.safetensors round trip (run_gradio.py:98-116) and ControlNetSDVModel.from_unet"""
    from mofa_video_amd import checkpoint, schema
    from mofa_video_amd.adapter import FlowControlNet
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    mods, (sdu, sdc, sdv), inp = tiny
    root = tmp_path / "svd"
    checkpoint.save_pretrained(str(root / "unet"), sdu, TINY, "UNetSpatioTemporalConditionModel")
    checkpoint.save_pretrained(str(root / "vae"), sdv, TINY_VAE, "AutoencoderKLTemporalDecoder", safe_serialization=False)
    unet = UNetSpatioTemporalConditionControlNetModel.from_pretrained(str(root), subfolder="unet", low_cpu_mem_usage=True,
                                                                      torch_dtype=torch.float16, variant="fp16")
    vae = AutoencoderKLTemporalDecoder.from_pretrained(str(root), subfolder="vae")
    ref = _call(FlowControlNetPipeline(**mods), inp, output_type="raw").frames
    got = _call(FlowControlNetPipeline(vae=vae, unet=unet, controlnet=mods["controlnet"], scheduler=mods["scheduler"]), inp,
                output_type="raw").frames
    assert torch.equal(ref, got)
    with pytest.raises(OSError):
        UNetSpatioTemporalConditionControlNetModel.from_pretrained(str(root), subfolder="nope")
    # from_unet: trunk copied from the UNet, zero-initialised output convolutions -> every residual is exactly zero
    cn = FlowControlNet.from_unet(sdu, config=TINY_CN)
    x = torch.randn(2, T, 8, H // 8, W // 8)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    cond2, flow2 = torch.cat([inp["cond"]] * 2), torch.cat([inp["flow"]] * 2)
    gd, gm, _, _ = cn(x.to(DEV), torch.tensor(0.8), inp["image_embeddings"].to(DEV), ids.to(DEV), controlnet_cond=cond2.to(DEV),
                      controlnet_flow=flow2.to(DEV), return_dict=False)
    assert all(float(g.abs().max()) == 0.0 for g in list(gd) + [gm])
    sd_cn = checkpoint.controlnet_state_dict_from_unet(sdu, schema.controlnet_schema(TINY_CN))
    k = "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    assert torch.equal(sd_cn[k], sdu[k]) and "up_blocks.0.resnets.0.spatial_res_block.conv1.weight" not in sd_cn


def test_too_many_frames_fails_at_entry_with_a_clear_message(tiny):
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    mods, _, inp = tiny
    with pytest.raises(ValueError, match="temporal attention"):
        FlowControlNetPipeline(**mods)(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W,
                                       num_frames=40, latents=torch.zeros(1, 40, 4, H // 8, W // 8),
                                       image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"])


# ---- long-video loop extensions (BASELINE config 5) ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def ldmk():
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import FlowControlNet, LandmarkFlowControlNet
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    from oracle.controlnet import FlowControlNet as OFlow
    from oracle.ldmk import LandmarkFlowControlNet as OLdmk
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    sdl = schema.synthetic_state_dict(schema.ldmk_controlnet_schema(LDMK_CN), seed=11)
    sdt = schema.synthetic_state_dict(schema.controlnet_schema(LDMK_CN), seed=12)
    sdu = schema.synthetic_state_dict(schema.unet_schema(LDMK_UNET), seed=10)
    sdv = schema.synthetic_state_dict(schema.vae_decoder_schema(**TINY_VAE), seed=13)
    of, od, ou = OLdmk(**LDMK_CN), OFlow(**LDMK_CN), OUnet(**LDMK_UNET)
    of.load_state_dict({k: t.float() for k, t in sdl.items()})
    od.load_state_dict({k: t.float() for k, t in sdt.items()})
    ou.load_state_dict({k: t.float() for k, t in sdu.items()})
    return (of.eval(), od.eval(), ou.eval(), LandmarkFlowControlNet(sdl, LDMK_CN, DEV), FlowControlNet(sdt, LDMK_CN, DEV),
            UNetSpatioTemporalConditionControlNetModel(sdu, LDMK_UNET, DEV), AutoencoderKLTemporalDecoder(sdv, TINY_VAE, DEV))


def _long_inputs(N):
    cross = LDMK_CN["cross_attention_dim"]
    inp = synthetic_inputs(N, H, W, cross_dim=cross, seed=46)
    lm = synthetic_landmarks(N, H, W, seed=47)
    drag = synthetic_inputs(N, H, W, cross_dim=cross, seed=48)["flow"] * 0.5
    mask = torch.zeros(1, 1, H, W)
    mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    return inp, lm, drag, mask


def test_keypoint_loop_with_hybrid_control_vs_oracle(ldmk):
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_keypoint_loop
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    of, od, ou, hf, hd, hu, hv = ldmk
    N, win, stride = 6, 4, 2
    inp, lm, drag, mask = _long_inputs(N)
    with torch.no_grad():
        ref = denoise_keypoint_loop(ou, of, OSch(), inp["latents"], inp["image_latents"], inp["image_embeddings"], inp["cond"],
                                    inp["flow"], lm, window_size=win, stride=stride, num_inference_steps=2,
                                    drag_controlnet=od, drag_flow=drag, mask=mask, ctrl_scale_traj=0.8)
    pipe = KeypointFlowControlNetPipeline(unet=hu, controlnet=hf, drag_controlnet=hd, scheduler=EulerDiscreteScheduler())
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
               stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"],
               output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"],
               drag_flow=drag, mask=mask, ctrl_scale_traj=0.8).frames
    e = rel_l2(out, ref)
    print(f"keypoint loop, hybrid control, latents after 2 steps: rel-L2 {e:.3e}")
    assert e < 2e-2, e


def test_keypoint_overlapped_decode_is_bit_identical_and_callbacks_run(ldmk):
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu, hv = ldmk
    N, win, stride = 8, 4, 2
    inp, lm, drag, mask = _long_inputs(N)

    def run(overlap, **kw):
        pipe = KeypointFlowControlNetPipeline(vae=hv, unet=hu, controlnet=hf, scheduler=EulerDiscreteScheduler(),
                                              overlap_decode=overlap)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                    stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, decode_chunk_size=2,
                    latents=inp["latents"], output_type="raw", image_embeddings=inp["image_embeddings"],
                    image_latents=inp["image_latents"], **kw).frames
    a, b = run(False), run(True)
    assert tuple(a.shape) == (1, 3, N, H, W) and torch.isfinite(a).all()
    assert torch.equal(a, b)                                            # same kernels, same inputs, another stream
    calls = []
    c = run(True, callback_on_step_end=lambda p, i, t, kw: calls.append(i))
    assert calls == [0, 1] and torch.equal(a, c)
    with pytest.raises(ValueError, match="stride"):
        KeypointFlowControlNetPipeline(vae=hv, unet=hu, controlnet=hf, scheduler=EulerDiscreteScheduler())(
            None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=4, stride=4,
            height=H, width=W, num_frames=N, latents=inp["latents"], image_embeddings=inp["image_embeddings"],
            image_latents=inp["image_latents"])


def _run_ranks(world, fn):
    """fn(rank, thread_world) on ``world`` virtual ranks (threads of this process, one GPU)"""
    import threading

    from mofa_video_amd.parallel import ThreadWorld
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            results[r] = fn(r, tw)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            for b in tw.barriers.values():
                b.abort()
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    return results


@pytest.mark.parametrize("world", [2, 3])
def test_keypoint_window_parallel_decode_overlaps_last_step(ldmk, world):
    """Window-parallel long video with fewer ranks than windows: in the last step the frames below the merged views are
    final after every round, their VAE chunks are decoded on a second stream during the next round (idle ranks first) and
    the rest is dealt after the loop.  Every chunk is decoded exactly once, by the rank the common table names, and equals
    the single-rank frames bit for bit."""
    from mofa_video_amd.parallel import ThreadComm, WindowParallel
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu, hv = ldmk
    N, win, stride, chunk = 10, 4, 2, 2                                  # 4 distinct windows, 5 decode chunks
    inp, lm, drag, mask = _long_inputs(N)

    def run(parallel=None, overlap=True):
        pipe = KeypointFlowControlNetPipeline(vae=hv, unet=hu, controlnet=hf, scheduler=EulerDiscreteScheduler(), parallel=parallel,
                                              overlap_decode=overlap)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                    stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, decode_chunk_size=chunk,
                    latents=inp["latents"], output_type="raw", image_embeddings=inp["image_embeddings"],
                    image_latents=inp["image_latents"]).frames
    ref = run()                                                          # [1, 3, N, H, W]
    for overlap in (True, False):
        res = _run_ranks(world, lambda r, tw: run(WindowParallel(ThreadComm(tw, r), r, world), overlap))
        seen = {}
        for r, chunks in enumerate(res):
            for s0, fr in chunks:
                assert s0 not in seen, (s0, r, seen)
                seen[s0] = r
                assert torch.equal(fr, ref[0, :, s0:s0 + fr.shape[0]].permute(1, 0, 2, 3)), (overlap, r, s0)
        assert sorted(seen) == list(range(0, N, chunk)), seen
        print(f"window-parallel world {world} overlap {overlap}: chunk first frame -> rank {seen}")


@pytest.mark.parametrize("world", [2, 4])
def test_keypoint_several_windows_frame_sharded(ldmk, world):
    """Several windows under parallel.FrameParallel (Layout of window_size frames): every window runs on all ranks, 2-way CFG x
    frame shards, and is gathered before the overlap average; latents as on one rank (GroupNorm summation order differs)
    and every decode chunk comes back from exactly one rank."""
    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu, hv = ldmk
    N, win, stride, chunk = 8, 4, 2, 2
    inp, lm, drag, mask = _long_inputs(N)

    def run(parallel=None, output_type="latent"):
        pipe = KeypointFlowControlNetPipeline(vae=hv, unet=hu, controlnet=hf, drag_controlnet=hd, scheduler=EulerDiscreteScheduler(),
                                              parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                    stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, decode_chunk_size=chunk,
                    latents=inp["latents"], output_type=output_type, image_embeddings=inp["image_embeddings"],
                    image_latents=inp["image_latents"], drag_flow=drag, mask=mask, ctrl_scale_traj=0.8).frames
    ref = run()
    res = _run_ranks(world, lambda r, tw: run(FrameParallel(Layout(world, r, win), ThreadComm(tw, r))))
    for r, o in enumerate(res):
        e = rel_l2(o, ref)
        print(f"keypoint loop, 3 windows frame-sharded, world {world} rank {r}: latents rel-L2 vs single rank {e:.3e}")
        assert tuple(o.shape) == tuple(ref.shape) and e < 2e-3, (r, e)
    ref_frames = run(output_type="raw")
    res = _run_ranks(world, lambda r, tw: run(FrameParallel(Layout(world, r, win), ThreadComm(tw, r)), "raw"))
    seen = set()
    for r, chunks in enumerate(res):
        for s0, fr in chunks:
            assert s0 not in seen
            seen.add(s0)
            e = rel_l2(fr, ref_frames[0, :, s0:s0 + fr.shape[0]].permute(1, 0, 2, 3))
            assert e < 1e-2, (r, s0, e)
    assert sorted(seen) == list(range(0, N, chunk))


@pytest.mark.parametrize("world,g", [(4, 2), (8, 4)])
def test_keypoint_grouped_windows(ldmk, world, g):
    """parallel.GroupedWindowParallel: world / g groups of g ranks, the windows of a step dealt to the groups, every group
    frame-parallel on its window (g = 2: the CFG pair; g = 4: 2-way CFG x 2 frame shards under the two-thread turn token).
    Latents as on one rank, every decode chunk from exactly one of the ``world`` ranks."""
    from mofa_video_amd.parallel import GroupedWindowParallel, ThreadComm
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu, hv = ldmk
    N, win, stride, chunk = 10, 4, 2, 2                                  # 4 distinct windows, 5 decode chunks
    inp, lm, drag, mask = _long_inputs(N)

    def run(parallel=None, output_type="latent"):
        pipe = KeypointFlowControlNetPipeline(vae=hv, unet=hu, controlnet=hf, drag_controlnet=hd, scheduler=EulerDiscreteScheduler(),
                                              parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                    stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, decode_chunk_size=chunk,
                    latents=inp["latents"], output_type=output_type, image_embeddings=inp["image_embeddings"],
                    image_latents=inp["image_latents"], drag_flow=drag, mask=mask, ctrl_scale_traj=0.8).frames
    ref, ref_frames = run(), run(output_type="raw")
    res = _run_ranks(world, lambda r, tw: run(GroupedWindowParallel(ThreadComm(tw, r), r, world, g, win)))
    for r, o in enumerate(res):
        e = rel_l2(o, ref)
        print(f"keypoint loop, {world // g} groups x {g} ranks, rank {r}: latents rel-L2 vs single rank {e:.3e}")
        assert tuple(o.shape) == tuple(ref.shape) and e < 2e-3, (r, e)
    res = _run_ranks(world, lambda r, tw: run(GroupedWindowParallel(ThreadComm(tw, r), r, world, g, win), "raw"))
    seen = {}
    for r, chunks in enumerate(res):
        for s0, fr in chunks:
            assert s0 not in seen, (s0, r, seen)
            seen[s0] = r
            assert rel_l2(fr, ref_frames[0, :, s0:s0 + fr.shape[0]].permute(1, 0, 2, 3)) < 1e-2, (r, s0)
    assert sorted(seen) == list(range(0, N, chunk)), seen
