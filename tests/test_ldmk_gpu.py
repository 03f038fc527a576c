"""GPU parity of the landmark MOFA-Adapter (config 3), the Hybrid dual-adapter pipeline (config 4) and the Keypoint
window loop (config 3/5 loop) against the CPU oracle on shared seeded fp16-valued weights.
Stated fp16 tolerance: rel-L2 <= 1e-2 per forward tensor, <= 2e-2 for latents after the loop."""
import pytest
import torch

from helpers import LDMK_CN, LDMK_UNET, rel_l2, synthetic_inputs, synthetic_landmarks

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = W = 256
CROSS = LDMK_CN["cross_attention_dim"]


@pytest.fixture(scope="module")
def models():
    from mofa_video_amd import schema
    from mofa_video_amd.adapter import FlowControlNet, LandmarkFlowControlNet
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from oracle.controlnet import FlowControlNet as OFlow
    from oracle.ldmk import LandmarkFlowControlNet as OLdmk
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel as OUnet
    sdl = schema.synthetic_state_dict(schema.ldmk_controlnet_schema(LDMK_CN), seed=11)
    sdt = schema.synthetic_state_dict(schema.controlnet_schema(LDMK_CN), seed=12)
    sdu = schema.synthetic_state_dict(schema.unet_schema(LDMK_UNET), seed=10)
    of, od, ou = OLdmk(**LDMK_CN), OFlow(**LDMK_CN), OUnet(**LDMK_UNET)
    of.load_state_dict({k: t.float() for k, t in sdl.items()})
    od.load_state_dict({k: t.float() for k, t in sdt.items()})
    ou.load_state_dict({k: t.float() for k, t in sdu.items()})
    hf, hd = LandmarkFlowControlNet(sdl, LDMK_CN, DEV), FlowControlNet(sdt, LDMK_CN, DEV)
    hu = UNetSpatioTemporalConditionControlNetModel(sdu, LDMK_UNET, DEV)
    return of.eval(), od.eval(), ou.eval(), hf, hd, hu


def test_landmark_adapter_forward(models):
    of, od, ou, hf, hd, hu = models
    T = 3
    inp = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=43)
    lm = synthetic_landmarks(T, H, W, seed=44)
    sigma = 3.0
    x = torch.cat([torch.cat([inp["latents"] * 5.0] * 2) / (sigma ** 2 + 1) ** 0.5,
                   inp["image_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)], dim=2)
    ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
    cond2, flow2, lm2 = torch.cat([inp["cond"]] * 2), torch.cat([inp["flow"]] * 2), torch.cat([lm] * 2)
    with torch.no_grad():
        rd, rm, _, rocc = of(x, torch.tensor(0.8), inp["image_embeddings"], ids, controlnet_cond=cond2,
                             controlnet_flow=flow2, landmarks=lm2, return_dict=False, conditioning_scale=0.9)
    gd, gm, _, gocc = hf(x.to(DEV), torch.tensor(0.8), inp["image_embeddings"].to(DEV), ids.to(DEV),
                         controlnet_cond=cond2.to(DEV), controlnet_flow=flow2.to(DEV), landmarks=lm2.to(DEV),
                         return_dict=False, conditioning_scale=0.9)
    for i, (r, g) in enumerate(zip(list(rd) + [rm], list(gd) + [gm])):
        e = rel_l2(g, r)
        print(f"ldmk residual {i}: rel-L2 {e:.3e}")
        assert tuple(g.shape) == tuple(r.shape) and e < 1e-2, (i, e)
    for lvl, (r, g) in enumerate(zip(rocc, gocc)):
        e = rel_l2(g, r)
        print(f"occlusion mask level {lvl}: rel-L2 {e:.3e}")
        assert tuple(g.shape) == tuple(r.shape) and e < 1e-2, (lvl, e)


def test_hybrid_pipeline(models):
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_hybrid
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    of, od, ou, hf, hd, hu = models
    T = 3
    inp = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=43)
    lm = synthetic_landmarks(T, H, W, seed=44)
    drag_flow = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=45)["flow"] * 0.5
    mask = torch.zeros(1, 1, H, W)
    mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    with torch.no_grad():
        ref = denoise_hybrid(ou, of, od, OSch(), inp["latents"], inp["image_latents"], inp["image_embeddings"],
                             inp["cond"], inp["flow"], lm, drag_flow, mask, num_inference_steps=2, ctrl_scale_traj=0.8,
                             ctrl_scale_ldmk=1.1)
    pipe = HybridFlowControlNetPipeline(unet=hu, face_controlnet=hf, drag_controlnet=hd, scheduler=EulerDiscreteScheduler())
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV),
               drag_flow=drag_flow, mask=mask, height=H, width=W, num_frames=T, num_inference_steps=2,
               latents=inp["latents"], output_type="latent", ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1,
               image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    e = rel_l2(out, ref)
    print(f"hybrid latents after 2 steps: rel-L2 {e:.3e}")
    assert e < 2e-2, e
    # both adapters' trunks on the second HIP stream beside the UNet encoder == the single-stream order, bit for bit; with the
    # decoder's CFG halves on the two streams as well (the default) half-size launches pick their own tiles: rounding noise
    assert pipe.overlap_adapter and pipe.split_decoder
    pipe.split_decoder = False
    out_ts = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV),
                  drag_flow=drag_flow, mask=mask, height=H, width=W, num_frames=T, num_inference_steps=2,
                  latents=inp["latents"], output_type="latent", ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1,
                  image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    assert rel_l2(out, out_ts) < 2e-3
    out = out_ts
    pipe.overlap_adapter = False
    out1 = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV),
                drag_flow=drag_flow, mask=mask, height=H, width=W, num_frames=T, num_inference_steps=2,
                latents=inp["latents"], output_type="latent", ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1,
                image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    assert torch.equal(out, out1)


def test_keypoint_window_loop(models):
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from oracle.pipeline import denoise_keypoint_loop
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    of, od, ou, hf, hd, hu = models
    N, win, stride = 6, 4, 2
    inp = synthetic_inputs(N, H, W, cross_dim=CROSS, seed=46)
    lm = synthetic_landmarks(N, H, W, seed=47)
    with torch.no_grad():
        ref = denoise_keypoint_loop(ou, of, OSch(), inp["latents"], inp["image_latents"], inp["image_embeddings"],
                                    inp["cond"], inp["flow"], lm, window_size=win, stride=stride, num_inference_steps=2)
    pipe = KeypointFlowControlNetPipeline(unet=hu, controlnet=hf, scheduler=EulerDiscreteScheduler())
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
               stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"],
               output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    e = rel_l2(out, ref)
    print(f"keypoint loop latents after 2 steps: rel-L2 {e:.3e}")
    assert e < 2e-2, e
    pipe.split_decoder = False
    out = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
               stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"],
               output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    pipe.overlap_adapter = False                                     # single-stream order: the same bits as trunk || encoder
    out1 = pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"],
                output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    assert torch.equal(out, out1)


@pytest.mark.parametrize("world", [2, 4])
def test_hybrid_pipeline_frame_sharded_equals_single_rank(models, world):
    """Hybrid (two adapters + mask blend) on virtual ranks of this GPU: 2-way CFG x frame shards must reproduce the
    single-rank latents (only the GroupNorm summation order differs)."""
    import threading

    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm, ThreadWorld
    from mofa_video_amd.pipeline import HybridFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu = models
    T = 4
    inp = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=43)
    lm = synthetic_landmarks(T, H, W, seed=44)
    drag_flow = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=45)["flow"] * 0.5
    mask = torch.zeros(1, 1, H, W)
    mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0

    def run(parallel=None):
        pipe = HybridFlowControlNetPipeline(unet=hu, face_controlnet=hf, drag_controlnet=hd, scheduler=EulerDiscreteScheduler(),
                                            parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV),
                    drag_flow=drag_flow, mask=mask, height=H, width=W, num_frames=T, num_inference_steps=2,
                    latents=inp["latents"], output_type="latent", ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1,
                    image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    ref = run()
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            results[r] = run(FrameParallel(Layout(world, r, T), ThreadComm(tw, r)))
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
    for r, o in enumerate(results):
        e = rel_l2(o, ref)
        print(f"hybrid world {world} rank {r}: latents rel-L2 vs single rank {e:.3e}")
        assert tuple(o.shape) == tuple(ref.shape) and e < 2e-3, (r, e)


@pytest.mark.parametrize("world", [2, 3])
def test_keypoint_loop_window_parallel_equals_single_rank(models, world):
    """Keypoint long-video loop with its distinct windows dealt to virtual ranks (parallel.WindowParallel): every rank
    must end with exactly the single-rank latents (same kernels on the same data, same averaging order)."""
    import threading

    from mofa_video_amd.parallel import ThreadComm, ThreadWorld, WindowParallel
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu = models
    N, win, stride = 6, 4, 2
    inp = synthetic_inputs(N, H, W, cross_dim=CROSS, seed=46)
    lm = synthetic_landmarks(N, H, W, seed=47)

    def run(parallel=None):
        pipe = KeypointFlowControlNetPipeline(unet=hu, controlnet=hf, scheduler=EulerDiscreteScheduler(), parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=win,
                    stride=stride, height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"],
                    output_type="latent", image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames
    ref = run()
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            results[r] = run(WindowParallel(ThreadComm(tw, r), r, world))
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
    for r, o in enumerate(results):
        assert torch.equal(o, ref), (r, rel_l2(o, ref))


@pytest.mark.parametrize("world,hybrid", [(2, False), (4, False), (4, True)])
def test_keypoint_single_window_frame_sharded_equals_single_rank(models, world, hybrid):
    """BASELINE config 3 is ONE window that is the whole clip (num_frames == window_size): with a parallel.FrameParallel the
    Keypoint pipeline frame-shards it (2-way CFG x frame shards) like the Traj / Hybrid pipelines -- it must reproduce the
    single-rank window loop (whose two identical views average to the stepped window)."""
    import threading

    from mofa_video_amd.parallel import FrameParallel, Layout, ThreadComm, ThreadWorld
    from mofa_video_amd.pipeline import KeypointFlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    of, od, ou, hf, hd, hu = models
    T = 5
    inp = synthetic_inputs(T, H, W, cross_dim=CROSS, seed=48)
    lm = synthetic_landmarks(T, H, W, seed=49)
    extra = {}
    if hybrid:
        mask = torch.zeros(1, 1, H, W)
        mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
        extra = dict(drag_flow=synthetic_inputs(T, H, W, cross_dim=CROSS, seed=50)["flow"] * 0.5, mask=mask, ctrl_scale_traj=0.8)

    def run(parallel=None, frames=T, window=T):
        pipe = KeypointFlowControlNetPipeline(unet=hu, controlnet=hf, drag_controlnet=hd if hybrid else None,
                                              scheduler=EulerDiscreteScheduler(), parallel=parallel)
        return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], landmarks=lm.to(DEV), window_size=window,
                    stride=max(window // 2, 1), height=H, width=W, num_frames=frames, num_inference_steps=2,
                    latents=inp["latents"], output_type="latent", image_embeddings=inp["image_embeddings"],
                    image_latents=inp["image_latents"], **extra).frames
    ref = run()
    tw = ThreadWorld(world)
    results, errors = [None] * world, []

    def worker(r):
        try:
            torch.cuda.set_device(0)
            results[r] = run(FrameParallel(Layout(world, r, T), ThreadComm(tw, r)))
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
    for r, o in enumerate(results):
        e = rel_l2(o, ref)
        print(f"keypoint single window (hybrid={hybrid}) world {world} rank {r}: latents rel-L2 vs single rank {e:.3e}")
        assert tuple(o.shape) == tuple(ref.shape) and e < 2e-3, (r, e)
    # a Layout built for another window size is refused (several windows: tests/test_pipeline_api_gpu.py)
    with pytest.raises(ValueError):
        run(FrameParallel(Layout(1, 0, T), ThreadComm(ThreadWorld(1), 0)), frames=T, window=T - 1)
