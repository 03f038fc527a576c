"""BASELINE.json configs[0], the designated end-to-end parity case (SURVEY.md 8d "Config 1"): the FULL-WIDTH SVD-XT UNet +
MOFA ControlNet (Traj) + temporal VAE decoder, 8 frames at 256x256, 2 denoise steps -- GPU fp16 through the C ABI against
the fp32 CPU oracle on the same seeded weights (reference checkpoint layout, fp16-valued) and inputs.  Unlike the reduced
TINY cases of test_model_gpu.py this runs the real channel widths (320 / 640 / 1280, heads (5, 10, 20, 20) in the UNet and
(5, 10, 10, 20) in the adapter trunk), K up to 23 040 and the 192x128 / 256x256 implicit-GEMM tiles the bench runs.

Stated fp16 tolerance (fp16 storage + fp32 accumulate vs the fp32 oracle): relative L2 <= 2e-2 on the latents after the
loop, <= 3e-2 on the decoded frames end to end (2e-2 with the decoder isolated on the oracle's latents).
The CPU side costs about a minute on 32 host threads (2 x ~6.4 TFLOP denoise steps + ~6 TFLOP decode).
"""
import os

import pytest
import torch

from helpers import oracle_models, rel_l2, synthetic_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, H, W, STEPS, CHUNK = 8, 256, 256, 2, 8


@pytest.fixture(scope="module")
def full_width():
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    ou, oc, ov, sdu, sdc, sdv = oracle_models(None, seed=0)          # default configs = the SVD-XT architecture
    assert sum(p.numel() for p in ou.parameters()) == 1524623082     # the public SVD-XT UNet size
    inp = synthetic_inputs(T, H, W, cross_dim=1024)
    from oracle.pipeline import denoise
    from oracle.scheduler import EulerDiscreteScheduler as OSch
    from oracle.vae import decode_latents as odecode
    with torch.no_grad():
        ref_lat = denoise(ou, oc, OSch(), inp["latents"], inp["image_latents"], inp["image_embeddings"], inp["cond"],
                          inp["flow"], num_inference_steps=STEPS)
        ref_frames = odecode(ov, ref_lat, T, decode_chunk_size=CHUNK)
    del ou, oc, ov
    from mofa_video_amd.adapter import FlowControlNet
    from mofa_video_amd.pipeline import FlowControlNetPipeline
    from mofa_video_amd.scheduler import EulerDiscreteScheduler
    from mofa_video_amd.unet import UNetSpatioTemporalConditionControlNetModel
    from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
    pipe = FlowControlNetPipeline(vae=AutoencoderKLTemporalDecoder(sdv, None, DEV),
                                  unet=UNetSpatioTemporalConditionControlNetModel(sdu, None, DEV),
                                  controlnet=FlowControlNet(sdc, None, DEV), scheduler=EulerDiscreteScheduler())
    del sdu, sdc, sdv
    return pipe, inp, ref_lat, ref_frames


def _run(pipe, inp, output_type):
    return pipe(None, controlnet_condition=inp["cond"], controlnet_flow=inp["flow"], height=H, width=W, num_frames=T,
                num_inference_steps=STEPS, decode_chunk_size=CHUNK, latents=inp["latents"], output_type=output_type,
                image_embeddings=inp["image_embeddings"], image_latents=inp["image_latents"]).frames


def test_config1_latents_and_frames_vs_cpu_oracle(full_width):
    from mofa_video_amd.vae import decode_latents
    pipe, inp, ref_lat, ref_frames = full_width
    lat = _run(pipe, inp, "latent")
    e = rel_l2(lat, ref_lat)
    print(f"config 1 (full width, {T} f {H}x{W}, {STEPS} steps): latents rel-L2 {e:.3e}")
    assert tuple(lat.shape) == tuple(ref_lat.shape) == (1, T, 4, H // 8, W // 8)
    assert e < 2e-2, e
    fr_iso = decode_latents(pipe.vae, ref_lat.to(DEV), T, CHUNK)       # decoder alone, on the oracle's latents
    e2 = rel_l2(fr_iso, ref_frames)
    print(f"config 1: decoded frames (oracle latents) rel-L2 {e2:.3e}")
    assert tuple(fr_iso.shape) == tuple(ref_frames.shape) == (1, 3, T, H, W)
    assert e2 < 2e-2, e2
    fr = _run(pipe, inp, "raw")                                         # the whole call: loop + chunked decode
    e3 = rel_l2(fr, ref_frames)
    print(f"config 1: decoded frames (end to end) rel-L2 {e3:.3e}")
    assert e3 < 3e-2, e3


@pytest.mark.parametrize("tile", ["192x128", "256x256", "256x320"])
def test_config1_same_result_on_every_forced_tile(full_width, tile):
    """the whole full-width loop with every implicit-GEMM launch forced onto one tile (ineligible launches -- unaligned rows,
    activation on a residual kind -- keep the default): same latents within the fp16 tolerance"""
    from mofa_video_amd import lib, ops
    pipe, inp, ref_lat, _ = full_width
    forced = {"192x128": lib.TILE_192X128, "256x256": lib.TILE_256X256, "256x320": lib.TILE_256X320}[tile]
    orig = ops.igemm

    def igemm_forced(*a, **kw):
        if kw.pop("tile", None) is None:
            try:
                return orig(*a, tile=forced, **kw)
            except lib.MofaHipError:                                    # MOFA_EINVAL: this launch is not eligible for the tile
                pass
        return orig(*a, **kw)
    ops.igemm = igemm_forced
    try:
        lat = _run(pipe, inp, "latent")
    finally:
        ops.igemm = orig
    e = rel_l2(lat, ref_lat)
    print(f"config 1, every igemm on the {tile} tile: latents rel-L2 {e:.3e}")
    assert e < 2e-2, e
