#!/usr/bin/env python
"""Golden fixtures for the landmark adapter, the Hybrid dual-adapter step and the Keypoint window loop, produced BY
RUNNING THE REFERENCE'S OWN CODE of the MOFA-Video-Hybrid / MOFA-Video-Keypoint trees in the build container
(same stubbing as tests/golden/make_golden.py: oracle blocks stand in for the absent diffusers, oracle/_ref for the
CuPy softsplat).  One tree per process (the trees share module names):

    python tests/golden/make_golden_ldmk.py hybrid      -> tests/golden/reference_golden_hybrid.pt
    python tests/golden/make_golden_ldmk.py keypoint    -> tests/golden/reference_golden_keypoint.pt
    python tests/golden/make_golden_ldmk.py             -> both (two subprocesses)

Reference code executed: models/ldmk_ctrlnet.py (FlowControlNet: ctor pieces, get_warped_frames, forward),
models/occlusion/hourglass.py (ForegroundMatting), models/traj_ctrlnet.py, models/controlnet_sdv.py,
models/unet_spatio_temporal_condition_controlnet.py, pipeline/pipeline.py (Hybrid __call__),
pipeline/svdxt_pipeline_ctrlnet_loop.py (Keypoint __call__), utils/scheduling_euler_discrete_karras_fix.py.
"""
import os
import subprocess
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

TREES = {"hybrid": "/root/reference/MOFA-Video-Hybrid", "keypoint": "/root/reference/MOFA-Video-Keypoint"}


def build_models(tree):
    import make_golden as MG
    from helpers import LDMK_CN, LDMK_UNET, TINY_VAE
    from mofa_video_amd import schema
    MG.REF = TREES[tree]
    MG.install_stubs()
    import models.ldmk_ctrlnet as LD
    import models.traj_ctrlnet as TR
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel

    kw = dict(block_out_channels=LDMK_CN["block_out_channels"], num_attention_heads=LDMK_CN["num_attention_heads"],
              cross_attention_dim=LDMK_CN["cross_attention_dim"])
    boc = kw["block_out_channels"]

    # the reference ctors ignore their config for the trunk (super().__init__() without arguments) and hard-code the
    # 320/640/1280 adapter widths: build the same objects at the reduced size in two steps, as make_golden.py does
    class ReducedLdmk(LD.FlowControlNet):
        def __init__(self, **k):
            LD.ControlNetSDVModel.__init__(self, **k)
            self.flow_encoder = LD.FlowControlNetFirstFrameEncoder(c_in=boc[0], channels=list(boc[:3]))
            self.controlnet_cond_embedding = LD.FlowControlNetConditioningEmbeddingSVD(
                conditioning_embedding_channels=boc[0], block_out_channels=(16, 32, 96, 256), conditioning_channels=3)
            self.controlnet_ldmk_embedding = LD.FlowControlNetConditioningEmbeddingSVD(
                conditioning_embedding_channels=boc[0], block_out_channels=(16, 32, 64, 128), conditioning_channels=3)
            ch = {"8": boc[0], "16": boc[0], "32": boc[1], "64": boc[2]}
            self.zero_outs = nn.ModuleDict({k2: nn.Conv2d(c, c, 1) for k2, c in ch.items()})
            self.occlusions = nn.ModuleDict({k2: LD.ForegroundMatting(c) for k2, c in ch.items()})

    class ReducedTraj(TR.FlowControlNet):
        def __init__(self, **k):
            TR.ControlNetSDVModel.__init__(self, **k)
            self.flow_encoder = TR.FlowControlNetFirstFrameEncoder(c_in=boc[0], channels=list(boc[:3]))
            self.controlnet_cond_embedding = TR.FlowControlNetConditioningEmbeddingSVD(
                conditioning_embedding_channels=boc[0], block_out_channels=(16, 32, 96, 256), conditioning_channels=3)

    sd_l = {k: t.float() for k, t in schema.synthetic_state_dict(schema.ldmk_controlnet_schema(LDMK_CN), seed=11).items()}
    sd_t = {k: t.float() for k, t in schema.synthetic_state_dict(schema.controlnet_schema(LDMK_CN), seed=12).items()}
    sd_u = {k: t.float() for k, t in schema.synthetic_state_dict(schema.unet_schema(LDMK_UNET), seed=10).items()}
    face = ReducedLdmk(**kw)
    face.load_state_dict(sd_l)
    drag = ReducedTraj(**kw)
    drag.load_state_dict(sd_t)
    un = UNetSpatioTemporalConditionControlNetModel(block_out_channels=boc, num_attention_heads=LDMK_UNET["num_attention_heads"],
                                                    cross_attention_dim=LDMK_UNET["cross_attention_dim"])
    un.load_state_dict(sd_u)
    with torch.device("meta"):
        full = LD.FlowControlNet()
    inv = dict(ldmk_full={k: tuple(v.shape) for k, v in full.state_dict().items()},
               ldmk_reduced={k: tuple(v.shape) for k, v in face.state_dict().items()})
    return MG, face.eval(), drag.eval(), un.eval(), inv


class VaeStub(nn.Module):
    def __init__(self, MG):
        super().__init__()
        self.config = MG._Cfg(block_out_channels=(128, 256, 512, 512), force_upcast=True, scaling_factor=0.18215)
        self.dtype = torch.float32
        self.captured = None

    def encode(self, image):
        g = torch.Generator().manual_seed(5)
        z = torch.randn(image.shape[0], 4, image.shape[2] // 8, image.shape[3] // 8, generator=g) / 0.18215
        self.captured = z
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: z))

    def forward(self, sample, num_frames=1):
        raise NotImplementedError


class ClipStub(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.p = nn.Parameter(torch.zeros(1))
        self.dim = dim

    def forward(self, image):
        g = torch.Generator().manual_seed(6)
        return types.SimpleNamespace(image_embeds=torch.randn(image.shape[0], self.dim, generator=g))


def main(tree):
    from helpers import LDMK_CN, synthetic_inputs, synthetic_landmarks
    from oracle.scheduler import SVD_XT_SCHEDULER
    MG, face, drag, un, inv = build_models(tree)
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    out = dict(inventory=inv)
    H = W = 128
    cross = LDMK_CN["cross_attention_dim"]
    if tree == "hybrid":
        import pipeline.pipeline as P
        T = 3
        inp = synthetic_inputs(T, H, W, cross_dim=cross, seed=43)
        lm = synthetic_landmarks(T, H, W, seed=44)
        drag_flow = synthetic_inputs(T, H, W, cross_dim=cross, seed=45)["flow"] * 0.5
        mask = torch.zeros(1, 1, H, W)
        mask[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
        # adapter forward alone
        sigma = 3.0
        xin = torch.cat([torch.cat([inp["latents"] * 5.0] * 2) / (sigma ** 2 + 1) ** 0.5,
                         inp["image_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)], dim=2)
        ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
        with torch.no_grad():
            dr, mr, _, om = face(xin, torch.tensor(0.8), inp["image_embeddings"], ids,
                                 controlnet_cond=torch.cat([inp["cond"]] * 2), controlnet_flow=torch.cat([inp["flow"]] * 2),
                                 landmarks=torch.cat([lm] * 2), return_dict=False, conditioning_scale=0.9)
        out["ldmk_forward"] = dict(xin=xin, down=dr, mid=mr, occlusion_masks=om, T=T, H=H, W=W, timestep=0.8,
                                   conditioning_scale=0.9)
        vae, clip = VaeStub(MG), ClipStub(cross)
        pipe = P.FlowControlNetPipeline(vae=vae, image_encoder=clip, unet=un, face_controlnet=face, drag_controlnet=drag,
                                        scheduler=EulerDiscreteScheduler(**SVD_XT_SCHEDULER), feature_extractor=None)
        image = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(7)) * 2 - 1
        with torch.no_grad():
            res = pipe(image, inp["cond"], controlnet_flow=inp["flow"], landmarks=lm, drag_flow=drag_flow, mask=mask,
                       height=H, width=W, num_frames=T, num_inference_steps=2, latents=inp["latents"].clone(),
                       output_type="latent", generator=torch.Generator().manual_seed(8), ctrl_scale_traj=0.8,
                       ctrl_scale_ldmk=1.1)
            emb = clip(image).image_embeds.unsqueeze(1)
        fr = res.frames if hasattr(res, "frames") else res[0]
        out["hybrid_pipeline"] = dict(latents_in=inp["latents"], image_latents=vae.captured, image_embeddings=emb,
                                      cond=inp["cond"], flow=inp["flow"], landmarks=lm, drag_flow=drag_flow, mask=mask,
                                      final_latents=fr, T=T, H=H, W=W, steps=2, ctrl_scale_traj=0.8, ctrl_scale_ldmk=1.1)
    else:
        import pipeline.svdxt_pipeline_ctrlnet_loop as P
        N, win, stride = 6, 4, 2
        inp = synthetic_inputs(N, H, W, cross_dim=cross, seed=46)
        lm = synthetic_landmarks(N, H, W, seed=47)
        vae, clip = VaeStub(MG), ClipStub(cross)
        pipe = P.FlowControlNetPipeline(vae=vae, image_encoder=clip, unet=un, controlnet=face,
                                        scheduler=EulerDiscreteScheduler(**SVD_XT_SCHEDULER), feature_extractor=None)
        image = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(7)) * 2 - 1
        with torch.no_grad():
            res = pipe(image, inp["cond"], controlnet_flow=inp["flow"], landmarks=lm, window_size=win, stride=stride,
                       height=H, width=W, num_frames=N, num_inference_steps=2, latents=inp["latents"].clone(),
                       output_type="latent", generator=torch.Generator().manual_seed(8))
            emb = clip(image).image_embeds.unsqueeze(1)
        fr = res.frames if hasattr(res, "frames") else res[0]
        out["keypoint_pipeline"] = dict(latents_in=inp["latents"], image_latents=vae.captured, image_embeddings=emb,
                                        cond=inp["cond"], flow=inp["flow"], landmarks=lm, final_latents=fr, N=N, H=H, W=W,
                                        window_size=win, stride=stride, steps=2)
    path = os.path.join(HERE, f"reference_golden_{tree}.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(sys.argv[1])
    else:
        for t in TREES:
            subprocess.run([sys.executable, os.path.abspath(__file__), t], check=True)
