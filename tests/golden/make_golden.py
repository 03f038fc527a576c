#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ BY RUNNING THE REFERENCE'S OWN PYTHON CODE in the build
container (it cannot travel to the GPU box; the fixtures do).

What runs from /root/reference (imported in place, nothing copied):
  * utils/scheduling_euler_discrete_karras_fix.py      EulerDiscreteScheduler (set_timesteps / scale / step)
  * models/svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py
        FlowControlNetConditioningEmbeddingSVD, FlowControlNetFirstFrameEncoder, FlowControlNet (ctor + forward)
  * models/controlnet_sdv.py                            ControlNetSDVModel ctor (trunk + zero-conv wiring)
  * models/unet_spatio_temporal_condition_controlnet.py UNetSpatioTemporalConditionControlNetModel (ctor + forward)
  * pipeline/pipeline.py                                FlowControlNetPipeline.__call__ (loop, CFG, time-id quirk,
                                                        chunked decode_latents)
What is substituted, because it is absent or cannot run here:
  * ``diffusers`` (not installed): a stub package whose block classes are this repo's oracle restatement
    (oracle/blocks.py, oracle/vae.py) and whose mixins are minimal -- so these fixtures pin the reference's
    IN-TREE code paths, not the third-party block arithmetic;
  * ``models.softsplat.softsplat`` (CuPy/CUDA only): the reference kernel text compiled for the host
    (oracle/_ref, see oracle/build_ref.py) behind the reference's own 'avg' wrapper arithmetic;
  * ``cupy``, ``torchvision``, ``models.cmp`` (CMP_demo is not on the denoise path): empty stubs.

Usage:  python tests/golden/make_golden.py      (writes tests/golden/*.pt, a few hundred KB)
"""
import contextlib
import enum
import inspect
import logging as pylogging
import os
import sys
import types

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/MOFA-Video-Traj"


# ---------------------------------------------------------------------------------------------------------
# stubs
# ---------------------------------------------------------------------------------------------------------
class _Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_cfg", _Cfg(cfg))
        init(self, *args, **kwargs)
    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self.__dict__["_cfg"]

    def __getattr__(self, name):          # diffusers lets config entries be read as attributes
        d = self.__dict__
        if "_cfg" in d and name in d["_cfg"]:
            return d["_cfg"][name]
        try:
            return super().__getattr__(name)
        except AttributeError:
            raise AttributeError(name)


class BaseOutput(dict):
    def __post_init__(self):
        for k, v in self.__dict__.items():
            self[k] = v


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    from oracle import blocks as B
    from oracle import vae as V
    from oracle.softsplat_ref import softsplat_avg_ref

    from transformers import CLIPImageProcessor, CLIPVisionModelWithProjection  # noqa: F401 (before stubbing torchvision)

    logging = types.SimpleNamespace(get_logger=lambda n: pylogging.getLogger(n))

    class KarrasDiffusionSchedulers(enum.Enum):
        EulerDiscreteScheduler = 1

    class ModelMixin(nn.Module):
        pass

    class DiffusionPipeline:
        def register_modules(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def _execution_device(self):
            return torch.device("cpu")

        @contextlib.contextmanager
        def progress_bar(self, total=None):
            yield types.SimpleNamespace(update=lambda *a: None)

        def maybe_free_model_hooks(self):
            pass

    class VaeImageProcessor:
        def __init__(self, vae_scale_factor=8):
            self.vae_scale_factor = vae_scale_factor

        def preprocess(self, image, height=None, width=None):
            assert torch.is_tensor(image) and tuple(image.shape[-2:]) == (height, width)
            return image

        def postprocess(self, video, output_type):
            return video

    def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
        return torch.randn(shape, generator=generator, dtype=dtype)

    _mod("diffusers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.utils", BaseOutput=BaseOutput, logging=logging)
    _mod("diffusers.utils.torch_utils", randn_tensor=randn_tensor)
    _mod("diffusers.schedulers")
    _mod("diffusers.schedulers.scheduling_utils", KarrasDiffusionSchedulers=KarrasDiffusionSchedulers,
         SchedulerMixin=type("SchedulerMixin", (), {}))
    _mod("diffusers.loaders", FromOriginalControlnetMixin=type("FromOriginalControlnetMixin", (), {}),
         UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    _mod("diffusers.models.attention_processor", ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=(),
         AttentionProcessor=object, AttnAddedKVProcessor=object, AttnProcessor=object)
    _mod("diffusers.models.embeddings", TextImageProjection=object, TextImageTimeEmbedding=object,
         TextTimeEmbedding=object, TimestepEmbedding=B.TimestepEmbedding, Timesteps=B.Timesteps)
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.unet_3d_blocks", UNetMidBlockSpatioTemporal=B.UNetMidBlockSpatioTemporal,
         get_down_block=B.get_down_block, get_up_block=B.get_up_block)
    _mod("diffusers.models", UNetSpatioTemporalConditionModel=object,
         AutoencoderKLTemporalDecoder=V.AutoencoderKLTemporalDecoder)
    _mod("diffusers.image_processor", VaeImageProcessor=VaeImageProcessor)
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    _mod("cupy")
    _mod("torchvision")
    _mod("torchvision.transforms")
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # the reference's own models/softsplat.py needs CuPy + CUDA; its kernel runs here through oracle/_ref
    sys.path.insert(0, REF)
    import models  # noqa: F401  (reference package)
    _mod("models.softsplat", softsplat=lambda tenIn, tenFlow, tenMetric, strMode: _softsplat(tenIn, tenFlow, tenMetric,
                                                                                             strMode, softsplat_avg_ref))
    _mod("models.cmp")
    _mod("models.cmp.models")
    _mod("models.cmp.utils")


def _softsplat(tenIn, tenFlow, tenMetric, strMode, avg):
    assert strMode == "avg" and tenMetric is None
    return avg(tenIn, tenFlow)


# ---------------------------------------------------------------------------------------------------------
def main():
    from helpers import TINY, TINY_CN, synthetic_inputs
    from mofa_video_amd import schema
    from oracle.scheduler import SVD_XT_SCHEDULER
    install_stubs()
    from utils.scheduling_euler_discrete_karras_fix import EulerDiscreteScheduler
    import models.svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine as A
    from models.unet_spatio_temporal_condition_controlnet import UNetSpatioTemporalConditionControlNetModel
    import pipeline.pipeline as P

    out = {}
    torch.manual_seed(0)

    # 1. scheduler ----------------------------------------------------------------------------------------
    sch = EulerDiscreteScheduler(**SVD_XT_SCHEDULER)
    gs = {}
    for n in (25, 2, 7):
        sch.set_timesteps(n)
        gs[n] = dict(sigmas=sch.sigmas.clone(), timesteps=sch.timesteps.clone(), init_noise_sigma=float(sch.init_noise_sigma))
    sch.set_timesteps(25)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, 4, 8, 8, generator=g) * 100
    v = torch.randn(1, 3, 4, 8, 8, generator=g)
    traj = []
    cur = x
    for t in sch.timesteps[:4]:
        scaled = sch.scale_model_input(cur, t)
        cur = sch.step(v, t, cur).prev_sample
        traj.append(dict(scaled=scaled.clone(), prev=cur.clone()))
    out["scheduler"] = dict(tables=gs, x=x, v=v, traj=traj)

    # 2. adapter CNNs (reference classes, seeded fp16-valued weights from the shared schema) ------------------
    sdc = {k: t.float() for k, t in schema.synthetic_state_dict(schema.controlnet_schema(TINY_CN), seed=1).items()}
    sdu = {k: t.float() for k, t in schema.synthetic_state_dict(schema.unet_schema(TINY), seed=0).items()}
    T, H, W = 3, 128, 128
    inp = synthetic_inputs(T, H, W, cross_dim=TINY["cross_attention_dim"])

    # 3. reference FlowControlNet / UNet built on the oracle blocks ------------------------------------------
    kw = dict(block_out_channels=TINY["block_out_channels"], num_attention_heads=TINY["num_attention_heads"],
              cross_attention_dim=TINY["cross_attention_dim"])
    # NOTE the reference FlowControlNet.__init__ calls super().__init__() WITHOUT arguments
    # (svdxt_..._norefine.py:213): its trunk is always the full-size ControlNetSDVModel default, heads
    # (5,10,10,20).  To run the reference forward at a reduced size we build the same object in two steps:
    # the reference ControlNetSDVModel ctor WITH the reduced config (heads pattern (1,2,2,4) mirrors
    # (5,10,10,20): head dim 128 at level 2), then the reference adapter sub-modules, exactly as
    # FlowControlNet.__init__ (:215-221) does.
    kw_cn = dict(kw, num_attention_heads=TINY_CN["num_attention_heads"])

    class ReducedFlowControlNet(A.FlowControlNet):
        def __init__(self, **k):
            A.ControlNetSDVModel.__init__(self, **k)
            boc = k["block_out_channels"]
            self.flow_encoder = A.FlowControlNetFirstFrameEncoder(c_in=boc[0], channels=list(boc[:3]))
            self.controlnet_cond_embedding = A.FlowControlNetConditioningEmbeddingSVD(
                conditioning_embedding_channels=boc[0], block_out_channels=(16, 32, 96, 256), conditioning_channels=3)

    cn = ReducedFlowControlNet(**kw_cn)
    cn.load_state_dict(sdc)
    cn.eval()
    un = UNetSpatioTemporalConditionControlNetModel(**kw)
    un.load_state_dict(sdu)
    un.eval()
    out["state_dict_keys"] = dict(
        controlnet={k: tuple(v.shape) for k, v in cn.state_dict().items()},
        unet={k: tuple(v.shape) for k, v in un.state_dict().items()})
    # default (SVD-XT) config key/shape inventory through the reference constructors, on the meta device
    with torch.device("meta"):
        cn_full = A.FlowControlNet(num_attention_heads=(5, 10, 20, 20))     # what from_pretrained(SVD config) passes
        un_full = UNetSpatioTemporalConditionControlNetModel(num_attention_heads=(5, 10, 20, 20))
    out["state_dict_keys_full"] = dict(
        controlnet={k: tuple(v.shape) for k, v in cn_full.state_dict().items()},
        unet={k: tuple(v.shape) for k, v in un_full.state_dict().items()})
    out["effective_heads_full"] = dict(
        controlnet=[b.attentions[0].transformer_blocks[0].attn1.heads for b in cn_full.down_blocks[:3]]
        + [cn_full.mid_block.attentions[0].transformer_blocks[0].attn1.heads],
        unet=[b.attentions[0].transformer_blocks[0].attn1.heads for b in un_full.down_blocks[:3]]
        + [un_full.mid_block.attentions[0].transformer_blocks[0].attn1.heads])
    print("effective heads (reference ctors, SVD-XT config passed):", out["effective_heads_full"])

    with torch.no_grad():
        ce = cn.controlnet_cond_embedding(inp["cond"])
        fe = cn.flow_encoder(ce)
        sigma = 3.0
        lat = inp["latents"] * 5.0
        xin = torch.cat([torch.cat([lat] * 2) / (sigma ** 2 + 1) ** 0.5,
                         inp["image_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)], dim=2)
        ids = torch.tensor([[6.0, 128.0, 0.02]] * 2)
        tt = torch.tensor(0.8)
        cond2, flow2 = torch.cat([inp["cond"]] * 2), torch.cat([inp["flow"]] * 2)
        dr, mr, _, _ = cn(xin, tt, inp["image_embeddings"], ids, controlnet_cond=cond2, controlnet_flow=flow2,
                          return_dict=False, conditioning_scale=0.7)
        npred = un(xin, tt, inp["image_embeddings"], down_block_additional_residuals=dr,
                   mid_block_additional_residual=mr, return_dict=False, added_time_ids=ids)[0]
    out["adapter"] = dict(cond_embedding=ce, flow_encoder=fe, xin=xin, down=dr, mid=mr, noise_pred=npred,
                          T=T, H=H, W=W, conditioning_scale=0.7, timestep=0.8)

    # 4. reference pipeline __call__ ---------------------------------------------------------------------------
    from oracle.vae import AutoencoderKLTemporalDecoder
    from helpers import TINY_VAE
    sdv = {k: t.float() for k, t in schema.synthetic_state_dict(schema.vae_decoder_schema(**TINY_VAE), seed=2).items()}

    class VaeStub(nn.Module):
        def __init__(self):
            super().__init__()
            self.inner = AutoencoderKLTemporalDecoder(**TINY_VAE)
            self.inner.load_state_dict(sdv)
            self.config = _Cfg(block_out_channels=TINY_VAE["block_out_channels"], force_upcast=True, scaling_factor=0.18215)
            self.dtype = torch.float32
            self.captured = None

        def encode(self, image):
            gg = torch.Generator().manual_seed(5)
            z = torch.randn(image.shape[0], 4, image.shape[2] // 8, image.shape[3] // 8, generator=gg) / 0.18215
            self.captured = z
            return types.SimpleNamespace(latent_dist=types.SimpleNamespace(mode=lambda: z))

        def decode(self, z, num_frames=1):
            return types.SimpleNamespace(sample=self.inner.decode(z, num_frames=num_frames))

        def forward(self, sample, num_frames=1):
            return self.decode(sample, num_frames)

    class ClipStub(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))

        def forward(self, image):
            gg = torch.Generator().manual_seed(6)
            return types.SimpleNamespace(image_embeds=torch.randn(image.shape[0], TINY["cross_attention_dim"], generator=gg))

    vae, clip = VaeStub(), ClipStub()
    pipe = P.FlowControlNetPipeline(vae=vae, image_encoder=clip, unet=un, controlnet=cn,
                                    scheduler=EulerDiscreteScheduler(**SVD_XT_SCHEDULER), feature_extractor=None)
    image = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(7)) * 2 - 1
    with torch.no_grad():
        res_lat = pipe(image, inp["cond"], inp["flow"], height=H, width=W, num_frames=T, num_inference_steps=2,
                       decode_chunk_size=2, latents=inp["latents"].clone(), output_type="latent",
                       generator=torch.Generator().manual_seed(8), fps=7, motion_bucket_id=127)
        emb = clip(image).image_embeds.unsqueeze(1)
        frames = pipe.decode_latents(res_lat.frames, T, 2)
    out["pipeline"] = dict(latents_in=inp["latents"], image_latents=vae.captured, image_embeddings=emb,
                           cond=inp["cond"], flow=inp["flow"], final_latents=res_lat.frames, frames=frames,
                           T=T, H=H, W=W, steps=2, decode_chunk_size=2)

    torch.save(out, os.path.join(HERE, "reference_golden.pt"))
    sz = os.path.getsize(os.path.join(HERE, "reference_golden.pt"))
    print(f"wrote tests/golden/reference_golden.pt ({sz / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
