"""Shared test fixtures: seeded synthetic weights (same fp16-rounded values on both sides) and synthetic
inputs of SURVEY.md 8(d) at reduced sizes the CPU oracle finishes in seconds."""
import math

import torch

from mofa_video_amd import schema

TINY = dict(block_out_channels=(64, 128, 256, 256), num_attention_heads=(1, 2, 4, 4), cross_attention_dim=128)
# the reference ControlNet trunk uses heads (5,10,10,20) -> head dim 128 at level 2; mirrored here (256 / 2)
TINY_CN = dict(TINY, num_attention_heads=(1, 2, 2, 4))
TINY_VAE = dict(block_out_channels=(64, 64, 128, 128))


# reduced configs for the landmark adapter / Hybrid / Keypoint cases: level 0 keeps 320 channels because the reference
# forward tests ``sample.shape[1] == 320`` literally (ldmk_ctrlnet.py:501); heads mirror (5,10,10,20) / (5,10,20,20)
LDMK_CN = dict(block_out_channels=(320, 128, 256, 256), num_attention_heads=(5, 2, 2, 4), cross_attention_dim=128)
LDMK_UNET = dict(block_out_channels=(320, 128, 256, 256), num_attention_heads=(5, 2, 4, 4), cross_attention_dim=128)


def synthetic_landmarks(T, H, W, seed=44):
    """pose images: sparse binary polylines-like pixels in [0,1], [1,T,3,H,W]"""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(1, T, 3, H, W, generator=g) < 0.02).float()


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def oracle_models(cfg=None, seed=0, vae_cfg=None, cn_cfg=None):
    """returns (oracle_unet, oracle_controlnet, oracle_vae, sd_unet, sd_ctrl, sd_vae) with fp16-valued weights"""
    from oracle.controlnet import FlowControlNet
    from oracle.unet import UNetSpatioTemporalConditionControlNetModel
    from oracle.vae import AutoencoderKLTemporalDecoder
    kw = cfg or {}
    ckw = cn_cfg if cn_cfg is not None else kw
    sdu = schema.synthetic_state_dict(schema.unet_schema(cfg), seed=seed)
    sdc = schema.synthetic_state_dict(schema.controlnet_schema(cn_cfg if cn_cfg is not None else cfg), seed=seed + 1)
    vkw = vae_cfg or {}
    sdv = schema.synthetic_state_dict(schema.vae_decoder_schema(**vkw), seed=seed + 2)
    u = UNetSpatioTemporalConditionControlNetModel(**kw)
    c = FlowControlNet(**ckw)
    v = AutoencoderKLTemporalDecoder(**vkw)
    u.load_state_dict({k: t.float() for k, t in sdu.items()})
    c.load_state_dict({k: t.float() for k, t in sdc.items()})
    v.load_state_dict({k: t.float() for k, t in sdv.items()})
    return u.eval(), c.eval(), v.eval(), sdu, sdc, sdv


def synthetic_inputs(T, H, W, cross_dim=1024, seed=42):
    """SURVEY 8(d) config-2 style inputs: one Gaussian-bump trajectory flow growing linearly over frames."""
    g = torch.Generator().manual_seed(seed)
    h, w = H // 8, W // 8
    latents = torch.randn(1, T, 4, h, w, generator=g)
    image_latents = torch.randn(1, 4, h, w, generator=g) / 0.18215
    image_embeddings = torch.randn(1, 1, cross_dim, generator=g)
    cond = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    ys = torch.arange(H, dtype=torch.float32).view(H, 1)
    xs = torch.arange(W, dtype=torch.float32).view(1, W)
    sig = 0.15 * min(H, W)
    bump = torch.exp(-((xs - W / 2) ** 2 + (ys - H / 2) ** 2) / (2 * sig * sig))
    peak = torch.tensor([64.0 * W / 1024, 32.0 * H / 576])
    flow = torch.zeros(1, T - 1, 2, H, W)
    for i in range(T - 1):
        f = (i + 1) / (T - 1)
        flow[0, i, 0] = bump * peak[0] * f
        flow[0, i, 1] = bump * peak[1] * f
    il2 = torch.cat([torch.zeros_like(image_latents), image_latents])
    emb2 = torch.cat([torch.zeros_like(image_embeddings), image_embeddings])
    return dict(latents=latents, image_latents=il2, image_embeddings=emb2, cond=cond, flow=flow)
