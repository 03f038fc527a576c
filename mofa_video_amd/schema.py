"""Checkpoint schema of the hot-path models: ``{state_dict key: shape}`` in the diffusers==0.24.0 layout the
reference checkpoints use (SURVEY.md 8b "state_dict keys that must load unchanged"), plus a seeded
synthetic-weight generator (no checkpoints exist offline).  Host-side only.
"""
import math

import torch

from .unet import DEFAULT_CONFIG


def _lin(d, p, n, k, bias=True):
    d[p + ".weight"] = (n, k)
    if bias:
        d[p + ".bias"] = (n,)


def _norm(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)


def _conv(d, p, n, c, *k):
    d[p + ".weight"] = (n, c) + tuple(k)
    d[p + ".bias"] = (n,)


def _resblock(d, p, cin, cout, temb):
    sp, tp = p + ".spatial_res_block", p + ".temporal_res_block"
    _norm(d, sp + ".norm1", cin)
    _conv(d, sp + ".conv1", cout, cin, 3, 3)
    if temb:
        _lin(d, sp + ".time_emb_proj", cout, temb)
    _norm(d, sp + ".norm2", cout)
    _conv(d, sp + ".conv2", cout, cout, 3, 3)
    if cin != cout:
        _conv(d, sp + ".conv_shortcut", cout, cin, 1, 1)
    _norm(d, tp + ".norm1", cout)
    _conv(d, tp + ".conv1", cout, cout, 3, 1, 1)
    if temb:
        _lin(d, tp + ".time_emb_proj", cout, temb)
    _norm(d, tp + ".norm2", cout)
    _conv(d, tp + ".conv2", cout, cout, 3, 1, 1)
    d[p + ".time_mixer.mix_factor"] = (1,)


def _attn(d, p, c, kv, bias=False):
    _lin(d, p + ".to_q", c, c, bias)
    _lin(d, p + ".to_k", c, kv, bias)
    _lin(d, p + ".to_v", c, kv, bias)
    _lin(d, p + ".to_out.0", c, c, True)


def _ff(d, p, c):
    _lin(d, p + ".net.0.proj", 8 * c, c)
    _lin(d, p + ".net.2", c, 4 * c)


def _transformer(d, p, c, cross):
    _norm(d, p + ".norm", c)
    _lin(d, p + ".proj_in", c, c)
    b = p + ".transformer_blocks.0"
    _norm(d, b + ".norm1", c); _attn(d, b + ".attn1", c, c)
    _norm(d, b + ".norm2", c); _attn(d, b + ".attn2", c, cross)
    _norm(d, b + ".norm3", c); _ff(d, b + ".ff", c)
    t = p + ".temporal_transformer_blocks.0"
    _norm(d, t + ".norm_in", c); _ff(d, t + ".ff_in", c)
    _norm(d, t + ".norm1", c); _attn(d, t + ".attn1", c, c)
    _norm(d, t + ".norm2", c); _attn(d, t + ".attn2", c, cross)
    _norm(d, t + ".norm3", c); _ff(d, t + ".ff", c)
    _lin(d, p + ".time_pos_embed.linear_1", 4 * c, c)
    _lin(d, p + ".time_pos_embed.linear_2", c, 4 * c)
    d[p + ".time_mixer.mix_factor"] = (1,)
    _lin(d, p + ".proj_out", c, c)


def _trunk(d, cfg):
    """conv_in, embeddings, down blocks, mid block -- shared by UNet and ControlNet."""
    boc = tuple(cfg["block_out_channels"])
    temb = boc[0] * 4
    cross = cfg["cross_attention_dim"]
    lpb = cfg["layers_per_block"]
    _conv(d, "conv_in", boc[0], cfg["in_channels"], 3, 3)
    _lin(d, "time_embedding.linear_1", temb, boc[0]); _lin(d, "time_embedding.linear_2", temb, temb)
    _lin(d, "add_embedding.linear_1", temb, cfg["projection_class_embeddings_input_dim"])
    _lin(d, "add_embedding.linear_2", temb, temb)
    out = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        cin, out = out, boc[i]
        for j in range(lpb):
            _resblock(d, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
            if t.startswith("CrossAttn"):
                _transformer(d, f"down_blocks.{i}.attentions.{j}", out, cross)
        if i != len(boc) - 1:
            _conv(d, f"down_blocks.{i}.downsamplers.0.conv", out, out, 3, 3)
    _resblock(d, "mid_block.resnets.0", boc[-1], boc[-1], temb)
    _transformer(d, "mid_block.attentions.0", boc[-1], cross)
    _resblock(d, "mid_block.resnets.1", boc[-1], boc[-1], temb)
    return temb, cross


def unet_schema(config=None):
    cfg = dict(DEFAULT_CONFIG); cfg.update(config or {})
    d = {}
    temb, cross = _trunk(d, cfg)
    boc = tuple(cfg["block_out_channels"])
    n = len(boc)
    rc = list(reversed(boc))
    nl = cfg["layers_per_block"] + 1
    out = rc[0]
    for i, t in enumerate(cfg["up_block_types"]):
        prev, out = out, rc[i]
        inc = rc[min(i + 1, n - 1)]
        for j in range(nl):
            skip = inc if j == nl - 1 else out
            rin = prev if j == 0 else out
            _resblock(d, f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
            if t.startswith("CrossAttn"):
                _transformer(d, f"up_blocks.{i}.attentions.{j}", out, cross)
        if i != n - 1:
            _conv(d, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3, 3)
    _norm(d, "conv_norm_out", boc[0])
    _conv(d, "conv_out", cfg["out_channels"], boc[0], 3, 3)
    return d


def controlnet_schema(config=None, conditioning_embedding_out_channels=(16, 32, 96, 256)):
    cfg = dict(DEFAULT_CONFIG); cfg.update(config or {})
    d = {}
    _trunk(d, cfg)
    boc = tuple(cfg["block_out_channels"])
    k = 0
    _conv(d, f"controlnet_down_blocks.{k}", boc[0], boc[0], 1, 1); k += 1
    for i in range(len(boc)):
        for _ in range(cfg["layers_per_block"]):
            _conv(d, f"controlnet_down_blocks.{k}", boc[i], boc[i], 1, 1); k += 1
        if i != len(boc) - 1:
            _conv(d, f"controlnet_down_blocks.{k}", boc[i], boc[i], 1, 1); k += 1
    _conv(d, "controlnet_mid_block", boc[-1], boc[-1], 1, 1)
    ce = conditioning_embedding_out_channels
    _conv(d, "controlnet_cond_embedding.conv_in", ce[0], 3, 3, 3)
    for i in range(len(ce) - 1):
        _conv(d, f"controlnet_cond_embedding.blocks.{2 * i}", ce[i], ce[i], 3, 3)
        _conv(d, f"controlnet_cond_embedding.blocks.{2 * i + 1}", ce[i + 1], ce[i], 3, 3)
    _conv(d, "controlnet_cond_embedding.conv_out", boc[0], ce[-1], 3, 3)
    # FlowControlNetFirstFrameEncoder(c_in=320, channels=[320, 640, 1280]) -- svdxt_...norefine.py:130-146;
    # expressed through block_out_channels so reduced test configs stay consistent with the trunk
    cin = boc[0]
    for i, ch in enumerate(boc[:3]):
        _conv(d, f"flow_encoder.encoders.{i}.conv_in", ch, cin, 3, 3)
        _conv(d, f"flow_encoder.zeroconvs.{i}", ch, ch, 1, 1)
        cin = ch
    return d


def _matting(d, p, c, num_blocks=3, block_expansion=64, max_features=512):
    """ForegroundMatting (MOFA-Video-Hybrid/models/occlusion/hourglass.py:227-245): hourglass on 2c+2 channels"""
    cin = 2 * c + 2
    for i in range(num_blocks):
        a = cin if i == 0 else min(max_features, block_expansion * (2 ** i))
        b = min(max_features, block_expansion * (2 ** (i + 1)))
        _conv(d, f"{p}.hourglass.encoder.down_blocks.{i}.conv", b, a, 3, 3)
    for j, i in enumerate(range(num_blocks)[::-1]):
        a = (1 if i == num_blocks - 1 else 2) * min(max_features, block_expansion * (2 ** (i + 1)))
        b = min(max_features, block_expansion * (2 ** i))
        _conv(d, f"{p}.hourglass.decoder.up_blocks.{j}.conv", b, a, 3, 3)
    _conv(d, f"{p}.matting_mask", 1, block_expansion, 7, 7)
    _conv(d, f"{p}.matting", c, block_expansion, 7, 7)


def ldmk_controlnet_schema(config=None):
    """landmark MOFA-Adapter (MOFA-Video-Hybrid/models/ldmk_ctrlnet.py:191-254): trajectory adapter without the
    flow-encoder zero convs + landmark embedding + per-scale zero_outs / ForegroundMatting"""
    cfg = dict(DEFAULT_CONFIG); cfg.update(config or {})
    d = controlnet_schema(config)
    for k in [k for k in d if k.startswith("flow_encoder.zeroconvs.")]:
        del d[k]
    boc = tuple(cfg["block_out_channels"])
    ce = (16, 32, 64, 128)
    _conv(d, "controlnet_ldmk_embedding.conv_in", ce[0], 3, 3, 3)
    for i in range(len(ce) - 1):
        _conv(d, f"controlnet_ldmk_embedding.blocks.{2 * i}", ce[i], ce[i], 3, 3)
        _conv(d, f"controlnet_ldmk_embedding.blocks.{2 * i + 1}", ce[i + 1], ce[i], 3, 3)
    _conv(d, "controlnet_ldmk_embedding.conv_out", boc[0], ce[-1], 3, 3)
    for k, c in (("8", boc[0]), ("16", boc[0]), ("32", boc[1]), ("64", boc[2])):
        _conv(d, f"zero_outs.{k}", c, c, 1, 1)
        _matting(d, f"occlusions.{k}", c)
    return d


def vae_decoder_schema(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3):
    d = {}
    boc = tuple(block_out_channels)
    _conv(d, "decoder.conv_in", boc[-1], latent_channels, 3, 3)
    for i in range(layers_per_block):
        _resblock(d, f"decoder.mid_block.resnets.{i}", boc[-1], boc[-1], None)
    a = "decoder.mid_block.attentions.0"
    _norm(d, a + ".group_norm", boc[-1])
    _attn(d, a, boc[-1], boc[-1], bias=True)
    rc = list(reversed(boc))
    out = rc[0]
    for i in range(len(boc)):
        prev, out = out, rc[i]
        for j in range(layers_per_block + 1):
            _resblock(d, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out, None)
        if i != len(boc) - 1:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out, 3, 3)
    _norm(d, "decoder.conv_norm_out", boc[0])
    _conv(d, "decoder.conv_out", out_channels, boc[0], 3, 3)
    _conv(d, "decoder.time_conv_out", out_channels, out_channels, 3, 1, 1)
    return d


def _resblock2d(d, p, cin, cout):
    _norm(d, p + ".norm1", cin)
    _conv(d, p + ".conv1", cout, cin, 3, 3)
    _norm(d, p + ".norm2", cout)
    _conv(d, p + ".conv2", cout, cout, 3, 3)
    if cin != cout:
        _conv(d, p + ".conv_shortcut", cout, cin, 1, 1)


def vae_encoder_schema(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, in_channels=3):
    """``encoder.*`` + ``quant_conv.*`` of AutoencoderKLTemporalDecoder (diffusers models/vae.py Encoder: DownEncoderBlock2D
    x4, UNetMidBlock2D with one 1-head attention, double_z conv_out); 34 163 664 parameters at the SVD configuration."""
    d = {}
    _conv(d, "encoder.conv_in", block_out_channels[0], in_channels, 3, 3)
    c = block_out_channels[0]
    for i, co in enumerate(block_out_channels):
        for j in range(layers_per_block):
            _resblock2d(d, f"encoder.down_blocks.{i}.resnets.{j}", c if j == 0 else co, co)
        if i != len(block_out_channels) - 1:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3, 3)
        c = co
    for j in range(2):
        _resblock2d(d, f"encoder.mid_block.resnets.{j}", c, c)
    a = "encoder.mid_block.attentions.0"
    _norm(d, a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{a}.{n}", c, c)
    _norm(d, "encoder.conv_norm_out", c)
    _conv(d, "encoder.conv_out", 2 * latent_channels, c, 3, 3)
    _conv(d, "quant_conv", 2 * latent_channels, 2 * latent_channels, 1, 1)
    return d


CLIP_VIT_H = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                  patch_size=14, projection_dim=1024, layer_norm_eps=1e-5)


def clip_vision_schema(config=None):
    """transformers CLIPVisionModelWithProjection ``state_dict`` (the reference's ``image_encoder``, run_gradio.py:98-100;
    SVD ships the OpenCLIP ViT-H/14 tower)."""
    c = dict(CLIP_VIT_H)
    c.update(config or {})
    h, p = c["hidden_size"], c["patch_size"]
    d = {}
    v = "vision_model."
    d[v + "embeddings.class_embedding"] = (h,)
    d[v + "embeddings.patch_embedding.weight"] = (h, 3, p, p)
    d[v + "embeddings.position_embedding.weight"] = ((c["image_size"] // p) ** 2 + 1, h)
    _norm(d, v + "pre_layrnorm", h)                                  # (sic) transformers' spelling
    for i in range(c["num_hidden_layers"]):
        q = f"{v}encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(d, q + "self_attn." + n, h, h)
        _norm(d, q + "layer_norm1", h)
        _lin(d, q + "mlp.fc1", c["intermediate_size"], h)
        _lin(d, q + "mlp.fc2", h, c["intermediate_size"])
        _norm(d, q + "layer_norm2", h)
    _norm(d, v + "post_layernorm", h)
    d["visual_projection.weight"] = (c["projection_dim"], h)
    return d


def _bn(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)
    d[p + ".running_mean"] = (c,)
    d[p + ".running_var"] = (c,)
    d[p + ".num_batches_tracked"] = ()


def cmp_schema():
    """CMP module at the inference configuration (Traj/models/cmp/experiments/semiauto_annot/
    resnet50_vip+mpii_liteflow/config.yaml): dilated ResNet-50 (resnet.py:49-166), shallownet8x (shallownet.py:4-41),
    MotionDecoderSkipLayer (decoder.py:96-188); keys as ``CMP(params).state_dict()`` (modules/cmp.py:6-25)."""
    d = {}
    e = "image_encoder"
    d[f"{e}.conv1.weight"] = (64, 3, 7, 7)
    _bn(d, f"{e}.bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for b in range(blocks):
            q = f"{e}.layer{li}.{b}"
            d[f"{q}.conv1.weight"] = (planes, inplanes, 1, 1)
            _bn(d, f"{q}.bn1", planes)
            d[f"{q}.conv2.weight"] = (planes, planes, 3, 3)
            _bn(d, f"{q}.bn2", planes)
            d[f"{q}.conv3.weight"] = (planes * 4, planes, 1, 1)
            _bn(d, f"{q}.bn3", planes * 4)
            if b == 0:
                d[f"{q}.downsample.0.weight"] = (planes * 4, inplanes, 1, 1)
                _bn(d, f"{q}.downsample.1", planes * 4)
            inplanes = planes * 4
    _conv(d, f"{e}.conv5", 256, 2048, 1, 1)
    f = "flow_encoder.features"
    _conv(d, f"{f}.0", 16, 4, 5, 5)
    _bn(d, f"{f}.1", 16)
    _conv(d, f"{f}.4", 16, 16, 3, 3)
    _bn(d, f"{f}.5", 16)
    g = "flow_decoder"
    for name, first in (("decoder1", 0), ("decoder2", 1), ("decoder4", 1), ("decoder8", 1)):
        cin = 256 + 16
        for j in range(3):
            _conv(d, f"{g}.{name}.{first + 3 * j}", 128, cin, 3, 3)
            _bn(d, f"{g}.{name}.{first + 3 * j + 1}", 128)
            cin = 128
    for name, cin, cout in (("fusion8", 512, 256), ("skipconv4", 256, 128), ("fusion4", 384, 128), ("skipconv2", 64, 32),
                            ("fusion2", 160, 64)):
        _conv(d, f"{g}.{name}.0", cout, cin, 3, 3)
        _bn(d, f"{g}.{name}.1", cout)
    _conv(d, f"{g}.head", 198, 64, 1, 1)
    return d


def synthetic_state_dict(schema, seed=0, device="cpu", dtype=torch.float16, gain=1.0):
    """Seeded random weights in the reference layout (default-init-like scales; zero-initialised reference
    layers are random too, otherwise the adapter would contribute nothing -- SURVEY 8c).  ``gain`` scales the
    matrix / convolution weights (2.0 keeps the ReLU + BatchNorm-eval CMP encoder's activations O(1))."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for k, shape in schema.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shape, dtype=torch.long, device=device)
            continue
        if k.endswith("running_var"):
            sd[k] = (0.75 + 0.5 * torch.rand(shape, generator=g, device=device)).to(dtype)
            continue
        if k.endswith("mix_factor"):
            t = torch.full(shape, 0.5, device=device) + 0.3 * torch.randn(shape, generator=g, device=device)
        elif len(shape) == 1:
            is_norm_w = k.endswith(".weight")
            t = 0.05 * torch.randn(shape, generator=g, device=device)
            if is_norm_w:
                t = t + 1.0
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * (gain / math.sqrt(fan_in))
        sd[k] = t.to(dtype)
    return sd
