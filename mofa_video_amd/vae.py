"""MI355X host mirror of ``AutoencoderKLTemporalDecoder`` (diffusers==0.24.0
models/autoencoder_kl_temporal_decoder.py): ``decode``, called by the reference at
MOFA-Video-Traj/pipeline/pipeline.py:194-220 (``decode_latents``: /scaling_factor, chunks of
``decode_chunk_size`` frames, each chunk decoded independently), and ``encode(x).latent_dist.mode()``, called once per
clip at pipeline.py:141-162 (built when the state_dict holds ``encoder.*``).  ``state_dict`` keys are the
diffusers ``decoder.*`` / ``encoder.*`` / ``quant_conv.*`` names.
"""
import math

import torch

from . import lib as L
from . import ops
from .blocks import Conv3x3, ConvT3, Ctx, GroupNorm, Linear, SpatioTemporalResBlock, Sub
from .weights import f32, pack_conv3x3


def _res(s):
    return SpatioTemporalResBlock(s, eps=1e-6, temporal_eps=1e-5, switch=True)


class _MidAttention:
    """diffusers Attention(heads=1, dim_head=512, group norm 32 eps 1e-6, bias, residual_connection)."""

    def __init__(self, s, scale=1.0):
        self.s = scale                                             # input / output stored as scale * value (Encoder); 1 in the decoder
        self.norm = GroupNorm(s.sub("group_norm"), 1e-6 * scale * scale)
        self.to_q, self.to_k, self.to_v = Linear(s.sub("to_q")), Linear(s.sub("to_k")), Linear(s.sub("to_v"))
        self.to_out = Linear(s.sub("to_out.0"))
        self.C = self.to_q.w.shape[0]

    def __call__(self, x, nframes, S):
        Cc = self.C
        assert S % 64 == 0 and Cc % 64 == 0
        h = self.norm(x, nframes, S)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        vt = ops.transpose_v(v, nframes, Cc // 64, S)                   # [nframes*C, S] = V^T per frame
        o = torch.empty((nframes * S, Cc), dtype=torch.float16, device=x.device)
        scale = 1.0 / math.sqrt(Cc)
        for f in range(nframes):                                         # scores materialised per frame
            rows = slice(f * S, (f + 1) * S)
            p = ops.igemm(q[rows], k[rows].contiguous(), s_acc=scale)    # [S, S] = Q K^T / sqrt(d)
            ops.softmax_rows_(p)
            ops.igemm(p, vt[f * Cc:(f + 1) * Cc], out=o[rows])           # P V
        return self.to_out(o, r1=x, s1=1.0, s_acc=self.s)


ENC_SCALE = 2.0 ** -5      # the encoder's residual stream and convolution outputs are stored as ENC_SCALE * value (see Encoder)


class _ResnetBlock2D:
    """diffusers ResnetBlock2D without time embedding (VAE encoder): GN-SiLU-conv, GN-SiLU-conv (+ 1x1 shortcut).
    ``s``: the block's input, the output of conv1 and the block's output are stored as s * value (Encoder)."""

    def __init__(self, s, eps=1e-6, scale=1.0):
        self.s = scale
        self.norm1, self.norm2 = GroupNorm(s.sub("norm1"), eps * scale * scale), GroupNorm(s.sub("norm2"), eps * scale * scale)
        self.conv1, self.conv2 = Conv3x3(s.sub("conv1")), Conv3x3(s.sub("conv2"))
        self.shortcut = Linear(s.sub("conv_shortcut")) if s.has("conv_shortcut.weight") else None
        if self.shortcut is not None and self.shortcut.b is not None:
            self.shortcut.b = self.shortcut.b * scale              # its input is already scaled: W (s x) + s b = s (W x + b)

    def __call__(self, x, n, H, W):
        h = self.conv1(self.norm1(x, n, H * W, silu=True), H, W, s_acc=self.s)
        h = self.norm2(h, n, H * W, silu=True)
        xs = self.shortcut(x) if self.shortcut is not None else x
        return self.conv2(h, H, W, r1=xs, s1=1.0, s_acc=self.s)


class Encoder:
    """diffusers models/vae.py Encoder (DownEncoderBlock2D x N, UNetMidBlock2D, GN-SiLU-conv_out) with ``quant_conv``
    folded into ``conv_out`` (both linear, nothing between them) and only the mean rows kept: the reference reads
    ``latent_dist.mode()`` = the first ``latent_channels`` of the moments (pipeline.py:150).

    RANGE.  The reference upcasts the VAE to fp32 for this call (``force_upcast``, pipeline.py:343-352) because trained VAE
    activations leave the fp16 range.  Here every tensor BETWEEN GroupNorms -- the residual stream and each convolution's output --
    is stored as ``scale`` x its value, scale = 2^-5: GroupNorm is invariant under a common factor once its eps is scaled with it
    (GN_{eps s^2}(s x) = GN_eps(x)), convolutions are linear (``s_acc = s`` where the input is a norm's output; the bias times s
    where the input is the scaled stream), and a power of two is exact in floating point.  So fp16 storage holds values up to
    2.1e6 (the stream of a trained SVD VAE stays far below) with unchanged relative precision down to 2e-3 in absolute value;
    accumulation, norm statistics and softmax are fp32 as everywhere.  tests/test_frontend_gpu.py drives an encoder whose stream
    reaches ~3e5 -- inf / NaN without the factor -- against the fp32 oracle."""

    def __init__(self, s, quant, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, scale=None):
        sc = self.scale = ENC_SCALE if scale is None else float(scale)
        self.conv_in = Conv3x3(s.sub("conv_in"))
        self.in_ld = self.conv_in.w.shape[1] // 9
        self.down = []
        n = len(block_out_channels)
        for i in range(n):
            b = s.sub(f"down_blocks.{i}")
            res = [_ResnetBlock2D(b.sub(f"resnets.{j}"), scale=sc) for j in range(layers_per_block)]
            # Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then a stride-2 conv = trailing-only padding
            down = Conv3x3(b.sub("downsamplers.0.conv"), stride=2, pad=L.PAD_TRAILING) if i != n - 1 else None
            if down is not None:
                down.b = down.b * sc                                 # (scaled stream in, scaled stream out)
            self.down.append((res, down))
        m = s.sub("mid_block")
        self.mid_res = [_ResnetBlock2D(m.sub(f"resnets.{j}"), scale=sc) for j in range(2)]
        self.mid_attn = _MidAttention(m.sub("attentions.0"), scale=sc)
        self.conv_norm_out = GroupNorm(s.sub("conv_norm_out"), 1e-6 * sc * sc)
        wc, bc = s.get("conv_out.weight").float(), s.get("conv_out.bias").float()            # [2z, C, 3, 3]
        wq, bq = quant.get("weight").float().flatten(1), quant.get("bias").float()            # [2z, 2z]
        w = torch.einsum("om,mcyx->ocyx", wq, wc)[:latent_channels]
        b = (wq @ bc + bq)[:latent_channels]
        self.out_w, self.out_b = s.dev(pack_conv3x3(w)), s.dev(f32(b))
        self.latent_channels = latent_channels

    def __call__(self, x, n, H, W):
        x = self.conv_in(x, H, W, s_acc=self.scale)
        for res, down in self.down:
            for r in res:
                x = r(x, n, H, W)
            if down is not None:
                x = down(x, H, W)
                H, W = (H - 2) // 2 + 1, (W - 2) // 2 + 1
        x = self.mid_res[0](x, n, H, W)
        x = self.mid_attn(x, n, H * W)
        x = self.mid_res[1](x, n, H, W)
        x = self.conv_norm_out(x, n, H * W, silu=True)
        return ops.igemm(x, self.out_w, self.out_b, geom=ops.conv3x3_geom(H, W)), H, W      # [n*H*W, z] mean (unscaled)


class _Posterior:
    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class TemporalDecoder:
    def __init__(self, s, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        self.conv_in = Conv3x3(s.sub("conv_in"))
        self.in_ld = self.conv_in.w.shape[1] // 9
        m = s.sub("mid_block")
        self.mid_res = [_res(m.sub(f"resnets.{i}")) for i in range(layers_per_block)]
        self.mid_attn = _MidAttention(m.sub("attentions.0"))
        self.up_blocks = []
        n = len(block_out_channels)
        for i in range(n):
            u = s.sub(f"up_blocks.{i}")
            res = [_res(u.sub(f"resnets.{j}")) for j in range(layers_per_block + 1)]
            up = Conv3x3(u.sub("upsamplers.0.conv"), up=2) if i != n - 1 else None
            self.up_blocks.append((res, up))
        self.conv_norm_out = GroupNorm(s.sub("conv_norm_out"), 1e-6)
        self.conv_out = Conv3x3(s.sub("conv_out"), pad_n=True)          # 3 -> 4 output rows
        self.time_conv_out = ConvT3(s.sub("time_conv_out"), pad_n=True)
        self.out_channels = self.conv_out.n_real

    def __call__(self, z_tokens, nframes, H, W):
        c = Ctx(1, nframes)
        x = self.conv_in(z_tokens, H, W)
        x = self.mid_res[0](x, c, H, W)
        x = self.mid_attn(x, nframes, H * W)
        for r in self.mid_res[1:]:
            x = r(x, c, H, W)
        for res, up in self.up_blocks:
            for r in res:
                x = r(x, c, H, W)
            if up is not None:
                x = up(x, H, W)
                H, W = H * 2, W * 2
        x = self.conv_norm_out(x, nframes, H * W, silu=True)
        kin = self.time_conv_out.w.shape[1] // 3                         # channel-padded width (64)
        y = torch.zeros((x.shape[0], kin), dtype=torch.float16, device=x.device)
        self.conv_out(x, H, W, out=y)
        y = self.time_conv_out(y, nframes, H * W)                        # [n*H*W, 4]
        return y, H, W


class AutoencoderKLTemporalDecoder:
    def __init__(self, state_dict, config=None, device="cuda"):
        cfg = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4,
                   scaling_factor=0.18215, force_upcast=True)
        cfg.update(config or {})
        self.config = type("Cfg", (dict,), {"__getattr__": dict.__getitem__})(cfg)
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.decoder = TemporalDecoder(Sub(state_dict, "decoder.", device), tuple(cfg["block_out_channels"]),
                                       cfg["layers_per_block"])
        self.encoder = None
        if "encoder.conv_in.weight" in state_dict:
            self.encoder = Encoder(Sub(state_dict, "encoder.", device), Sub(state_dict, "quant_conv.", device),
                                   tuple(cfg["block_out_channels"]), cfg["layers_per_block"], cfg["latent_channels"],
                                   scale=cfg.get("encoder_range_scale"))

    def encode(self, x):
        """x fp32 [n, 3, H, W] in [-1, 1] -> ``.latent_dist.mode()`` fp32 [n, 4, H/8, W/8] (the reference upcasts the
        VAE to fp32 for this call, pipeline.py:343-352, for range; here fp16 storage of 2^-5-scaled tensors: ``Encoder``)"""
        if self.encoder is None:
            raise ValueError("this AutoencoderKLTemporalDecoder was built from a state_dict without encoder.* weights")
        n, c, H, W = x.shape
        xt = ops.nchw_to_tokens(x.to(self.device, torch.float32).contiguous(), ld=self.encoder.in_ld)
        y, h, w = self.encoder(xt, n, H, W)
        return _EncoderOutput(_Posterior(ops.tokens_to_nchw(y, n, self.encoder.latent_channels, h, w)))

    @classmethod
    def from_module(cls, module, device="cuda"):
        return cls(module.state_dict(), None, device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", variant=None, **unused):
        """``vae/`` of the SVD-XT checkpoint directory (loaded by ``FlowControlNetPipeline.from_pretrained`` in the reference)"""
        from . import checkpoint
        path = checkpoint.resolve_dir(pretrained_model_name_or_path, subfolder)
        keys = ("block_out_channels", "layers_per_block", "latent_channels", "scaling_factor", "force_upcast")
        cfg = {k: v for k, v in checkpoint.load_config(path).items() if k in keys}
        return cls(checkpoint.load_state_dict(path, variant), cfg, device)

    def decode(self, z, num_frames=1, _prescale=1.0):
        """z [n, 4, h, w] (n = batch*num_frames, one temporal group of num_frames) -> fp32 [n, 3, 8h, 8w]"""
        n, Cz, h, w = z.shape
        assert n == num_frames, "one clip chunk per call (reference decodes chunk by chunk, batch 1)"
        zt = ops.nchw_to_tokens(z.to(self.device, torch.float32), ld=self.decoder.in_ld, scale=_prescale)
        y, H, W = self.decoder(zt, n, h, w)
        return ops.tokens_to_nchw(y, n, self.decoder.out_channels, H, W)


def decode_latents(vae, latents, num_frames, decode_chunk_size=14):
    """pipeline.py:194-220.  latents fp32 [1, T, 4, h, w] -> fp32 frames [1, 3, T, H, W]"""
    lat = latents.flatten(0, 1)
    frames = []
    for i in range(0, lat.shape[0], decode_chunk_size):
        chunk = lat[i:i + decode_chunk_size]          # 1/scaling_factor is applied inside the layout kernel
        frames.append(vae.decode(chunk, num_frames=chunk.shape[0], _prescale=1.0 / vae.config.scaling_factor))
    frames = torch.cat(frames, dim=0)
    return frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()
