"""Oracle (test infrastructure): fp32 restatement of the diffusers==0.24.0 blocks
the reference builds its UNet / ControlNet / VAE from.

The reference reaches these through
  ``get_down_block`` / ``get_up_block`` / ``UNetMidBlockSpatioTemporal``
  (MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:169-232,
   MOFA-Video-Traj/models/controlnet_sdv.py:270-309),
  ``Timesteps`` / ``TimestepEmbedding`` (unet_...controlnet.py:137-143).
diffusers itself is not in /root/reference and not installable here; this is
its published v0.24.0 algorithm (models/resnet.py, models/attention.py,
models/transformer_temporal.py, models/unet_3d_blocks.py, models/embeddings.py)
with identical module / parameter names, so diffusers checkpoints load with
``load_state_dict`` unchanged.  PARITY UNPINNED for these blocks (no reference
tests exist) except for the parameter-count checksums asserted in
tests/test_oracle_structure.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# embeddings (diffusers/models/embeddings.py)
# ----------------------------------------------------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False,
                           downscale_freq_shift=1.0, scale=1.0, max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half_dim, dtype=torch.float32,
                                                    device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels,
                                      flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


# ----------------------------------------------------------------------------
# resnets (diffusers/models/resnet.py)
# ----------------------------------------------------------------------------
class Downsample2D(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels=None, temb_channels=512, groups=32, eps=1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalResnetBlock(nn.Module):
    """Input [B, C, T, H, W]; GroupNorm statistics span T*H*W; Conv3d (3,1,1)."""

    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = nn.GroupNorm(32, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv3d(in_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(32, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))
        assert in_channels == out_channels

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None]  # [B, T, C, 1, 1]
            h = h + t.permute(0, 2, 1, 3, 4)
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class AlphaBlender(nn.Module):
    def __init__(self, alpha, merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        self.merge_strategy = merge_strategy
        self.switch_spatial_to_temporal_mix = switch_spatial_to_temporal_mix
        assert merge_strategy in ("learned", "learned_with_images")
        self.register_parameter("mix_factor", nn.Parameter(torch.Tensor([alpha])))

    def get_alpha(self, image_only_indicator, ndims):
        if self.merge_strategy == "learned":
            return torch.sigmoid(self.mix_factor)
        alpha = torch.where(image_only_indicator.bool(),
                            torch.ones(1, 1, device=image_only_indicator.device),
                            torch.sigmoid(self.mix_factor)[..., None])
        if ndims == 5:
            alpha = alpha[:, None, :, None, None]
        elif ndims == 3:
            alpha = alpha.reshape(-1)[:, None, None]
        return alpha

    def forward(self, x_spatial, x_temporal, image_only_indicator=None):
        alpha = self.get_alpha(image_only_indicator, x_spatial.ndim).to(x_spatial.dtype)
        if self.switch_spatial_to_temporal_mix:
            alpha = 1.0 - alpha
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, temb_channels=512, eps=1e-6, temporal_eps=None,
                 merge_factor=0.5, merge_strategy="learned_with_images", switch_spatial_to_temporal_mix=False):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.spatial_res_block = ResnetBlock2D(in_channels, out_channels, temb_channels, eps=eps)
        self.temporal_res_block = TemporalResnetBlock(out_channels, out_channels, temb_channels,
                                                      eps=temporal_eps if temporal_eps is not None else eps)
        self.time_mixer = AlphaBlender(merge_factor, merge_strategy, switch_spatial_to_temporal_mix)

    def forward(self, hidden_states, temb=None, image_only_indicator=None):
        num_frames = image_only_indicator.shape[-1]
        hidden_states = self.spatial_res_block(hidden_states, temb)
        batch_frames, channels, height, width = hidden_states.shape
        batch_size = batch_frames // num_frames
        hs_mix = hidden_states[None, :].reshape(batch_size, num_frames, channels, height, width).permute(0, 2, 1, 3, 4)
        hidden_states = hs_mix
        if temb is not None:
            temb = temb.reshape(batch_size, num_frames, -1)
        hidden_states = self.temporal_res_block(hidden_states, temb)
        hidden_states = self.time_mixer(x_spatial=hs_mix, x_temporal=hidden_states,
                                        image_only_indicator=image_only_indicator)
        return hidden_states.permute(0, 2, 1, 3, 4).reshape(batch_frames, channels, height, width)


# ----------------------------------------------------------------------------
# attention (diffusers/models/attention.py, attention_processor.py)
# ----------------------------------------------------------------------------
# softmax(q k^T / sqrt(d)) v.  A module-level hook so that tests which run this oracle on the GPU at the full 576x1024
# geometry (S = 9216 keys per frame: the materialised fp32 scores of all 50 frames are 85 GB) can substitute an
# exact batch-chunked evaluation of the same formula (tests/test_fullgeom_gpu.py).
SDPA = F.scaled_dot_product_attention


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False,
                 norm_num_groups=None, eps=1e-5, residual_connection=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.residual_connection = residual_connection
        cross = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.group_norm = (nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross, inner, bias=bias)
        self.to_v = nn.Linear(cross, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        if self.group_norm is not None:
            hidden_states = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
        B, S, inner = q.shape
        d = inner // self.heads
        q = q.view(B, -1, self.heads, d).transpose(1, 2)
        k = k.view(B, -1, self.heads, d).transpose(1, 2)
        v = v.view(B, -1, self.heads, d).transpose(1, 2)
        o = SDPA(q, k, v)
        o = o.transpose(1, 2).reshape(B, -1, inner)
        o = self.to_out[0](o)
        if input_ndim == 4:
            o = o.transpose(-1, -2).reshape(b, c, h, w)
        if self.residual_connection:
            o = o + residual
        return o


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        inner = int(dim * mult)
        dim_out = dim_out if dim_out is not None else dim
        self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim_out)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, cross_attention_dim=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim=cross_attention_dim, heads=num_attention_heads,
                               dim_head=attention_head_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states=None):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, time_mix_inner_dim, num_attention_heads, attention_head_dim, cross_attention_dim=None):
        super().__init__()
        self.is_res = dim == time_mix_inner_dim
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=time_mix_inner_dim)
        self.norm1 = nn.LayerNorm(time_mix_inner_dim)
        self.attn1 = Attention(time_mix_inner_dim, heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm2 = nn.LayerNorm(time_mix_inner_dim)
        self.attn2 = Attention(time_mix_inner_dim, cross_attention_dim=cross_attention_dim,
                               heads=num_attention_heads, dim_head=attention_head_dim)
        self.norm3 = nn.LayerNorm(time_mix_inner_dim)
        self.ff = FeedForward(time_mix_inner_dim)

    def forward(self, hidden_states, num_frames, encoder_hidden_states=None):
        batch_frames, seq_length, channels = hidden_states.shape
        batch_size = batch_frames // num_frames
        hidden_states = hidden_states[None, :].reshape(batch_size, num_frames, seq_length, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3).reshape(batch_size * seq_length, num_frames, channels)
        residual = hidden_states
        hidden_states = self.ff_in(self.norm_in(hidden_states))
        if self.is_res:
            hidden_states = hidden_states + residual
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states) + hidden_states
        ff_out = self.ff(self.norm3(hidden_states))
        hidden_states = ff_out + hidden_states if self.is_res else ff_out
        hidden_states = hidden_states[None, :].reshape(batch_size, seq_length, num_frames, channels)
        hidden_states = hidden_states.permute(0, 2, 1, 3).reshape(batch_size * num_frames, seq_length, channels)
        return hidden_states


class TransformerSpatioTemporalModel(nn.Module):
    """diffusers/models/transformer_temporal.py (v0.24.0).

    ``time_context_hw_major=True`` reproduces v0.24.0's construction of the
    temporal block's cross-attention context:
        ctx_first[None].broadcast_to(h*w, B, 1, D).reshape(h*w*B, 1, D)
    i.e. hw-major rows, while the temporal block's token rows are B-major.
    Row r = b*hw + s therefore receives the context of batch (r mod B).
    (Later diffusers releases build it B-major; set False for that.)
    """

    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=320, num_layers=1,
                 cross_attention_dim=None, time_context_hw_major=True):
        super().__init__()
        inner_dim = num_attention_heads * attention_head_dim
        self.inner_dim = inner_dim
        self.in_channels = in_channels
        self.time_context_hw_major = time_context_hw_major
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList([
            BasicTransformerBlock(inner_dim, num_attention_heads, attention_head_dim, cross_attention_dim)
            for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList([
            TemporalBasicTransformerBlock(inner_dim, inner_dim, num_attention_heads, attention_head_dim,
                                          cross_attention_dim)
            for _ in range(num_layers)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_proj = Timesteps(in_channels, True, 0)
        self.time_mixer = AlphaBlender(alpha=0.5, merge_strategy="learned_with_images")
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def forward(self, hidden_states, encoder_hidden_states=None, image_only_indicator=None):
        batch_frames, _, height, width = hidden_states.shape
        num_frames = image_only_indicator.shape[-1]
        batch_size = batch_frames // num_frames

        time_context = encoder_hidden_states
        first = time_context[None, :].reshape(batch_size, num_frames, -1, time_context.shape[-1])[:, 0]
        if self.time_context_hw_major:
            time_context = first[None, :].broadcast_to(height * width, batch_size, 1, first.shape[-1])
            time_context = time_context.reshape(height * width * batch_size, 1, first.shape[-1])
        else:
            time_context = first[:, None].broadcast_to(batch_size, height * width, 1, first.shape[-1])
            time_context = time_context.reshape(batch_size * height * width, 1, first.shape[-1])

        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch_frames, height * width, inner_dim)
        hidden_states = self.proj_in(hidden_states)

        num_frames_emb = torch.arange(num_frames, device=hidden_states.device).repeat(batch_size, 1).reshape(-1)
        t_emb = self.time_proj(num_frames_emb).to(hidden_states.dtype)
        emb = self.time_pos_embed(t_emb)[:, None, :]

        for block, temporal_block in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            hidden_states = block(hidden_states, encoder_hidden_states=encoder_hidden_states)
            hidden_states_mix = hidden_states + emb
            hidden_states_mix = temporal_block(hidden_states_mix, num_frames=num_frames,
                                               encoder_hidden_states=time_context)
            hidden_states = self.time_mixer(x_spatial=hidden_states, x_temporal=hidden_states_mix,
                                            image_only_indicator=image_only_indicator)

        hidden_states = self.proj_out(hidden_states)
        hidden_states = hidden_states.reshape(batch_frames, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        return hidden_states + residual


# ----------------------------------------------------------------------------
# UNet blocks (diffusers/models/unet_3d_blocks.py)
# ----------------------------------------------------------------------------
class DownBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps=1e-5)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, image_only_indicator=None):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb, image_only_indicator=image_only_indicator)
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class CrossAttnDownBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280, add_downsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(in_channels if i == 0 else out_channels, out_channels, temb_channels, eps=1e-6)
            for i in range(num_layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(num_attention_heads, out_channels // num_attention_heads,
                                           in_channels=out_channels, num_layers=transformer_layers_per_block,
                                           cross_attention_dim=cross_attention_dim)
            for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        output_states = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb, image_only_indicator=image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 image_only_indicator=image_only_indicator)
            output_states = output_states + (hidden_states,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
            output_states = output_states + (hidden_states,)
        return hidden_states, output_states


class UNetMidBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, temb_channels, num_layers=1, transformer_layers_per_block=1,
                 num_attention_heads=1, cross_attention_dim=1280):
        super().__init__()
        resnets = [SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5)]
        attentions = []
        for _ in range(num_layers):
            attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, in_channels // num_attention_heads, in_channels=in_channels,
                num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim))
            resnets.append(SpatioTemporalResBlock(in_channels, in_channels, temb_channels, eps=1e-5))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, image_only_indicator=None):
        hidden_states = self.resnets[0](hidden_states, temb, image_only_indicator=image_only_indicator)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 image_only_indicator=image_only_indicator)
            hidden_states = resnet(hidden_states, temb, image_only_indicator=image_only_indicator)
        return hidden_states


class UpBlockSpatioTemporal(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers=1,
                 resnet_eps=1e-6, add_upsample=True):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(SpatioTemporalResBlock(resnet_in_channels + res_skip_channels, out_channels,
                                                  temb_channels, eps=resnet_eps))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, image_only_indicator=None):
        for resnet in self.resnets:
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator=image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class CrossAttnUpBlockSpatioTemporal(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers=1,
                 transformer_layers_per_block=1, resnet_eps=1e-6, num_attention_heads=1,
                 cross_attention_dim=1280, add_upsample=True):
        super().__init__()
        resnets, attentions = [], []
        for i in range(num_layers):
            res_skip_channels = in_channels if (i == num_layers - 1) else out_channels
            resnet_in_channels = prev_output_channel if i == 0 else out_channels
            resnets.append(SpatioTemporalResBlock(resnet_in_channels + res_skip_channels, out_channels,
                                                  temb_channels, eps=resnet_eps))
            attentions.append(TransformerSpatioTemporalModel(
                num_attention_heads, out_channels // num_attention_heads, in_channels=out_channels,
                num_layers=transformer_layers_per_block, cross_attention_dim=cross_attention_dim))
        self.resnets = nn.ModuleList(resnets)
        self.attentions = nn.ModuleList(attentions)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None,
                image_only_indicator=None):
        for resnet, attn in zip(self.resnets, self.attentions):
            res_hidden_states = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res_hidden_states], dim=1)
            hidden_states = resnet(hidden_states, temb, image_only_indicator=image_only_indicator)
            hidden_states = attn(hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 image_only_indicator=image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample,
                   num_attention_heads=None, cross_attention_dim=None, transformer_layers_per_block=1, **_ignored):
    """diffusers/models/unet_3d_blocks.py:get_down_block (SpatioTemporal branches).
    ``resnet_eps`` / ``resnet_act_fn`` passed by the reference are ignored by these block classes."""
    if down_block_type == "DownBlockSpatioTemporal":
        return DownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, add_downsample)
    if down_block_type == "CrossAttnDownBlockSpatioTemporal":
        return CrossAttnDownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers,
                                                transformer_layers_per_block, num_attention_heads,
                                                cross_attention_dim, add_downsample)
    raise ValueError(down_block_type)


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, num_attention_heads=None, cross_attention_dim=None,
                 transformer_layers_per_block=1, **_ignored):
    if up_block_type == "UpBlockSpatioTemporal":
        return UpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels,
                                     num_layers, add_upsample=add_upsample)
    if up_block_type == "CrossAttnUpBlockSpatioTemporal":
        return CrossAttnUpBlockSpatioTemporal(in_channels, out_channels, prev_output_channel, temb_channels,
                                              num_layers, transformer_layers_per_block,
                                              num_attention_heads=num_attention_heads,
                                              cross_attention_dim=cross_attention_dim, add_upsample=add_upsample)
    raise ValueError(up_block_type)
