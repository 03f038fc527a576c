"""Oracle (test infrastructure): fp32 CPU restatement of the temporal VAE
(``AutoencoderKLTemporalDecoder``, diffusers==0.24.0 models/autoencoder_kl_temporal_decoder.py): the *decoder*, reached
by the reference at MOFA-Video-Traj/pipeline/pipeline.py:194-220 (``decode_latents``), and the *encoder* half
(diffusers models/vae.py ``Encoder`` + ``quant_conv`` + ``DiagonalGaussianDistribution.mode``), reached once per clip at
pipeline.py:141-162 / :338-352.
Module names follow diffusers (``decoder.*``, ``encoder.*``, ``quant_conv``) so SVD checkpoints load.
PARITY UNPINNED beyond the parameter-count checksums (decoder 63 579 183, encoder + quant_conv 34 163 664): diffusers is
not in the reference tree nor installed here; the published module graph is restated.
"""
import torch
import torch.nn as nn

import torch.nn.functional as F

from .blocks import Attention, ResnetBlock2D, SpatioTemporalResBlock, Upsample2D


def _res(cin, cout):
    return SpatioTemporalResBlock(cin, cout, temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0,
                                  merge_strategy="learned", switch_spatial_to_temporal_mix=True)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, attention_head_dim=512, num_layers=1):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels)
                                      for i in range(num_layers)])
        self.attentions = nn.ModuleList([Attention(query_dim=in_channels, heads=in_channels // attention_head_dim,
                                                   dim_head=attention_head_dim, eps=1e-6, norm_num_groups=32,
                                                   bias=True, residual_connection=True)])

    def forward(self, hidden_states, image_only_indicator):
        hidden_states = self.resnets[0](hidden_states, image_only_indicator=image_only_indicator)
        for resnet, attn in zip(self.resnets[1:], self.attentions):
            hidden_states = attn(hidden_states)
            hidden_states = resnet(hidden_states, image_only_indicator=image_only_indicator)
        return hidden_states


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=1, add_upsample=True):
        super().__init__()
        self.resnets = nn.ModuleList([_res(in_channels if i == 0 else out_channels, out_channels)
                                      for i in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, image_only_indicator):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, image_only_indicator=image_only_indicator)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(num_layers=layers_per_block, in_channels=block_out_channels[-1],
                                                 out_channels=block_out_channels[-1],
                                                 attention_head_dim=block_out_channels[-1])
        self.up_blocks = nn.ModuleList([])
        rc = list(reversed(block_out_channels))
        output_channel = rc[0]
        for i in range(len(block_out_channels)):
            prev, output_channel = output_channel, rc[i]
            self.up_blocks.append(UpBlockTemporalDecoder(prev, output_channel, num_layers=layers_per_block + 1,
                                                         add_upsample=(i != len(block_out_channels) - 1)))
        self.conv_norm_out = nn.GroupNorm(32, block_out_channels[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, sample, image_only_indicator, num_frames=1):
        sample = self.conv_in(sample)
        sample = self.mid_block(sample, image_only_indicator=image_only_indicator)
        for up in self.up_blocks:
            sample = up(sample, image_only_indicator=image_only_indicator)
        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))
        bf, c, h, w = sample.shape
        b = bf // num_frames
        sample = sample[None, :].reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        sample = self.time_conv_out(sample)
        return sample.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class DownEncoderBlock2D(nn.Module):
    """diffusers unet_2d_blocks.DownEncoderBlock2D: resnets without time embedding, Downsample2D(padding=0)."""

    def __init__(self, in_channels, out_channels, num_layers, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels=None, eps=1e-6) for i in range(num_layers)])
        self.downsamplers = None
        if add_downsample:
            op = nn.Module()
            op.conv = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=0)
            self.downsamplers = nn.ModuleList([op])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))   # resnet.py Downsample2D
        return x


class UNetMidBlock2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb_channels=None, eps=1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(query_dim=channels, heads=1, dim_head=channels, eps=1e-6,
                                                   norm_num_groups=32, bias=True, residual_connection=True)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, in_channels=3, out_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([])
        c = block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            self.down_blocks.append(DownEncoderBlock2D(c, co, layers_per_block, i != len(block_out_channels) - 1))
            c = co
        self.mid_block = UNetMidBlock2D(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_out = nn.Conv2d(c, 2 * out_channels, 3, padding=1)                   # double_z

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class _Posterior:
    """DiagonalGaussianDistribution: only ``mode()`` (= the mean half) is used by the reference (pipeline.py:150)."""

    def __init__(self, moments):
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)

    def mode(self):
        return self.mean


class _EncOut:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKLTemporalDecoder(nn.Module):
    scaling_factor = 0.18215

    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 with_encoder=False):
        super().__init__()
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        if with_encoder:
            self.encoder = Encoder(out_channels, latent_channels, block_out_channels, layers_per_block)
            self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def encode(self, x):
        return _EncOut(_Posterior(self.quant_conv(self.encoder(x))))

    def decode(self, z, num_frames=1):
        batch_size = z.shape[0] // num_frames
        ioi = torch.zeros(batch_size, num_frames, dtype=z.dtype, device=z.device)
        return self.decoder(z, num_frames=num_frames, image_only_indicator=ioi)


def decode_latents(vae, latents, num_frames, decode_chunk_size=14):
    """MOFA-Video-Traj/pipeline/pipeline.py:194-220."""
    latents = latents.flatten(0, 1)
    latents = 1 / vae.scaling_factor * latents
    frames = []
    for i in range(0, latents.shape[0], decode_chunk_size):
        chunk = latents[i:i + decode_chunk_size]
        frames.append(vae.decode(chunk, num_frames=chunk.shape[0]))
    frames = torch.cat(frames, dim=0)
    frames = frames.reshape(-1, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4)
    return frames.float()
