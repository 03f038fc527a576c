"""Oracle (test infrastructure): fp32 CPU restatement of
``UNetSpatioTemporalConditionControlNetModel``
(MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:69-245 ctor,
 :356-504 forward).  Module names equal the reference's so its checkpoints load.
"""
import torch
import torch.nn as nn

from .blocks import (Timesteps, TimestepEmbedding, UNetMidBlockSpatioTemporal, get_down_block, get_up_block)

SVD_XT_HEADS = (5, 10, 20, 20)  # SVD-XT unet/config.json; the code default (5,10,10,20) is overridden by it


class UNetSpatioTemporalConditionControlNetModel(nn.Module):
    def __init__(self, in_channels=8, out_channels=4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=SVD_XT_HEADS, num_frames=25):
        super().__init__()
        self.config = dict(in_channels=in_channels, out_channels=out_channels,
                           block_out_channels=tuple(block_out_channels),
                           addition_time_embed_dim=addition_time_embed_dim, num_frames=num_frames,
                           num_attention_heads=tuple(num_attention_heads), cross_attention_dim=cross_attention_dim,
                           layers_per_block=layers_per_block,
                           projection_class_embeddings_input_dim=projection_class_embeddings_input_dim)
        # unet_...controlnet.py:128-143
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        time_embed_dim = block_out_channels[0] * 4
        self.time_proj = Timesteps(block_out_channels[0], True, downscale_freq_shift=0)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, time_embed_dim)

        n = len(down_block_types)
        layers = [layers_per_block] * n if isinstance(layers_per_block, int) else list(layers_per_block)
        tl = [transformer_layers_per_block] * n
        cad = (cross_attention_dim,) * n

        # down (:163-182)
        self.down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        for i, t in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            self.down_blocks.append(get_down_block(
                t, num_layers=layers[i], transformer_layers_per_block=tl[i], in_channels=input_channel,
                out_channels=output_channel, temb_channels=time_embed_dim, add_downsample=(i != n - 1),
                cross_attention_dim=cad[i], num_attention_heads=num_attention_heads[i]))
        # mid (:185-191)
        self.mid_block = UNetMidBlockSpatioTemporal(
            block_out_channels[-1], temb_channels=time_embed_dim, transformer_layers_per_block=tl[-1],
            cross_attention_dim=cad[-1], num_attention_heads=num_attention_heads[-1])
        # up (:196-233)
        self.up_blocks = nn.ModuleList([])
        rc = list(reversed(block_out_channels))
        rh = list(reversed(num_attention_heads))
        rl = list(reversed(layers))
        output_channel = rc[0]
        for i, t in enumerate(up_block_types):
            prev_output_channel, output_channel = output_channel, rc[i]
            input_channel = rc[min(i + 1, n - 1)]
            self.up_blocks.append(get_up_block(
                t, num_layers=rl[i] + 1, transformer_layers_per_block=tl[i], in_channels=input_channel,
                out_channels=output_channel, prev_output_channel=prev_output_channel,
                temb_channels=time_embed_dim, add_upsample=(i != n - 1), cross_attention_dim=cad[i],
                num_attention_heads=rh[i]))
        # out (:236-245)
        self.conv_norm_out = nn.GroupNorm(32, block_out_channels[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def embed_time(self, sample, timestep, added_time_ids):
        """:386-417 -- time + added-time-id embeddings, [batch, 1280]."""
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float64, device=sample.device)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None].to(sample.device)
        batch_size = sample.shape[0]
        timesteps = timesteps.expand(batch_size)
        emb = self.time_embedding(self.time_proj(timesteps).to(sample.dtype))
        time_embeds = self.add_time_proj(added_time_ids.flatten()).reshape((batch_size, -1)).to(emb.dtype)
        return emb + self.add_embedding(time_embeds)

    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True, added_time_ids=None):
        batch_size, num_frames = sample.shape[:2]
        emb = self.embed_time(sample, timestep, added_time_ids)
        sample = sample.flatten(0, 1)                                            # :421
        emb = emb.repeat_interleave(num_frames, dim=0)                           # :424
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)  # :426
        sample = self.conv_in(sample)                                            # :429
        image_only_indicator = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)

        down_block_res_samples = (sample,)
        for blk in self.down_blocks:                                             # :434-459
            if blk.has_cross_attention:
                sample, res = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  image_only_indicator=image_only_indicator)
            else:
                sample, res = blk(sample, temb=emb, image_only_indicator=image_only_indicator)
            down_block_res_samples += res
            # residual quirk: the whole accumulated tuple is re-added inside the loop,
            # zip() truncating to the shorter list
            new = ()
            for r, add in zip(down_block_res_samples, down_block_additional_residuals):
                new = new + (r + add,)
            down_block_res_samples = new

        sample = self.mid_block(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                image_only_indicator=image_only_indicator)      # :463-468
        sample = sample + mid_block_additional_residual                          # :469

        for blk in self.up_blocks:                                               # :473-491
            res = down_block_res_samples[-len(blk.resnets):]
            down_block_res_samples = down_block_res_samples[:-len(blk.resnets)]
            if blk.has_cross_attention:
                sample = blk(sample, res_hidden_states_tuple=res, temb=emb,
                             encoder_hidden_states=encoder_hidden_states,
                             image_only_indicator=image_only_indicator)
            else:
                sample = blk(sample, res_hidden_states_tuple=res, temb=emb,
                             image_only_indicator=image_only_indicator)

        sample = self.conv_out(self.conv_act(self.conv_norm_out(sample)))        # :494-496
        sample = sample.reshape(batch_size, num_frames, *sample.shape[1:])       # :499
        return (sample,)
