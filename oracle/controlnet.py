"""Oracle (test infrastructure): fp32 CPU restatement of the MOFA-Adapter (trajectory).

Follows
  * ``ControlNetSDVModel.__init__``  MOFA-Video-Traj/models/controlnet_sdv.py:156-309
    (trunk = UNet encoder + mid, 12 + 1 zero 1x1 convs, :259-300)
  * ``FlowControlNetConditioningEmbeddingSVD``  MOFA-Video-Traj/models/
    svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py:66-101
  * ``FlowControlNetFirstFrameEncoder`` ibid. :130-155
  * ``FlowControlNet.get_warped_frames`` :223-234, ``.forward`` :236-383
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import Timesteps, TimestepEmbedding, UNetMidBlockSpatioTemporal, get_down_block
from .softsplat import softsplat

# FlowControlNet.__init__ calls ``super().__init__()`` with NO arguments (svdxt_..._norefine.py:213): the trunk is
# always ControlNetSDVModel's default architecture, whose num_attention_heads default is (5, 10, 10, 20)
# (controlnet_sdv.py:180) -- not the SVD-XT UNet's (5, 10, 20, 20).  Level 2 therefore runs 10 heads x 128.
CONTROLNET_TRUNK_HEADS = (5, 10, 10, 20)


class FlowControlNetConditioningEmbeddingSVD(nn.Module):
    def __init__(self, conditioning_embedding_channels, conditioning_channels=3,
                 block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(conditioning_channels, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList([])
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(cin, cin, 3, padding=1))
            self.blocks.append(nn.Conv2d(cin, cout, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], conditioning_embedding_channels, 3, padding=1)  # zero-init in ref

    def forward(self, conditioning):
        e = F.silu(self.conv_in(conditioning))
        for b in self.blocks:
            e = F.silu(b(e))
        return self.conv_out(e)


class FlowControlNetFirstFrameEncoderLayer(nn.Module):
    def __init__(self, c_in, c_out, is_downsample=False):
        super().__init__()
        self.conv_in = nn.Conv2d(c_in, c_out, 3, padding=1, stride=2 if is_downsample else 1)

    def forward(self, feature):
        return F.silu(self.conv_in(feature))


class FlowControlNetFirstFrameEncoder(nn.Module):
    def __init__(self, c_in=320, channels=(320, 640, 1280), downsamples=(True, True, True), use_zeroconv=True):
        super().__init__()
        self.encoders = nn.ModuleList([])
        self.zeroconvs = nn.ModuleList([])
        for ch, ds in zip(channels, downsamples):
            self.encoders.append(FlowControlNetFirstFrameEncoderLayer(c_in, ch, is_downsample=ds))
            self.zeroconvs.append(nn.Conv2d(ch, ch, 1) if use_zeroconv else nn.Identity())
            c_in = ch

    def forward(self, first_frame):
        feature, deep = first_frame, []
        for enc, zc in zip(self.encoders, self.zeroconvs):
            feature = enc(feature)
            deep.append(zc(feature))
        return deep


class FlowControlNet(nn.Module):
    def __init__(self, in_channels=8,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=CONTROLNET_TRUNK_HEADS, num_frames=25,
                 conditioning_channels=3, conditioning_embedding_out_channels=(16, 32, 96, 256)):
        super().__init__()
        self.config = dict(in_channels=in_channels, block_out_channels=tuple(block_out_channels),
                           addition_time_embed_dim=addition_time_embed_dim, num_frames=num_frames,
                           num_attention_heads=tuple(num_attention_heads), cross_attention_dim=cross_attention_dim,
                           layers_per_block=layers_per_block,
                           projection_class_embeddings_input_dim=projection_class_embeddings_input_dim)
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], 3, padding=1)
        time_embed_dim = block_out_channels[0] * 4
        self.time_proj = Timesteps(block_out_channels[0], True, downscale_freq_shift=0)
        self.time_embedding = TimestepEmbedding(block_out_channels[0], time_embed_dim)
        self.add_time_proj = Timesteps(addition_time_embed_dim, True, downscale_freq_shift=0)
        self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, time_embed_dim)

        n = len(down_block_types)
        layers = [layers_per_block] * n
        self.down_blocks = nn.ModuleList([])
        self.controlnet_down_blocks = nn.ModuleList([])
        output_channel = block_out_channels[0]
        self.controlnet_down_blocks.append(nn.Conv2d(output_channel, output_channel, 1))       # controlnet_sdv.py:259-262
        for i, t in enumerate(down_block_types):
            input_channel, output_channel = output_channel, block_out_channels[i]
            is_final = i == n - 1
            self.down_blocks.append(get_down_block(
                t, num_layers=layers[i], transformer_layers_per_block=transformer_layers_per_block,
                in_channels=input_channel, out_channels=output_channel, temb_channels=time_embed_dim,
                add_downsample=not is_final, cross_attention_dim=cross_attention_dim,
                num_attention_heads=num_attention_heads[i]))
            for _ in range(layers[i]):                                                         # :285-288
                self.controlnet_down_blocks.append(nn.Conv2d(output_channel, output_channel, 1))
            if not is_final:                                                                   # :290-293
                self.controlnet_down_blocks.append(nn.Conv2d(output_channel, output_channel, 1))
        self.controlnet_mid_block = nn.Conv2d(block_out_channels[-1], block_out_channels[-1], 1)  # :296-300
        self.mid_block = UNetMidBlockSpatioTemporal(
            block_out_channels[-1], temb_channels=time_embed_dim,
            transformer_layers_per_block=transformer_layers_per_block, cross_attention_dim=cross_attention_dim,
            num_attention_heads=num_attention_heads[-1])
        # svdxt_...norefine.py:215-221
        # reference: FlowControlNetFirstFrameEncoder() = (c_in=320, channels=[320,640,1280]); written through
        # block_out_channels so reduced test configs stay consistent (identical at the default config)
        self.flow_encoder = FlowControlNetFirstFrameEncoder(c_in=block_out_channels[0],
                                                            channels=tuple(block_out_channels[:3]))
        self.controlnet_cond_embedding = FlowControlNetConditioningEmbeddingSVD(
            conditioning_embedding_channels=block_out_channels[0],
            block_out_channels=conditioning_embedding_out_channels, conditioning_channels=conditioning_channels)

    # -- timestep-invariant part: svdxt_...norefine.py:297-319 -------------------------
    def get_warped_frames(self, first_frame, flows):
        warped = []
        for i in range(flows.shape[1]):
            w = softsplat(tenIn=first_frame.float(), tenFlow=flows[:, i].float(), tenMetric=None, strMode='avg')
            warped.append(w.to(first_frame.dtype).unsqueeze(1))
        return torch.cat(warped, dim=1)

    def warped_cond_features(self, controlnet_cond, controlnet_flow):
        cond = self.controlnet_cond_embedding(controlnet_cond)                       # :298
        feats = [cond] + self.flow_encoder(cond)                                     # :300
        fb, fl, fc, fh, fw = controlnet_flow.shape
        scale_flows = {}
        for scale in (8, 16, 32, 64):                                                # :302-309
            sf = F.interpolate(controlnet_flow.reshape(-1, fc, fh, fw), scale_factor=1 / scale)
            scale_flows[scale] = sf.reshape(fb, fl, fc, fh // scale, fw // scale) / scale
        out = []
        for f in feats:                                                              # :311-319
            cb, cc, ch, cw = f.shape
            w = self.get_warped_frames(f, scale_flows[fh // ch])
            w = torch.cat([f.unsqueeze(1), w], dim=1)
            out.append(w.reshape(cb * (fl + 1), cc, ch, cw))
        return out

    def embed_time(self, sample, timestep, added_time_ids):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.float64, device=sample.device)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None].to(sample.device)
        batch_size = sample.shape[0]
        timesteps = timesteps.expand(batch_size)
        emb = self.time_embedding(self.time_proj(timesteps).to(sample.dtype))
        te = self.add_time_proj(added_time_ids.flatten()).reshape((batch_size, -1)).to(emb.dtype)
        return emb + self.add_embedding(te)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                controlnet_flow=None, image_only_indicator=None, return_dict=True, guess_mode=False,
                conditioning_scale=1.0):
        batch_size, num_frames = sample.shape[:2]
        emb = self.embed_time(sample, timestep, added_time_ids)                      # :251-282
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(num_frames, dim=0)
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)
        sample = self.conv_in(sample)                                                # :294
        warped = self.warped_cond_features(controlnet_cond, controlnet_flow)
        image_only_indicator = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)

        count, length = 0, len(warped)
        sample = sample + warped[count]                                              # :328
        count += 1
        down_block_res_samples = (sample,)
        for blk in self.down_blocks:                                                 # :333-351
            if blk.has_cross_attention:
                sample, res = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  image_only_indicator=image_only_indicator)
            else:
                sample, res = blk(sample, temb=emb, image_only_indicator=image_only_indicator)
            sample = sample + warped[min(count, length - 1)]
            count += 1
            down_block_res_samples += res
        sample = sample + warped[-1]                                                 # :354
        sample = self.mid_block(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                image_only_indicator=image_only_indicator)          # :357-362
        outs = ()
        for r, zc in zip(down_block_res_samples, self.controlnet_down_blocks):       # :364-370
            outs = outs + (zc(r),)
        mid = self.controlnet_mid_block(sample)                                      # :372
        outs = [o * conditioning_scale for o in outs]                                # :375-376
        mid = mid * conditioning_scale
        return (outs, mid, controlnet_flow, None)
