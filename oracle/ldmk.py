"""Oracle (test infrastructure): fp32 CPU restatement of the MOFA-Adapter (landmark) of the Hybrid / Keypoint trees.

Follows
  * ``ForegroundMatting`` / ``Hourglass``   MOFA-Video-Hybrid/models/occlusion/hourglass.py:27-114, :227-280
    (3 conv3x3+ReLU "down" blocks without pooling, 3 conv3x3+ReLU "up" blocks with skip concat, two 7x7 heads)
  * ``FlowControlNet`` (landmark)          MOFA-Video-Hybrid/models/ldmk_ctrlnet.py:191-254 (ctor),
    :291-320 (get_warped_frames: softsplat -> ForegroundMatting -> zero_out), :322-574 (forward)
    (MOFA-Video-Keypoint/models/ldmk_ctrlnet.py is the same file)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .controlnet import (CONTROLNET_TRUNK_HEADS, FlowControlNet, FlowControlNetConditioningEmbeddingSVD,
                         FlowControlNetFirstFrameEncoder)
from .softsplat import softsplat


class _ConvReLU(nn.Module):
    """DownBlock2d / UpBlock2d of hourglass.py:27-56 (norm / pooling / interpolation are commented out there)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)

    def forward(self, x):
        return F.relu(self.conv(x))


class HourglassEncoder(nn.Module):
    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        self.down_blocks = nn.ModuleList([
            _ConvReLU(in_features if i == 0 else min(max_features, block_expansion * (2 ** i)),
                      min(max_features, block_expansion * (2 ** (i + 1)))) for i in range(num_blocks)])

    def forward(self, x):
        outs = [x]
        for b in self.down_blocks:
            outs.append(b(outs[-1]))
        return outs[1:]


class HourglassDecoder(nn.Module):
    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        ups = []
        for i in range(num_blocks)[::-1]:
            cin = (1 if i == num_blocks - 1 else 2) * min(max_features, block_expansion * (2 ** (i + 1)))
            ups.append(_ConvReLU(cin, min(max_features, block_expansion * (2 ** i))))
        self.up_blocks = nn.ModuleList(ups)
        self.out_filters = block_expansion

    def forward(self, x):
        new_out = None
        for b in self.up_blocks:
            out = x.pop()
            if new_out is not None:
                out = torch.cat([out, new_out], dim=1)
            new_out = b(out)
        return new_out


class Hourglass(nn.Module):
    def __init__(self, block_expansion, in_features, num_blocks=3, max_features=256):
        super().__init__()
        self.encoder = HourglassEncoder(block_expansion, in_features, num_blocks, max_features)
        self.decoder = HourglassDecoder(block_expansion, in_features, num_blocks, max_features)
        self.out_filters = self.decoder.out_filters

    def forward(self, x):
        return self.decoder(self.encoder(x))


class ForegroundMatting(nn.Module):
    def __init__(self, num_channels, num_blocks=3, block_expansion=64, max_features=512):
        super().__init__()
        self.hourglass = Hourglass(block_expansion, num_channels * 2 + 2, num_blocks, max_features)
        self.matting_mask = nn.Conv2d(self.hourglass.out_filters, 1, 7, padding=3)
        self.matting = nn.Conv2d(self.hourglass.out_filters, num_channels, 7, padding=3)

    def forward(self, reference_image, dense_flow, warped_image):
        h = self.hourglass(torch.cat([reference_image, dense_flow, warped_image], dim=1))
        mask = torch.sigmoid(self.matting_mask(h))
        return warped_image * mask + self.matting(h) * (1 - mask), mask


class LandmarkFlowControlNet(FlowControlNet):
    """ldmk_ctrlnet.py ``FlowControlNet``: trajectory adapter + landmark embedding + per-scale occlusion/matting."""

    def __init__(self, block_out_channels=(320, 640, 1280, 1280), num_attention_heads=CONTROLNET_TRUNK_HEADS, **kw):
        super().__init__(block_out_channels=block_out_channels, num_attention_heads=num_attention_heads, **kw)
        boc = block_out_channels
        # ldmk_ctrlnet.py:152-161 -- first-frame encoder WITHOUT zero convs
        self.flow_encoder = FlowControlNetFirstFrameEncoder(c_in=boc[0], channels=tuple(boc[:3]), use_zeroconv=False)
        self.controlnet_ldmk_embedding = FlowControlNetConditioningEmbeddingSVD(
            conditioning_embedding_channels=boc[0], block_out_channels=(16, 32, 64, 128), conditioning_channels=3)
        ch = {"8": boc[0], "16": boc[0], "32": boc[1], "64": boc[2]}           # :238-254 (320,320,640,1280)
        self.zero_outs = nn.ModuleDict({k: nn.Conv2d(c, c, 1) for k, c in ch.items()})
        self.occlusions = nn.ModuleDict({k: ForegroundMatting(c) for k, c in ch.items()})

    def get_warped_frames(self, first_frame, flows, scale):                      # :291-320
        warped, masks = [], []
        for i in range(flows.shape[1]):
            w = softsplat(tenIn=first_frame.float(), tenFlow=flows[:, i].float(), tenMetric=None, strMode='avg')
            w = w.to(first_frame.dtype)
            w, m = self.occlusions[str(scale)](first_frame, flows[:, i], w)
            w = self.zero_outs[str(scale)](w)
            warped.append(w.unsqueeze(1))
            masks.append(m.unsqueeze(1))
        return torch.cat(warped, dim=1), torch.cat(masks, dim=1)

    def warped_cond_features(self, controlnet_cond, controlnet_flow):
        cond = self.controlnet_cond_embedding(controlnet_cond)
        feats = [cond] + self.flow_encoder(cond)
        fb, fl, fc, fh, fw = controlnet_flow.shape
        scale_flows = {}
        for scale in (8, 16, 32, 64):
            sf = F.interpolate(controlnet_flow.reshape(-1, fc, fh, fw), scale_factor=1 / scale)
            scale_flows[scale] = sf.reshape(fb, fl, fc, fh // scale, fw // scale) / scale
        out, masks = [], []
        for f in feats:
            cb, cc, ch, cw = f.shape
            w, m = self.get_warped_frames(f, scale_flows[fh // ch], fh // ch)
            w = torch.cat([f.unsqueeze(1), w], dim=1)
            out.append(w.reshape(cb * (fl + 1), cc, ch, cw))
            masks.append(m)
        return out, masks

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                controlnet_flow=None, landmarks=None, image_only_indicator=None, return_dict=True, guess_mode=False,
                conditioning_scale=1.0):
        batch_size, num_frames = sample.shape[:2]
        emb = self.embed_time(sample, timestep, added_time_ids)
        sample = sample.flatten(0, 1)
        emb = emb.repeat_interleave(num_frames, dim=0)
        encoder_hidden_states = encoder_hidden_states.repeat_interleave(num_frames, dim=0)
        sample = self.conv_in(sample)
        ldmk = self.controlnet_ldmk_embedding(landmarks.flatten(0, 1))               # :394-397
        scale_landmarks = {ldmk.shape[-2]: ldmk}
        for scale in (2, 4):                                                          # :399-403
            s = F.interpolate(ldmk, scale_factor=1 / scale)
            scale_landmarks[s.shape[-2]] = s
        warped, occlusion_masks = self.warped_cond_features(controlnet_cond, controlnet_flow)
        image_only_indicator = torch.zeros(batch_size, num_frames, dtype=sample.dtype, device=sample.device)
        count, length = 0, len(warped)
        sample = sample + warped[count] + scale_landmarks[sample.shape[-2]]           # :474
        count += 1
        down_block_res_samples = (sample,)
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                sample, res = blk(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                  image_only_indicator=image_only_indicator)
            else:
                sample, res = blk(sample, temb=emb, image_only_indicator=image_only_indicator)
            if sample.shape[1] == self.config["block_out_channels"][0]:               # :501-504 (== 320)
                sample = sample + warped[min(count, length - 1)] + scale_landmarks[sample.shape[-2]]
            else:
                sample = sample + warped[min(count, length - 1)]
            count += 1
            down_block_res_samples += res
        sample = sample + warped[-1]
        sample = self.mid_block(sample, temb=emb, encoder_hidden_states=encoder_hidden_states,
                                image_only_indicator=image_only_indicator)
        outs = ()
        for r, zc in zip(down_block_res_samples, self.controlnet_down_blocks):
            outs = outs + (zc(r),)
        mid = self.controlnet_mid_block(sample)
        outs = [o * conditioning_scale for o in outs]
        mid = mid * conditioning_scale
        return (outs, mid, controlnet_flow, occlusion_masks)
