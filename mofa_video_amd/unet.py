"""MI355X host mirror of ``UNetSpatioTemporalConditionControlNetModel``
(reference: MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:69-245 ctor, :356-504 forward).

Same constructor config, same ``forward`` signature and return convention, same ``state_dict`` keys
(diffusers layout); the arithmetic is libmofa_hip.so.  ``forward_tokens`` is the layout-conversion-free
entry the pipeline uses (token-major fp16 in/out).
"""
import torch

from . import ops
from .blocks import BIG, Conv3x3, Ctx, DownBlock, GroupNorm, MidBlock, Sub, TembBatch, TimeEmbedding, UpBlock, drive

SVD_XT_HEADS = (5, 10, 20, 20)
DEFAULT_CONFIG = dict(in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                      addition_time_embed_dim=256, projection_class_embeddings_input_dim=768, layers_per_block=2,
                      cross_attention_dim=1024, num_attention_heads=SVD_XT_HEADS, num_frames=25,
                      down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                      up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3)


class _Config(dict):
    __getattr__ = dict.__getitem__


def residual_multiplicity(n_skips_after_block, n_residuals):
    """Reference quirk (unet_..._controlnet.py:434-459): inside the down-block loop the whole
    accumulated skip tuple is re-zipped with the residual list after every block, so skip i receives
    residual i once per remaining block.  Returns the multiplicity of each residual."""
    mult = [0] * n_residuals
    for n in n_skips_after_block:
        for i in range(min(n, n_residuals)):
            mult[i] += 1
    return mult


class UNetSpatioTemporalConditionControlNetModel:
    def __init__(self, state_dict, config=None, device="cuda", dtype=torch.float16):
        cfg = dict(DEFAULT_CONFIG)
        cfg.update(config or {})
        self.config = _Config(cfg)
        self.device, self.dtype = torch.device(device), dtype
        s = Sub(state_dict, "", device)
        boc = tuple(cfg["block_out_channels"])
        heads = tuple(cfg["num_attention_heads"])
        n = len(boc)
        lpb = cfg["layers_per_block"]
        self.conv_in = Conv3x3(s.sub("conv_in"))
        self.time = TimeEmbedding(s, boc[0], cfg["addition_time_embed_dim"])
        with TembBatch() as self.temb_batch:          # every time_emb_proj of the network -> one GEMM per step
            self.down_blocks = []
            for i, t in enumerate(cfg["down_block_types"]):
                self.down_blocks.append(DownBlock(s.sub(f"down_blocks.{i}"), lpb, heads[i], cross=t.startswith("CrossAttn"),
                                                  downsample=(i != n - 1)))
            self.mid_block = MidBlock(s.sub("mid_block"), heads[-1])
            rh = list(reversed(heads))
            self.up_blocks = []
            for i, t in enumerate(cfg["up_block_types"]):
                self.up_blocks.append(UpBlock(s.sub(f"up_blocks.{i}"), lpb + 1, rh[i], cross=t.startswith("CrossAttn"),
                                              upsample=(i != n - 1)))
        self.conv_norm_out = GroupNorm(s.sub("conv_norm_out"), 1e-5)
        self.conv_out = Conv3x3(s.sub("conv_out"))
        self.in_ld = self.conv_in.w.shape[1] // 9      # channel-padded input width (64)

    @classmethod
    def from_module(cls, module, device="cuda"):
        """Build from any torch module with the reference parameter names (e.g. a loaded checkpoint)."""
        return cls(module.state_dict(), getattr(module, "config", None), device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", variant=None, **unused):
        """``config.json`` + ``diffusion_pytorch_model.safetensors`` of a checkpoint directory, as
        MOFA-Video-Traj/run_gradio.py:98-104 loads the SVD-XT UNet (``subfolder="unet"``); extra diffusers keyword
        arguments (``low_cpu_mem_usage``, ``torch_dtype`` ...) are accepted and ignored: storage is always fp16"""
        from . import checkpoint
        path = checkpoint.resolve_dir(pretrained_model_name_or_path, subfolder)
        cfg = {k: v for k, v in checkpoint.load_config(path).items() if k in DEFAULT_CONFIG}
        return cls(checkpoint.load_state_dict(path, variant), cfg, device)

    # ------------------------------------------------------------------------------------------------
    def make_ctx(self, timestep, encoder_hidden_states, added_time_ids, B, T, base=None, half=None, par=None):
        """B, T = LOCAL batch / frame counts.  half: global CFG-half index when this rank computes one half only
        (encoder_hidden_states / added_time_ids are then still the global 2-row tensors); par: FrameParallel."""
        c = base if base is not None else Ctx(B, T)
        ts = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        ts = ts.expand(B).contiguous() if ts.numel() == 1 else ts.contiguous()
        ids = added_time_ids.to(self.device, torch.float32)
        if half is not None:
            ids = ids[half:half + B]
        c.temb_act = self.time(ts, ids.contiguous())
        c.temb_all = self.temb_batch.run(c.temb_act)
        if c.ctx16 is None:
            e = encoder_hidden_states.to(self.device, torch.float32)
            e = e.reshape(e.shape[0], -1).contiguous()
            if half is not None:
                c.ctx16_all = ops.cast_f32_to_f16(e)
                c.ctx16 = c.ctx16_all[half:half + B].contiguous()
                c.half = half
            else:
                c.ctx16 = ops.cast_f32_to_f16(e)
        c.par = par
        return c

    def forward_tokens(self, x, c, H, W, down_res, mid_res):
        """x: fp16 [B*T*H*W, in_ld] (channels >= in_channels zero); down_res: 12 token tensors; mid_res: token
        tensor.  Returns fp16 [B*T*H*W, 4] noise prediction (token-major)."""
        return self.decode_tokens(self.encode_tokens(x, c, H, W), c, down_res, mid_res)

    def encode_layers(self, x, c, H, W):
        """conv_in + down blocks + mid block as a layer generator (blocks.run_lockstep); returns what ``encode_tokens`` returns"""
        sample = self.conv_in(x, H, W, stats=True)                   # (the first resnet's norm1 reads it next)
        skips = [sample]
        counts = []
        for blk in self.down_blocks:
            sample, H, W, outs = yield from blk.layers(sample, c, H, W)
            skips += [o[0] for o in outs]
            counts.append(len(skips))
            yield
        sample = yield from self.mid_block.layers(sample, c, H, W)
        return sample, skips, counts, H, W

    def encode_tokens(self, x, c, H, W):
        """conv_in + down blocks + mid block: the part of the forward that does not depend on the adapter's residuals (the
        ControlNet trunk of the same step is independent of it)"""
        return drive(self.encode_layers(x, c, H, W))

    def decode_tokens(self, enc, c, down_res, mid_res):
        sample, skips, counts, H, W = enc
        # residual quirk, applied after the mid block has consumed the last (un-added) skip: skip i + multiplicity x residual i.
        # Nothing is added in place and nothing is copied for the channel concats: the sums are written straight into their
        # columns of the concat buffers by the up blocks (blocks.UpBlock), the encoder's tensors stay as they are.
        mult = residual_multiplicity(counts, len(down_res))
        assert len(skips) == len(down_res)
        lazy = list(zip(skips, down_res, mult))
        from .blocks import concat_target
        cat, tgt = concat_target(sample.shape[0], sample.shape[1], lazy, sample)
        ops.axpby_out(mid_res, sample, 1.0, 1.0, out=tgt)            # mid block output + mid residual -> first concat operand
        sample = cat
        for blk in self.up_blocks:
            sample, H, W = blk(sample, lazy, c, H, W)
        sample = self.conv_norm_out(sample, c.N, H * W, silu=True)
        return self.conv_out(sample, H, W)

    # reference signature ------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict=True, added_time_ids=None):
        B, T, Cin, H, W = sample.shape
        c = self.make_ctx(timestep, encoder_hidden_states, added_time_ids, B, T)
        x = ops.nchw_to_tokens(sample.reshape(B * T, Cin, H, W).to(self.device, torch.float32), ld=self.in_ld)
        res = [ops.nchw_to_tokens(r.to(self.device, torch.float32)) for r in down_block_additional_residuals]
        mid = ops.nchw_to_tokens(mid_block_additional_residual.to(self.device, torch.float32))
        out = self.forward_tokens(x, c, H, W, res, mid)
        out = ops.tokens_to_nchw(out, B * T, self.config.out_channels, H, W).reshape(B, T, -1, H, W).to(sample.dtype)
        if not return_dict:
            return (out,)
        return _Config(sample=out)

    __call__ = forward
