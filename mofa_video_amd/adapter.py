"""MI355X host mirror of the MOFA-Adapter (trajectory) ``FlowControlNet``.

Reference: MOFA-Video-Traj/models/svdxt_featureflow_forward_controlnet_s2d_fixcmp_norefine.py
  (FlowControlNetConditioningEmbeddingSVD :66-101, FlowControlNetFirstFrameEncoder :130-155,
   FlowControlNet.get_warped_frames :223-234, FlowControlNet.forward :236-383)
and MOFA-Video-Traj/models/controlnet_sdv.py:156-309 (trunk + 13 zero convs).

``prepare_condition`` computes everything that depends only on (controlnet_cond, controlnet_flow): the
condition embedding CNN, the first-frame pyramid, the 4 flow pyramids and all (T-1) x 4 forward-splat warps.
The reference recomputes them every denoise step and for both CFG halves although they are timestep- and
half-invariant; hoisting is results-identical.
"""
import torch

from . import lib as L
from . import ops
from .blocks import BIG, Conv3x3, Ctx, DownBlock, Linear, MidBlock, Sub, TembBatch, TimeEmbedding, drive
from .unet import DEFAULT_CONFIG, _Config


class _CondEmbedding:
    """FlowControlNetConditioningEmbeddingSVD: 3->16->16->32->32->96->96->256->320, SiLU between."""

    def __init__(self, s):
        self.convs = [(Conv3x3(s.sub("conv_in")), True)]
        i = 0
        while s.has(f"blocks.{i}.weight"):
            self.convs.append((Conv3x3(s.sub(f"blocks.{i}"), stride=2 if i % 2 == 1 else 1), True))
            i += 1
        self.convs.append((Conv3x3(s.sub("conv_out")), False))
        self.in_ld = self.convs[0][0].w.shape[1] // 9

    def __call__(self, x, H, W):
        for conv, silu in self.convs:
            n = conv.n_real
            g = ops.conv3x3_geom(H, W, conv.stride, 1)
            ld = (n + 63) // 64 * 64
            out = torch.zeros((x.shape[0] // (H * W) * g.Hout * g.Wout, ld), dtype=torch.float16, device=x.device)
            conv(x, H, W, act=L.ACT_SILU if silu else L.ACT_NONE, out=out)
            x, H, W = out, g.Hout, g.Wout
        return x, H, W


class FlowControlNet:
    # The reference FlowControlNet.__init__ calls ``super().__init__()`` WITHOUT arguments
    # (svdxt_..._norefine.py:213), so whatever config.json says, its trunk is ControlNetSDVModel's default
    # architecture (controlnet_sdv.py:158-183): in particular num_attention_heads = (5, 10, 10, 20), i.e. the
    # 1280-channel level-2 transformers run 10 heads of dim 128 (the SVD-XT UNet runs (5, 10, 20, 20)).
    TRUNK_HEADS = (5, 10, 10, 20)

    def __init__(self, state_dict, config=None, device="cuda", dtype=torch.float16):
        cfg = dict(DEFAULT_CONFIG)
        cfg["num_attention_heads"] = self.TRUNK_HEADS
        cfg.update(config or {})
        self.config = _Config(cfg)
        self.device, self.dtype = torch.device(device), dtype
        s = Sub(state_dict, "", device)
        boc = tuple(cfg["block_out_channels"])
        heads = tuple(cfg["num_attention_heads"])
        n = len(boc)
        lpb = cfg["layers_per_block"]
        self.conv_in = Conv3x3(s.sub("conv_in"))
        self.in_ld = self.conv_in.w.shape[1] // 9
        self.time = TimeEmbedding(s, boc[0], cfg["addition_time_embed_dim"])
        with TembBatch() as self.temb_batch:          # every time_emb_proj of the trunk -> one GEMM per step
            self.down_blocks = []
            for i, t in enumerate(cfg["down_block_types"]):
                self.down_blocks.append(DownBlock(s.sub(f"down_blocks.{i}"), lpb, heads[i], cross=t.startswith("CrossAttn"),
                                                  downsample=(i != n - 1)))
            self.mid_block = MidBlock(s.sub("mid_block"), heads[-1])
        nz = 0
        while s.has(f"controlnet_down_blocks.{nz}.weight"):
            nz += 1
        self.controlnet_down_blocks = [Linear(s.sub(f"controlnet_down_blocks.{i}")) for i in range(nz)]
        self.controlnet_mid_block = Linear(s.sub("controlnet_mid_block"))
        self.cond_embedding = _CondEmbedding(s.sub("controlnet_cond_embedding"))
        fe = s.sub("flow_encoder")
        self.flow_encoders, self.flow_zeroconvs = [], []
        i = 0
        while fe.has(f"encoders.{i}.conv_in.weight"):
            self.flow_encoders.append(Conv3x3(fe.sub(f"encoders.{i}.conv_in"), stride=2))
            self.flow_zeroconvs.append(Linear(fe.sub(f"zeroconvs.{i}")) if fe.has(f"zeroconvs.{i}.weight") else None)
            i += 1

    @classmethod
    def from_module(cls, module, device="cuda"):
        return cls(module.state_dict(), getattr(module, "config", None), device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", variant=None, **unused):
        """checkpoint directory of the trained adapter (MOFA-Video-Traj/run_gradio.py:106-110).  As in the reference, the
        trunk architecture does NOT come from ``config.json``: ``FlowControlNet.__init__`` calls ``super().__init__()``
        without arguments (svdxt_..._norefine.py:213), so it is always ControlNetSDVModel's default (heads (5, 10, 10, 20))"""
        from . import checkpoint
        path = checkpoint.resolve_dir(pretrained_model_name_or_path, subfolder)
        checkpoint.load_config(path)                                  # must exist, as for diffusers' loader
        return cls(checkpoint.load_state_dict(path, variant), None, device)

    @classmethod
    def _schema(cls, config=None):
        from . import schema
        return schema.controlnet_schema(config)

    @classmethod
    def from_unet(cls, unet, load_weights_from_unet=True, device="cuda", seed=0, config=None, **unused):
        """``ControlNetSDVModel.from_unet`` (MOFA-Video-Traj/models/controlnet_sdv.py:572-628): a fresh adapter whose
        ``conv_in`` / ``time_embedding`` / ``down_blocks`` / ``mid_block`` are copies of the UNet's, zero-initialised output
        convolutions, default-initialised everything else.  ``unet``: a state_dict, or anything with ``state_dict()``"""
        from . import checkpoint
        sd = unet if isinstance(unet, dict) else unet.state_dict()
        return cls(checkpoint.controlnet_state_dict_from_unet(sd, cls._schema(config), load_weights_from_unet, seed), config, device)

    # -- timestep-invariant adapter work (svdxt_...norefine.py:297-319) -------------------------------------
    def prepare_condition(self, controlnet_cond, controlnet_flow, frames=None):
        """controlnet_cond [1,3,H,W]; controlnet_flow [1,T-1,2,H,W] -> list of 4 token-major fp16 tensors
        [T*h_l*w_l, C_l]: frame 0 = the first-frame feature, frames 1.. = its forward-splat by flow 0->i.
        frames = (f0, f1): only that frame range (a rank's shard of the clip)."""
        cond = controlnet_cond.to(self.device, torch.float32)
        flow = controlnet_flow.to(self.device, torch.float32)
        assert cond.shape[0] == 1 and flow.shape[0] == 1, "one clip per call (both CFG halves share it)"
        _, _, H, W = cond.shape
        x = ops.nchw_to_tokens(cond, ld=self.cond_embedding.in_ld)
        f, h, w = self.cond_embedding(x, H, W)                              # [h*w, 320] at H/8
        feats = [(f, h, w)]
        e = f
        for enc, zc in zip(self.flow_encoders, self.flow_zeroconvs):
            e = enc(e, h, w, act=L.ACT_SILU)
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            feats.append((zc(e) if zc is not None else e, h, w))
        T = flow.shape[1] + 1
        f0, f1 = frames if frames is not None else (0, T)
        w0 = max(f0, 1)                                                     # first warped frame of the range
        fl = flow[0, w0 - 1:f1 - 1].contiguous() if f1 > w0 else None       # flows 0 -> i for i in [w0, f1)
        warped = []
        for (ft, h, w) in feats:
            s = H // h
            hw = h * w
            allf = torch.empty(((f1 - f0) * hw, ft.shape[1]), dtype=torch.float16, device=self.device)
            off = 0
            if f0 == 0:
                ops.copy2d(ft, allf[:hw])
                off = hw
            if fl is not None:
                fs = ops.flow_downscale(fl, s)                              # F.interpolate(nearest, 1/s) / s
                wr = ops.softsplat_avg_tokens(ft, fs, h, w)                 # [(f1-w0)*h*w, C]
                ops.copy2d(wr, allf[off:])
            warped.append(allf)
        return warped

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

    def _add_warped(self, sample, warped, B):
        rows = warped.shape[0]
        for b in range(B):
            ops.axpby_(warped, sample[b * rows:(b + 1) * rows], 1.0, 1.0)

    def forward_tokens(self, x, c, H, W, warped, conditioning_scale=1.0):
        """warped: the list from ``prepare_condition`` (or an ``AdapterCondition`` carrying .warped and, for the
        landmark adapter, .ldmk = {h*w: landmark embedding tokens}).
        -> (12 residual token tensors, mid residual), already multiplied by conditioning_scale."""
        return drive(self.forward_layers(x, c, H, W, warped, conditioning_scale))

    def forward_layers(self, x, c, H, W, warped, conditioning_scale=1.0):
        """``forward_tokens`` as a layer generator (blocks.run_lockstep)"""
        B = c.B
        cs = float(conditioning_scale)
        ldmk = getattr(warped, "ldmk", None)
        warped = getattr(warped, "warped", warped)
        c0 = self.config.block_out_channels[0]
        sample = self.conv_in(x, H, W)
        self._add_warped(sample, warped[0], B)                               # :328
        if ldmk is not None:
            self._add_warped(sample, ldmk[H * W], B)                         # ldmk_ctrlnet.py:474
        zi = 0
        outs = [self.controlnet_down_blocks[zi](sample, s_acc=cs)]           # zero conv applied eagerly so the
        zi += 1                                                              # later in-place adds are safe
        count, length = 1, len(warped)
        for blk in self.down_blocks:
            sample, H, W, res = yield from blk.layers(sample, c, H, W)
            for (r, _, _) in res:
                outs.append(self.controlnet_down_blocks[zi](r, s_acc=cs))
                zi += 1
            self._add_warped(sample, warped[min(count, length - 1)], B)      # :349
            if ldmk is not None and sample.shape[1] == c0:                   # ldmk_ctrlnet.py:501-504 (== 320)
                self._add_warped(sample, ldmk[H * W], B)
            count += 1
            yield
        self._add_warped(sample, warped[-1], B)                              # :354
        sample = yield from self.mid_block.layers(sample, c, H, W)
        mid = self.controlnet_mid_block(sample, s_acc=cs)
        return outs, mid

    # reference signature ------------------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                controlnet_flow=None, image_only_indicator=None, return_dict=True, guess_mode=False,
                conditioning_scale=1.0):
        B, T, Cin, H, W = sample.shape
        c = self.make_ctx(timestep, encoder_hidden_states, added_time_ids, B, T)
        # the reference feeds identical cond/flow to both CFG halves (pipeline.py:393-397); the warp set is
        # computed from batch element 0 and shared
        warped = self.prepare_condition(controlnet_cond[:1], controlnet_flow[:1])
        x = ops.nchw_to_tokens(sample.reshape(B * T, Cin, H, W).to(self.device, torch.float32), ld=self.in_ld)
        outs, mid = self.forward_tokens(x, c, H, W, warped, conditioning_scale)
        dims = []
        h, w = H, W
        dims.append((h, w))
        for i, blk in enumerate(self.down_blocks):
            for _ in blk.resnets:
                dims.append((h, w))
            if blk.down is not None:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                dims.append((h, w))
        res = [ops.tokens_to_nchw(o, B * T, o.shape[1], hh, ww).to(sample.dtype) for o, (hh, ww) in zip(outs, dims)]
        midn = ops.tokens_to_nchw(mid, B * T, mid.shape[1], h, w).to(sample.dtype)
        if not return_dict:
            return (res, midn, controlnet_flow, None)
        return _Config(down_block_res_samples=res, mid_block_res_sample=midn, controlnet_flow=controlnet_flow,
                       cmp_output=None)

    __call__ = forward



# =========================================================================================================
# landmark MOFA-Adapter (Hybrid / Keypoint trees)
# =========================================================================================================
class AdapterCondition:
    """timestep-invariant adapter state of one clip (or one window): warped first-frame pyramids, landmark embedding
    pyramid, occlusion masks"""

    def __init__(self, warped, ldmk=None, occlusion_masks=None):
        self.warped, self.ldmk, self.occlusion_masks = warped, ldmk, occlusion_masks


class _Matting:
    """ForegroundMatting (MOFA-Video-Hybrid/models/occlusion/hourglass.py:227-280) over all flow frames at once."""

    def __init__(self, s, C):
        self.C = C
        hg = s.sub("hourglass")
        self.enc = [Conv3x3(hg.sub(f"encoder.down_blocks.{i}.conv")) for i in range(3)]
        self.dec = [Conv3x3(hg.sub(f"decoder.up_blocks.{i}.conv")) for i in range(3)]
        self.head_mask = Conv3x3(s.sub("matting_mask"), pad_n=True)         # 7x7, 64 -> 1 (padded to 4 rows)
        self.head_mat = Conv3x3(s.sub("matting"))                           # 7x7, 64 -> C
        self.in_ld = self.enc[0].w.shape[1] // 9                            # 2C+2 padded to a multiple of 64

    def __call__(self, ref, flow, warped, nf, h, w):
        """ref fp16 [h*w, C]; flow fp32 [nf,2,h,w]; warped fp16 [nf*h*w, C] -> (blended [nf*h*w, C], mask fp32 [nf*h*w])"""
        C, hw = self.C, h * w
        x = torch.zeros((nf * hw, self.in_ld), dtype=torch.float16, device=ref.device)
        for i in range(nf):
            ops.copy2d(ref, x[i * hw:(i + 1) * hw, :C])
        ops.nchw_to_tokens(flow, out=x[:, C:C + 2])
        self._copy_unaligned(warped, x, C)     # column C+2 is only 4-byte aligned: scalar-store path
        relu = dict(act=L.ACT_RELU)
        e1 = self.enc[0](x, h, w, **relu)
        e2 = self.enc[1](e1, h, w, **relu)
        e3 = self.enc[2](e2, h, w, **relu)
        d = self.dec[0](e3, h, w, **relu)
        d = self.dec[1](ops.concat_channels(e2, d), h, w, **relu)
        d = self.dec[2](ops.concat_channels(e1, d), h, w, **relu)
        g7 = ops.conv3x3_geom(h, w, ksize=7)
        logit = ops.igemm(d, self.head_mask.w, self.head_mask.b, geom=g7)    # [nf*hw, 4], column 0 = logit
        mat = ops.igemm(d, self.head_mat.w, self.head_mat.b, geom=g7)
        return ops.matting_blend(warped, mat, logit)

    @staticmethod
    def _copy_unaligned(warped, x, C):
        # the warped block starts at column C+2 (4-byte aligned only): go through an fp32 NCHW view of the tokens
        n = warped.shape[0]
        t = ops.tokens_to_nchw(warped, 1, warped.shape[1], n, 1)            # [1, C, n, 1] fp32
        ops.nchw_to_tokens(t, out=x[:, C + 2:2 * C + 2])


class LandmarkFlowControlNet(FlowControlNet):
    """``FlowControlNet`` of MOFA-Video-Hybrid/models/ldmk_ctrlnet.py (= MOFA-Video-Keypoint/models/ldmk_ctrlnet.py):
    the trajectory adapter (first-frame encoder without zero convs) + landmark-image embedding added at the 320-channel
    stages + per-scale ForegroundMatting and zero-out on every warped frame.  ``forward`` adds the ``landmarks``
    argument and returns the occlusion masks 4th (:322-339, :569-570)."""

    @classmethod
    def _schema(cls, config=None):
        from . import schema
        return schema.ldmk_controlnet_schema(config)

    def __init__(self, state_dict, config=None, device="cuda", dtype=torch.float16):
        super().__init__(state_dict, config, device, dtype)
        s = Sub(state_dict, "", device)
        boc = tuple(self.config.block_out_channels)
        self.ldmk_embedding = _CondEmbedding(s.sub("controlnet_ldmk_embedding"))
        ch = {"8": boc[0], "16": boc[0], "32": boc[1], "64": boc[2]}
        self.zero_outs = {k: Linear(s.sub(f"zero_outs.{k}")) for k in ch}
        self.occlusions = {k: _Matting(s.sub(f"occlusions.{k}"), c) for k, c in ch.items()}

    def prepare_condition(self, controlnet_cond, controlnet_flow, landmarks=None, frames=None):
        """landmarks [1,T,3,H,W] (pose images).  -> AdapterCondition for frames [f0, f1)."""
        cond = controlnet_cond.to(self.device, torch.float32)
        flow = controlnet_flow.to(self.device, torch.float32)
        assert cond.shape[0] == 1 and flow.shape[0] == 1
        _, _, H, W = cond.shape
        T = flow.shape[1] + 1
        f0, f1 = frames if frames is not None else (0, T)
        x = ops.nchw_to_tokens(cond, ld=self.cond_embedding.in_ld)
        f, h, w = self.cond_embedding(x, H, W)
        feats = [(f, h, w)]
        e = f
        for enc in self.flow_encoders:                                       # no zero convs (ldmk_ctrlnet.py:152-161)
            e = enc(e, h, w, act=L.ACT_SILU)
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            feats.append((e, h, w))
        w0 = max(f0, 1)
        fl = flow[0, w0 - 1:f1 - 1].contiguous() if f1 > w0 else None
        warped, masks = [], []
        for (ft, h, w) in feats:
            s = H // h
            hw = h * w
            allf = torch.empty(((f1 - f0) * hw, ft.shape[1]), dtype=torch.float16, device=self.device)
            off = 0
            if f0 == 0:
                ops.copy2d(ft, allf[:hw])
                off = hw
            if fl is not None:
                nf = fl.shape[0]
                fs = ops.flow_downscale(fl, s)
                wr = ops.softsplat_avg_tokens(ft, fs, h, w)                  # :301
                wr, m = self.occlusions[str(s)](ft, fs, wr, nf, h, w)        # :310-312
                self.zero_outs[str(s)](wr, out=allf[off:])                   # :316
                masks.append(m.reshape(nf, 1, h, w))
            warped.append(allf)
        ldmk = None
        if landmarks is not None:
            lm = landmarks.to(self.device, torch.float32)[0, f0:f1].contiguous()      # [Tl,3,H,W]
            Tl = lm.shape[0]
            xl = ops.nchw_to_tokens(lm, ld=self.ldmk_embedding.in_ld)
            l0, lh, lw = self.ldmk_embedding(xl, H, W)                       # [Tl*h*w, 320]
            ldmk = {lh * lw: l0}
            half = ops.subsample_tokens(l0, Tl, lh, lw, 2)                   # F.interpolate(scale_factor=1/2), :399-403
            ldmk[(lh // 2) * (lw // 2)] = half
        return AdapterCondition(warped, ldmk, masks)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, controlnet_cond=None,
                controlnet_flow=None, landmarks=None, image_only_indicator=None, return_dict=True, guess_mode=False,
                conditioning_scale=1.0):
        B, T, Cin, H, W = sample.shape
        c = self.make_ctx(timestep, encoder_hidden_states, added_time_ids, B, T)
        cond = self.prepare_condition(controlnet_cond[:1], controlnet_flow[:1], landmarks[:1])
        x = ops.nchw_to_tokens(sample.reshape(B * T, Cin, H, W).to(self.device, torch.float32), ld=self.in_ld)
        outs, mid = self.forward_tokens(x, c, H, W, cond, conditioning_scale)
        dims, h, w = [(H, W)], H, W
        for blk in self.down_blocks:
            dims += [(h, w)] * len(blk.resnets)
            if blk.down is not None:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                dims.append((h, w))
        res = [ops.tokens_to_nchw(o, B * T, o.shape[1], hh, ww).to(sample.dtype) for o, (hh, ww) in zip(outs, dims)]
        midn = ops.tokens_to_nchw(mid, B * T, mid.shape[1], h, w).to(sample.dtype)
        occ = [m.unsqueeze(0).expand(B, -1, -1, -1, -1) for m in cond.occlusion_masks]
        if not return_dict:
            return (res, midn, controlnet_flow, occ)
        return _Config(down_block_res_samples=res, mid_block_res_sample=midn, controlnet_flow=controlnet_flow,
                       occlusion_masks=occ)

    __call__ = forward
