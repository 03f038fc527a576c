"""MI355X host mirror of ``FlowControlNetPipeline`` (MOFA-Video-Traj/pipeline/pipeline.py:87-527).

Same constructor modules and ``__call__`` signature / defaults / return type.  The denoise loop runs
entirely in libmofa_hip.so on token-major fp16 activations:

    per clip (hoisted, timestep-invariant -- SURVEY F7/F11):
        adapter.prepare_condition  (cond CNN, first-frame pyramid, flow pyramids, 96 forward-splat warps)
        cross-attention row vectors and frame-position embeddings of every transformer layer
    per step:
        mofa_prepare_model_input   (scale by 1/sqrt(sigma^2+1), concat image latents, both CFG halves)
        FlowControlNet.forward_tokens -> 12 + 1 residuals
        UNet.forward_tokens           -> noise prediction
        mofa_cfg_euler_step         (CFG with per-frame guidance + v-prediction Euler step, fp32 latents)
    decode: temporal VAE decoder, chunks of ``decode_chunk_size`` frames.

Image conditioning (once per clip before the loop, pipeline.py:330-352) also runs on the library when the
pipeline holds an ``image_encoder`` (mofa_video_amd.clip) and a VAE with encoder weights: ``image`` is then the PIL image /
[1,3,H,W] tensor in [0, 1] of the reference call (mofa_video_amd/frontend.py).  Precomputed conditioning can be passed
instead through the keyword-only extensions ``image_embeddings`` ([1,1,1024] or [2,1,1024]) and ``image_latents``
([1,4,h,w] or [2,4,h,w]).
Reference quirks kept: ``added_time_ids`` is always [6, 128, 0.02] (pipeline.py:430-440); CFG is always on
(max_guidance_scale > 1); the scheduler's unused per-step randn draw is not reproduced (no effect on results).
"""
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import frontend, ops
from .blocks import Ctx
from .output import tensor2vid
from .vae import decode_latents


@dataclass
class FlowControlNetPipelineOutput:
    frames: Union[List, np.ndarray, torch.FloatTensor]
    controlnet_flow: Union[List, np.ndarray, torch.FloatTensor]


def _to_tensor_image(image, height, width, device):
    """controlnet_condition: VaeImageProcessor.preprocess for tensors is a resize + (2x-1); here only tensors in [-1,1]
    that are already H x W are accepted (the conditioning *image* goes through frontend.image_to_01 instead)."""
    if not torch.is_tensor(image):
        raise ValueError("mofa_video_amd pipeline expects a torch tensor [1,3,H,W] in [-1,1] for image / "
                         f"controlnet_condition, got {type(image)}")
    if image.dim() == 3:
        image = image.unsqueeze(0)
    if tuple(image.shape[-2:]) != (height, width):
        raise ValueError(f"image is {tuple(image.shape[-2:])}, expected ({height}, {width})")
    return image.to(device, torch.float32)


class FlowControlNetPipeline:
    def __init__(self, vae=None, image_encoder=None, unet=None, controlnet=None, scheduler=None,
                 feature_extractor=None, parallel=None):
        """parallel: optional ``parallel.FrameParallel`` -- this process then computes one CFG half / one frame shard
        of every clip (mofa_video_amd/parallel.py); all ranks must call the pipeline with identical inputs."""
        self.vae, self.image_encoder, self.unet, self.controlnet = vae, image_encoder, unet, controlnet
        self.scheduler, self.feature_extractor = scheduler, feature_extractor
        self.vae_scale_factor = 8
        self.device = unet.device
        self.parallel = parallel

    def check_inputs(self, image, height, width):                      # pipeline.py:222-234
        if image is not None and not torch.is_tensor(image) and not isinstance(image, list) and not hasattr(image, "convert"):
            raise ValueError("`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or "
                             f"`List[PIL.Image.Image]` but is {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def prepare_latents(self, batch_size, num_frames, num_channels_latents, height, width, generator, latents=None):
        shape = (batch_size, num_frames, num_channels_latents // 2, height // 8, width // 8)
        if latents is None:
            latents = torch.randn(shape, generator=generator, dtype=torch.float32,
                                  device=generator.device if generator is not None else "cpu")
        return latents.to(self.device, torch.float32) * self.scheduler.init_noise_sigma   # :272

    def _encode_image(self, image01):                                   # pipeline.py:114-139
        if self.image_encoder is None:
            raise ValueError("no image_encoder: pass image_embeddings=... or build the pipeline with one")
        return frontend.encode_image(self.image_encoder, image01)

    def _encode_vae_image(self, image01, noise_aug_strength, generator):  # pipeline.py:141-162, :338-352
        if self.vae is None or getattr(self.vae, "encoder", None) is None:
            raise ValueError("the VAE has no encoder weights: pass image_latents=... or load encoder.* / quant_conv.*")
        return frontend.encode_vae_image(self.vae, image01, noise_aug_strength, generator)

    def _conditioning(self, image, image_embeddings, image_latents, height=None, width=None, noise_aug_strength=0.02,
                      generator=None):
        """image: PIL / tensor in [0, 1] (what the reference's numpy_to_pt yields); either half can be supplied
        precomputed through the ``image_embeddings`` / ``image_latents`` extensions."""
        dev = self.device
        image01 = None
        if image_embeddings is None or image_latents is None:
            if image is None:
                raise ValueError("pass `image`, or both image_embeddings=... and image_latents=...")
            image01 = frontend.image_to_01(image, height, width, dev)
        if image_embeddings is None:
            image_embeddings = self._encode_image(image01)
        if image_latents is None:
            image_latents = self._encode_vae_image(image01, noise_aug_strength, generator)
        emb = image_embeddings.to(dev, torch.float32).reshape(-1, 1, image_embeddings.shape[-1])
        if emb.shape[0] == 1:                                             # :133-139 uncond = zeros
            emb = torch.cat([torch.zeros_like(emb), emb])
        il = image_latents.to(dev, torch.float32)
        if il.shape[0] == 1:                                              # :153-159
            il = torch.cat([torch.zeros_like(il), il])
        return emb, il.contiguous()

    # ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, height: int = 576,
                 width: int = 1024, num_frames: Optional[int] = None, num_inference_steps: int = 25,
                 min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7,
                 motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1,
                 generator=None, latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pt",
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_dict: bool = True,
                 controlnet_cond_scale=1.0, batch_size=1, *, image_embeddings=None, image_latents=None):
        unet, cn, sch, dev = self.unet, self.controlnet, self.scheduler, self.device
        num_frames = num_frames if num_frames is not None else unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        self.check_inputs(image, height, width)
        if batch_size != 1 or num_videos_per_prompt != 1:
            raise ValueError("one clip per call (as every reference entry point does)")
        if not max_guidance_scale > 1.0:
            raise ValueError("the reference pipeline is only well-defined with classifier-free guidance on")
        h, w = height // 8, width // 8
        T = num_frames

        # 3./4. image conditioning (computed before the hot path)
        emb, il = self._conditioning(image, image_embeddings, image_latents, height, width, noise_aug_strength, generator)

        # 4./5. schedule + latents (every rank prepares the full clip's latents; it keeps its own frames below)
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps
        lat = self.prepare_latents(1, T, unet.config.in_channels, height, width, generator, latents)
        lat = lat.reshape(T, 4, h, w).contiguous()

        # adapter condition: identical for both CFG halves (:393-397) -> computed once
        cond = _to_tensor_image(controlnet_condition, height, width, dev)
        flow = controlnet_flow.to(dev, torch.float32)
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)   # :430-440

        par = self.parallel
        lay = par.lay if par is not None else None
        if lay is not None:
            assert lay.T == T, "Layout was built for a different frame count"
        f0, f1 = (lay.f0, lay.f1) if lay is not None else (0, T)
        Tl = f1 - f0                                                      # frames held by this rank
        Bl = lay.B_loc if lay is not None else 2                          # CFG halves computed by this rank
        half = lay.half if lay is not None else None
        fpar = par if (lay is not None and lay.sharded_frames) else None
        warped = cn.prepare_condition(cond[:1], flow[:1], frames=(f0, f1))
        lat = lat[f0:f1].contiguous()
        gspan = (max_guidance_scale - min_guidance_scale) / max(T - 1, 1)  # per-frame guidance is linear in the frame
        g0, g1 = min_guidance_scale + gspan * f0, min_guidance_scale + gspan * (f1 - 1)

        c_cn, c_un = Ctx(Bl, Tl), Ctx(Bl, Tl)                             # hold the per-clip invariant caches
        rows = Tl * h * w
        x_in = torch.zeros((2 * rows, max(unet.in_ld, cn.in_ld)), dtype=torch.float16, device=dev)
        x_loc = x_in if Bl == 2 else x_in[half * rows:(half + 1) * rows]
        self._num_timesteps = len(timesteps)
        for i, t in enumerate(timesteps):                                 # :447-511
            sigma, sigma_next = sch.sigma_pair(i)
            ops.prepare_model_input(lat, il, x_in, sigma)
            cn.make_ctx(float(t), emb, added_time_ids, Bl, Tl, base=c_cn, half=half, par=fpar)
            down_res, mid_res = cn.forward_tokens(x_loc, c_cn, h, w, warped, controlnet_cond_scale)
            unet.make_ctx(float(t), emb, added_time_ids, Bl, Tl, base=c_un, half=half, par=fpar)
            noise = unet.forward_tokens(x_loc, c_un, h, w, down_res, mid_res)
            if Bl == 1:
                noise = par.gather_cfg(noise)                             # both halves of this frame shard
            ops.cfg_euler_step_(lat, noise, sigma, sigma_next, g0, g1)
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, t, {"latents": lat.reshape(1, Tl, 4, h, w)})
                if out and "latents" in out:
                    lat = out["latents"].to(dev, torch.float32).reshape(Tl, 4, h, w).contiguous()

        if fpar is not None:                                              # reassemble the clip's latents on every rank
            lat = fpar.gather_frames(lat.reshape(Tl, 4 * h * w), 1).reshape(T, 4, h, w)
        latents_out = lat.reshape(1, T, 4, h, w)
        if output_type == "latent":
            frames = latents_out
        elif lay is None or lay.world == 1:
            frames = decode_latents(self.vae, latents_out, T, decode_chunk_size)       # fp32 [1,3,T,H,W]
            if output_type != "raw":                                                   # pipeline.py:518
                frames = tensor2vid(frames, None, output_type=output_type)
        else:
            # VAE chunks are independent (pipeline.py:204-213): dealt round-robin to all ranks; the result is the list
            # of (first_frame, fp32 [n,3,H,W]) chunks this rank decoded
            frames = []
            sf = 1.0 / self.vae.config.scaling_factor
            for ci, s0 in enumerate(range(0, T, decode_chunk_size)):
                if ci % lay.world == lay.rank:
                    z = latents_out[0, s0:s0 + decode_chunk_size]
                    fr = self.vae.decode(z, num_frames=z.shape[0], _prescale=sf)           # fp32 [n,3,H,W]
                    if output_type != "raw":
                        fr = tensor2vid(fr.permute(1, 0, 2, 3).unsqueeze(0), None, output_type=output_type)[0]
                    frames.append((s0, fr))
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)



# =========================================================================================================
# Hybrid: face (landmark) adapter + drag (trajectory) adapter, residuals blended by the user mask
# (MOFA-Video-Hybrid/pipeline/pipeline.py:293-320 signature, :443-507 loop, :479-489 blend)
# =========================================================================================================
class HybridFlowControlNetPipeline(FlowControlNetPipeline):
    def __init__(self, vae=None, image_encoder=None, unet=None, face_controlnet=None, drag_controlnet=None,
                 scheduler=None, feature_extractor=None, parallel=None):
        super().__init__(vae, image_encoder, unet, face_controlnet, scheduler, feature_extractor, parallel=parallel)
        self.face_controlnet, self.drag_controlnet = face_controlnet, drag_controlnet

    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, landmarks=None, drag_flow=None,
                 mask=None, height: int = 576, width: int = 1024, num_frames: Optional[int] = None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1, generator=None,
                 latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pt",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 return_dict: bool = True, ctrl_scale_traj=1.0, ctrl_scale_ldmk=1.0, batch_size=1, *,
                 image_embeddings=None, image_latents=None):
        unet, face, drag, sch, dev = self.unet, self.face_controlnet, self.drag_controlnet, self.scheduler, self.device
        T = num_frames if num_frames is not None else unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else T
        self.check_inputs(image, height, width)
        h, w = height // 8, width // 8
        emb, il = self._conditioning(image, image_embeddings, image_latents, height, width, noise_aug_strength, generator)
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps
        lat = self.prepare_latents(1, T, unet.config.in_channels, height, width, generator, latents).reshape(T, 4, h, w).contiguous()
        cond = _to_tensor_image(controlnet_condition, height, width, dev)
        # frame sharding exactly as in FlowControlNetPipeline.__call__ (2-way CFG x frame shards)
        par = self.parallel
        lay = par.lay if par is not None else None
        if lay is not None:
            assert lay.T == T, "Layout was built for a different frame count"
        f0, f1 = (lay.f0, lay.f1) if lay is not None else (0, T)
        Tl = f1 - f0
        Bl = lay.B_loc if lay is not None else 2
        half = lay.half if lay is not None else None
        fpar = par if (lay is not None and lay.sharded_frames) else None
        cf = face.prepare_condition(cond[:1], controlnet_flow.to(dev, torch.float32)[:1], landmarks[:1], frames=(f0, f1))
        cd = drag.prepare_condition(cond[:1], drag_flow.to(dev, torch.float32)[:1], frames=(f0, f1))
        lat = lat[f0:f1].contiguous()
        gspan = (max_guidance_scale - min_guidance_scale) / max(T - 1, 1)
        g0, g1 = min_guidance_scale + gspan * f0, min_guidance_scale + gspan * (f1 - 1)
        # user mask, nearest-resized to every residual resolution (:481, :488) -- timestep-invariant
        m = mask.to(dev, torch.float32).reshape(1, height, width)
        masks = {}
        hh, ww = h, w
        for _ in range(4):
            masks[hh * ww] = ops.resize_nearest_f32(m, hh, ww).reshape(-1).contiguous()
            hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
        c_f, c_d, c_u = Ctx(Bl, Tl), Ctx(Bl, Tl), Ctx(Bl, Tl)
        rows = Tl * h * w
        x_in = torch.zeros((2 * rows, unet.in_ld), dtype=torch.float16, device=dev)
        x_loc = x_in if Bl == 2 else x_in[half * rows:(half + 1) * rows]
        for i, t in enumerate(timesteps):
            sigma, sigma_next = sch.sigma_pair(i)
            ops.prepare_model_input(lat, il, x_in, sigma)
            face.make_ctx(float(t), emb, added_time_ids, Bl, Tl, base=c_f, half=half, par=fpar)
            df, mf = face.forward_tokens(x_loc, c_f, h, w, cf, ctrl_scale_ldmk)
            drag.make_ctx(float(t), emb, added_time_ids, Bl, Tl, base=c_d, half=half, par=fpar)
            dd, md = drag.forward_tokens(x_loc, c_d, h, w, cd, ctrl_scale_traj)
            down = []
            for a, b in zip(df, dd):
                hw = a.shape[0] // (Bl * Tl)
                down.append(ops.mask_blend(a, b, masks[hw], hw))
            hw = mf.shape[0] // (Bl * Tl)
            mid = ops.mask_blend(mf, md, masks[hw], hw)
            unet.make_ctx(float(t), emb, added_time_ids, Bl, Tl, base=c_u, half=half, par=fpar)
            noise = unet.forward_tokens(x_loc, c_u, h, w, down, mid)
            if Bl == 1:
                noise = par.gather_cfg(noise)
            ops.cfg_euler_step_(lat, noise, sigma, sigma_next, g0, g1)
        if fpar is not None:
            lat = fpar.gather_frames(lat.reshape(Tl, 4 * h * w), 1).reshape(T, 4, h, w)
        latents_out = lat.reshape(1, T, 4, h, w)
        if output_type == "latent":
            frames = latents_out
        elif lay is None or lay.world == 1:
            frames = decode_latents(self.vae, latents_out, T, decode_chunk_size)
            if output_type != "raw":
                frames = tensor2vid(frames, None, output_type=output_type)
        else:                                             # VAE chunks dealt round-robin, as in FlowControlNetPipeline
            frames = []
            sf = 1.0 / self.vae.config.scaling_factor
            for ci, s0 in enumerate(range(0, T, decode_chunk_size)):
                if ci % lay.world == lay.rank:
                    z = latents_out[0, s0:s0 + decode_chunk_size]
                    fr = self.vae.decode(z, num_frames=z.shape[0], _prescale=sf)
                    if output_type != "raw":
                        fr = tensor2vid(fr.permute(1, 0, 2, 3).unsqueeze(0), None, output_type=output_type)[0]
                    frames.append((s0, fr))
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)


# =========================================================================================================
# Keypoint long video ("periodic sampling"): overlapping temporal windows with frame 0 prepended, one Euler step per
# window, overlap average (MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:289-294 signature,
# :426-429 views, :445-511 loop)
# =========================================================================================================
def window_views(num_frames, window_size, stride):
    window_num = (num_frames - window_size) // stride + 1
    views = [(1 + i * stride, i * stride + window_size) for i in range(window_num)]
    return views + [(num_frames - window_size + 1, num_frames)]


class KeypointFlowControlNetPipeline(FlowControlNetPipeline):
    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, landmarks=None, window_size: int = 25,
                 stride: int = 12, height: int = 576, width: int = 1024, num_frames: Optional[int] = None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1, generator=None,
                 latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pt",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 return_dict: bool = True, controlnet_cond_scale=1.0, batch_size=1, *, image_embeddings=None,
                 image_latents=None):
        unet, cn, sch, dev = self.unet, self.controlnet, self.scheduler, self.device
        N = num_frames if num_frames is not None else unet.config.num_frames
        Tw = window_size
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else N
        self.check_inputs(image, height, width)
        h, w = height // 8, width // 8
        emb, il = self._conditioning(image, image_embeddings, image_latents, height, width, noise_aug_strength, generator)
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps
        lat = self.prepare_latents(1, N, unet.config.in_channels, height, width, generator, latents).reshape(N, 4, h, w).contiguous()
        cond = _to_tensor_image(controlnet_condition, height, width, dev)
        flow = controlnet_flow.to(dev, torch.float32)
        views = window_views(N, Tw, stride)
        # adapter state per DISTINCT window is timestep-invariant: computed once per clip (the reference recomputes it
        # every step; its last view often repeats the previous one -- SURVEY 3.5 -- and is computed once here)
        conds = {}
        for (t0, t1) in views:
            if (t0, t1) not in conds:
                lm = torch.cat([landmarks[:, 0:1], landmarks[:, t0:t1]], dim=1)
                conds[(t0, t1)] = cn.prepare_condition(cond[:1], flow[:1, t0 - 1:t1 - 1], lm)
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
        ctxs = {v: (Ctx(2, Tw), Ctx(2, Tw)) for v in conds}
        distinct = list(conds)                                                     # distinct windows, in view order
        wpar = self.parallel                                                       # parallel.WindowParallel or None
        x_in = torch.zeros((2 * Tw * h * w, unet.in_ld), dtype=torch.float16, device=dev)
        value = torch.empty_like(lat)
        fsz = 4 * h * w
        for i, t in enumerate(timesteps):
            sigma, sigma_next = sch.sigma_pair(i)
            count = [0] * N
            touched = [False] * N
            done = {}

            def step_window(t0, t1):
                lw = torch.cat([lat[0:1], lat[t0:t1]], dim=0).contiguous()            # frame 0 + window frames
                ops.prepare_model_input(lw, il, x_in, sigma)
                c_cn, c_un = ctxs[(t0, t1)]
                cn.make_ctx(float(t), emb, added_time_ids, 2, Tw, base=c_cn)
                down, mid = cn.forward_tokens(x_in, c_cn, h, w, conds[(t0, t1)], controlnet_cond_scale)
                unet.make_ctx(float(t), emb, added_time_ids, 2, Tw, base=c_un)
                noise = unet.forward_tokens(x_in, c_un, h, w, down, mid)
                ops.cfg_euler_step_(lw, noise, sigma, sigma_next, min_guidance_scale, max_guidance_scale)
                return lw
            if wpar is None:
                for key in distinct:
                    done[key] = step_window(*key)
            else:                                         # window-parallel: one window per rank and round
                for rnd in wpar.rounds(distinct):
                    mine = rnd[wpar.rank]
                    lw = step_window(*mine) if mine is not None else torch.zeros((Tw,) + tuple(lat.shape[1:]),
                                                                               dtype=lat.dtype, device=dev)
                    for key, got in zip(rnd, wpar.gather(lw)):
                        if key is not None:
                            done[key] = got
            for idx, (t0, t1) in enumerate(views):
                lw = done[(t0, t1)]
                # value[0:t1] += lw (first view) / value[t0:t1] += lw[1:] (others)   (:502-507)
                dst0, src0 = (0, 0) if idx == 0 else (t0, 1)
                if idx == 0 and t0 != 1:
                    raise ValueError("the first window must start at frame 1")
                for k in range(t1 - dst0):
                    f = dst0 + k
                    src = lw[src0 + k].reshape(-1)
                    dstv = value[f].reshape(-1)
                    ops.axpby_f32_(src, dstv, 1.0, 1.0 if touched[f] else 0.0)
                    touched[f] = True
                    count[f] += 1
            for f in range(N):                                                         # latents = value / count (:511)
                if count[f]:
                    ops.axpby_f32_(value[f].reshape(-1), lat[f].reshape(-1), 1.0 / count[f], 0.0)
        latents_out = lat.reshape(1, N, 4, h, w)
        frames = latents_out if output_type == "latent" else decode_latents(self.vae, latents_out, N, decode_chunk_size)
        if output_type not in ("latent", "raw"):
            frames = tensor2vid(frames, None, output_type=output_type)
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)
