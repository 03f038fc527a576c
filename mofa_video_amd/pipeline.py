"""MI355X host mirrors of the reference pipelines: ``FlowControlNetPipeline`` (MOFA-Video-Traj/pipeline/pipeline.py:87-527),
``HybridFlowControlNetPipeline`` (MOFA-Video-Hybrid/pipeline/pipeline.py:293-507) and ``KeypointFlowControlNetPipeline``
(MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:289-511).

Same constructor modules, ``__call__`` signatures, defaults (``output_type="pil"``) and return types.  The denoise loops run
entirely in libmofa_hip.so on token-major fp16 activations:

    per clip (hoisted, timestep-invariant -- SURVEY F7/F11):
        adapter.prepare_condition  (cond CNN, first-frame pyramid, flow pyramids, 96 forward-splat warps)
        cross-attention row vectors and frame-position embeddings of every transformer layer
    per step:
        mofa_prepare_model_input   (scale by 1/sqrt(sigma^2+1), concat image latents, both CFG halves)
        FlowControlNet.forward_tokens -> 12 + 1 residuals   (Hybrid: two adapters + mask blend)
        UNet.forward_tokens           -> noise prediction
        mofa_cfg_euler_step         (CFG with per-frame guidance + v-prediction Euler step, fp32 latents)
    decode: temporal VAE decoder, chunks of ``decode_chunk_size`` frames.

Within a step the adapter's ControlNet trunk(s) and the UNet's encoder half (conv_in, down blocks, mid block) do not depend
on each other (the residuals are added after the mid block): without frame sharding the trunks are enqueued on a second HIP
stream and the encoder on the caller's (``_denoise_forward``; ``overlap_adapter=False`` restores the single-stream order).  Every
launch is a grid of persistent one-per-CU workgroups, so the second stream's kernel takes exactly the CUs the first one's tail
round leaves idle, and one stream's launch gap is covered by the other's kernel: 3.0-4.4 % of a denoise step on one MI355X
(tools/two_stream_probe.py), bit-identical results (same kernels, same tiles, same order per stream).

Image conditioning (once per clip before the loop, pipeline.py:330-352) also runs on the library when the
pipeline holds an ``image_encoder`` (mofa_video_amd.clip) and a VAE with encoder weights: ``image`` is then the PIL image /
[1,3,H,W] tensor in [0, 1] of the reference call (mofa_video_amd/frontend.py).  Precomputed conditioning can be passed
instead through the keyword-only extensions ``image_embeddings`` ([1,1,1024] or [2,1,1024]) and ``image_latents``
([1,4,h,w] or [2,4,h,w]).

Reference quirks kept: ``added_time_ids`` is always [6, 128, 0.02] (pipeline.py:430-440); CFG is always on
(max_guidance_scale > 1).  Stated deviations: (1) the latents stay fp32 between steps unless the pipeline is built with
``round_latents_to_fp16=True`` (the reference's fp16 run rounds them every step, scheduling_..._karras_fix.py:520);
(2) random draws (initial latents, noise augmentation) come from ``torch.randn`` in fp32 on the generator's device, so a
seed does not reproduce the reference's fp16 CUDA draws -- pass ``latents=`` for reproducibility across implementations;
(3) the scheduler's unused per-step randn draw is not made (no effect on results).
"""
import os
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


MAX_TEMPORAL_FRAMES = 32        # mofa_attn_temporal_f16 holds one clip's keys in a wave: T <= 32
GRAPH_STEPS_DEFAULT = os.environ.get("MOFA_GRAPH_STEPS", "0") == "1"


class _Shard:
    """what this rank computes of a T-frame clip (mofa_video_amd/parallel.py): frames [f0, f1), Bl CFG halves"""

    def __init__(self, par, T):
        self.par = par
        self.lay = par.lay if par is not None else None
        if self.lay is not None and self.lay.T != T:
            raise ValueError(f"the parallel Layout was built for {self.lay.T} frames, the clip has {T}")
        self.f0, self.f1 = (self.lay.f0, self.lay.f1) if self.lay is not None else (0, T)
        self.Tl = self.f1 - self.f0
        self.Bl = self.lay.B_loc if self.lay is not None else 2
        self.half = self.lay.half if self.lay is not None else None
        self.fpar = par if (self.lay is not None and self.lay.sharded_frames) else None
        self.world = self.lay.world if self.lay is not None else 1
        self.rank = self.lay.rank if self.lay is not None else 0


class FlowControlNetPipeline:
    def __init__(self, vae=None, image_encoder=None, unet=None, controlnet=None, scheduler=None,
                 feature_extractor=None, parallel=None, round_latents_to_fp16=False):
        """parallel: optional ``parallel.FrameParallel`` -- this process then computes one CFG half / one frame shard
        of every clip (mofa_video_amd/parallel.py); all ranks must call the pipeline with identical inputs.
        round_latents_to_fp16: round the latents to fp16 after every Euler step, as the reference's fp16 run does."""
        self.vae, self.image_encoder, self.unet, self.controlnet = vae, image_encoder, unet, controlnet
        self.scheduler, self.feature_extractor = scheduler, feature_extractor
        self.vae_scale_factor = 8
        self.device = unet.device
        self.parallel = parallel
        self.round_latents_to_fp16 = round_latents_to_fp16
        self.overlap_adapter = True          # adapter trunk(s) || UNet encoder on two HIP streams (see _denoise_forward)
        self.split_decoder = True            # ... and the UNet decoder's two CFG halves on the same two streams
        # capture ONE denoise step per clip in a hipGraph and replay it for the remaining steps (no host scalar enters a step:
        # the scheduler's numbers come from a device table, _clip_loop).  Off for a rank that exchanges data inside a step (the
        # gloo / virtual-rank transports are host side), with a step callback, and while launches are being timed.
        self.graph_steps = GRAPH_STEPS_DEFAULT
        self._adapter_stream = None
        self._loop_stream = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, unet=None, controlnet=None, device="cuda", variant=None,
                        **modules):
        """the call of MOFA-Video-Traj/run_gradio.py:112-116: ``vae/``, ``image_encoder/`` (and ``unet/`` when not given) of
        a Stable-Video-Diffusion checkpoint directory; ``unet`` / ``controlnet`` (``face_controlnet`` / ``drag_controlnet``
        for the Hybrid class) are passed in already built, as the reference does.  Other diffusers keyword arguments
        (``torch_dtype``, ``low_cpu_mem_usage``, ``local_files_only`` ...) are accepted and ignored."""
        from .clip import CLIPVisionModelWithProjection
        from .scheduler import EulerDiscreteScheduler
        from .unet import UNetSpatioTemporalConditionControlNetModel
        from .vae import AutoencoderKLTemporalDecoder
        from . import checkpoint
        root = pretrained_model_name_or_path
        if unet is None:
            unet = UNetSpatioTemporalConditionControlNetModel.from_pretrained(root, subfolder="unet", device=device, variant=variant)
        vae = AutoencoderKLTemporalDecoder.from_pretrained(root, subfolder="vae", device=device, variant=variant)
        enc = CLIPVisionModelWithProjection.from_pretrained(root, subfolder="image_encoder", device=device, variant=variant)
        try:
            sch = EulerDiscreteScheduler(**{k: v for k, v in checkpoint.load_config(checkpoint.resolve_dir(root, "scheduler")).items()
                                            if k in EulerDiscreteScheduler().config})
        except OSError:
            sch = EulerDiscreteScheduler()
        names = ("face_controlnet", "drag_controlnet", "parallel", "round_latents_to_fp16", "feature_extractor")
        kw = {k: v for k, v in modules.items() if k in names}
        if controlnet is not None:
            kw["controlnet"] = controlnet
        return cls(vae=vae, image_encoder=enc, unet=unet, scheduler=sch, **kw)

    # ---- one network evaluation of a denoise step ---------------------------------------------------------------------------
    def _denoise_forward(self, x_loc, t, emb, added_time_ids, Bl, Tl, half, fpar, h, w, adapters, c_un, masks=None):
        """``adapters``: [(controlnet, its Ctx, its per-clip condition, conditioning scale)], one entry, or two (face, drag)
        whose residuals are blended by ``masks`` (Hybrid/pipeline/pipeline.py:479-489) -> the UNet's noise prediction (tokens).
        Without frame sharding the trunks run on ``self._adapter_stream`` while the UNet's encoder half runs on the caller's
        stream; the streams join before the residuals are added (unet.decode_tokens).  Frame-sharded ranks enqueue the two
        networks layer by layer in lockstep so that their exchanges are issued in program order (_denoise_forward_sharded)."""
        unet = self.unet

        def trunks():
            res = []
            for net, ctx, cond, scale in adapters:
                net.make_ctx(t, emb, added_time_ids, Bl, Tl, base=ctx, half=half, par=fpar)
                res.append(net.forward_tokens(x_loc, ctx, h, w, cond, scale))
            down, mid = res[0]
            if len(res) == 2:
                down, mid = _blend_residuals(down, mid, res[1][0], res[1][1], masks, Bl * Tl)
            return down, mid

        if not self.overlap_adapter or (fpar is not None and not fpar.two_streams):
            down, mid = trunks()
            unet.make_ctx(t, emb, added_time_ids, Bl, Tl, base=c_un, half=half, par=fpar)
            return unet.forward_tokens(x_loc, c_un, h, w, down, mid)
        if fpar is not None:
            return self._denoise_forward_sharded(adapters, masks, x_loc, t, emb, added_time_ids, Bl, Tl, half, fpar, h, w, c_un)
        cur = torch.cuda.current_stream(self.device)
        if self._adapter_stream is None:
            self._adapter_stream = torch.cuda.Stream(device=self.device)
        side = self._adapter_stream
        side.wait_stream(cur)                                   # the model input (and everything before it) is ready
        with torch.cuda.stream(side):
            down, mid = trunks()
        unet.make_ctx(t, emb, added_time_ids, Bl, Tl, base=c_un, half=half, par=fpar)
        enc = unet.encode_tokens(x_loc, c_un, h, w)
        cur.wait_stream(side)
        for r in list(down) + [mid]:                            # allocated on the side stream, consumed (and freed) on this one
            r.record_stream(cur)
        if not (self.split_decoder and Bl == 2 and half is None):
            return unet.decode_tokens(enc, c_un, down, mid)
        # The UNet's decoder half has nothing to run beside, so its two CFG halves go down the two streams (rows [0, rows) and
        # [rows, 2 rows) of every token tensor; the per-half contexts are the ones a rank of the 2-way CFG layout uses): one
        # half's GroupNorm / attention / epilogue phases overlap the other's MFMA phases.  Half-size launches choose their
        # tiles for themselves, so this order is deterministic but not bit-identical to the single-stream one.
        if c_un.halves is None:
            c_un.halves = [Ctx(1, Tl), Ctx(1, Tl)]
        sample, skips, counts, Hm, Wm = enc
        outs = [None, None]
        side.wait_stream(cur)                                   # the encoder's outputs are ready
        # Lifetime: the second half READS rows of `sample`, every skip and every residual on `side` while this function's
        # references may be dropped on `cur` (decode_tokens pops the skip views): tell the allocator about the second reader, so
        # that none of these blocks can be handed out again before `side` is done with them, whatever decode_tokens keeps alive
        for tt in [sample, mid] + list(skips) + list(down):
            tt.record_stream(side)
        for hf, st in enumerate((cur, side)):
            with torch.cuda.stream(st):
                ch = c_un.halves[hf]
                unet.make_ctx(t[hf:hf + 1] if torch.is_tensor(t) else t, emb, added_time_ids, 1, Tl, base=ch, half=hf, par=None)

                def rows_of(tt, hf=hf):
                    n = tt.shape[0] // 2
                    return tt[hf * n:(hf + 1) * n]
                outs[hf] = unet.decode_tokens((rows_of(sample), [rows_of(k) for k in skips], counts, Hm, Wm), ch,
                                              [rows_of(r) for r in down], rows_of(mid))
        cur.wait_stream(side)
        outs[1].record_stream(cur)
        return torch.cat(outs, 0)

    def _denoise_forward_sharded(self, adapters, masks, x_loc, t, emb, added_time_ids, Bl, Tl, half, fpar, h, w, c_un):
        """Frame-sharded ranks: trunk(s) and UNet encoder still overlap on two HIP streams, but both issue collectives, and all
        ranks must issue them in one order.  ONE host thread therefore enqueues both networks layer by layer in lockstep
        (blocks.run_lockstep: a trunk layer on the second stream, the matching encoder layer on the caller's, ...): the order
        is the program order on every rank, each network's wait for its GroupNorm partials / halo frames / token gather is a
        stream wait covered by the other network's kernels, and the host never blocks (a second host thread under a turn
        token gave the same order at 2.3 x the host time per step, profiles/r04_shard_proxy.log).  The decoder half runs on the
        caller's stream alone."""
        import contextlib
        from .blocks import run_lockstep
        unet, dev = self.unet, self.device
        cur = torch.cuda.current_stream(dev)
        if self._adapter_stream is None:
            self._adapter_stream = torch.cuda.Stream(device=dev)
        side = self._adapter_stream
        side.wait_stream(cur)

        def trunk_layers():
            res = []
            for net, ctx, cond, scale in adapters:
                net.make_ctx(t, emb, added_time_ids, Bl, Tl, base=ctx, half=half, par=fpar)
                res.append((yield from net.forward_layers(x_loc, ctx, h, w, cond, scale)))
            down, mid = res[0]
            if len(res) == 2:
                down, mid = _blend_residuals(down, mid, res[1][0], res[1][1], masks, Bl * Tl)
            return down, mid

        def encoder_layers():
            unet.make_ctx(t, emb, added_time_ids, Bl, Tl, base=c_un, half=half, par=fpar)
            return (yield from unet.encode_layers(x_loc, c_un, h, w))

        @contextlib.contextmanager
        def on_side():
            fpar.lane = 0
            with torch.cuda.stream(side):
                yield

        @contextlib.contextmanager
        def on_cur():
            fpar.lane = 1
            yield
        try:
            (down, mid), enc = run_lockstep([trunk_layers(), encoder_layers()], [on_side, on_cur])
        finally:
            fpar.lane = 0
        cur.wait_stream(side)
        for r in list(down) + [mid]:
            r.record_stream(cur)
        return unet.decode_tokens(enc, c_un, down, mid)

    # ---- pieces of the reference __call__ ---------------------------------------------------------------------------------
    def check_inputs(self, image, height, width):                      # pipeline.py:222-234
        if image is not None and not torch.is_tensor(image) and not isinstance(image, list) and not hasattr(image, "convert"):
            raise ValueError("`image` has to be of type `torch.FloatTensor` or `PIL.Image.Image` or "
                             f"`List[PIL.Image.Image]` but is {type(image)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")

    def _check_call(self, batch_size, num_videos_per_prompt, max_guidance_scale, frames_per_forward):
        """The reference repeats the image embeddings / latents by ``num_videos_per_prompt`` (pipeline.py:130-131, :162) and draws
        ``batch_size * num_videos_per_prompt`` latents (:377-387), but doubles ``controlnet_condition`` / ``controlnet_flow`` only for
        CFG (:392-396) and builds ``added_time_ids`` for ``batch_size`` (:430-440): with anything but 1 x 1 its own
        ``FlowControlNet.forward`` / ``add_embedding`` reshape fail on the batch mismatch, and every entry point calls it with
        1 x 1.  One clip per call therefore IS the reference's behaviour; several videos = several calls with different
        generators / latents."""
        if batch_size != 1 or num_videos_per_prompt != 1:
            raise ValueError("one clip per call (batch_size = num_videos_per_prompt = 1, the only combination the reference's "
                             "own forward accepts: its controlnet_condition / added_time_ids are not repeated)")
        if not max_guidance_scale > 1.0:
            raise ValueError("the reference pipeline is only well-defined with classifier-free guidance on")
        if frames_per_forward > MAX_TEMPORAL_FRAMES:
            raise ValueError(f"{frames_per_forward} frames per forward pass: the temporal attention kernel handles at most "
                             f"{MAX_TEMPORAL_FRAMES} (use KeypointFlowControlNetPipeline's window loop for long clips)")

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
        precomputed through the ``image_embeddings`` / ``image_latents`` extensions.  As in the reference the CLIP branch
        sees the image at its OWN size (pipeline.py:118-122: straight into the 224 x 224 antialiased resize), the VAE branch
        the ``preprocess(image, height, width)`` result (:338)."""
        dev = self.device
        if (image_embeddings is None or image_latents is None) and image is None:
            raise ValueError("pass `image`, or both image_embeddings=... and image_latents=...")
        if image_embeddings is None:
            image_embeddings = self._encode_image(frontend.image_to_01(image, None, None, dev))
        if image_latents is None:
            image_latents = self._encode_vae_image(frontend.image_to_01(image, height, width, dev), noise_aug_strength, generator)
        emb = image_embeddings.to(dev, torch.float32).reshape(-1, 1, image_embeddings.shape[-1])
        if emb.shape[0] == 1:                                             # :133-139 uncond = zeros
            emb = torch.cat([torch.zeros_like(emb), emb])
        il = image_latents.to(dev, torch.float32)
        if il.shape[0] == 1:                                              # :153-159
            il = torch.cat([torch.zeros_like(il), il])
        return emb, il.contiguous()

    def _condition_image(self, controlnet_condition, height, width):
        """``self.image_processor.preprocess(controlnet_condition, height=height, width=width)`` (pipeline.py:391): PIL /
        numpy / tensor, resized, [0, 1] -> [-1, 1] unless the tensor already holds negative values"""
        if controlnet_condition is None:
            raise ValueError("`controlnet_condition` is required")
        return frontend.preprocess(controlnet_condition, height, width, self.device)

    def _round(self, lat):
        if not self.round_latents_to_fp16:
            return lat
        return ops.cast_f16_to_f32(ops.cast_f32_to_f16(lat.contiguous())).reshape(lat.shape)

    def _clip_loop(self, lat, il, emb, added_time_ids, adapters, c_un, masks, sh, h, w, g0, g1, callback):
        """The denoise loop of one clip (pipeline.py:447-511) on this rank's CFG half / frames: ``lat`` fp32 [Tl,4,h,w] is stepped
        in place through every timestep of the scheduler and returned.

        Everything a step takes from the scheduler (sigma, sigma_next, the network's timestep, 1 / sqrt(sigma^2 + 1)) is uploaded
        ONCE per clip as a device table (scheduler.step_table) and the kernels read their row of it: no host scalar, no H2D copy
        inside a step.  That makes a step capturable: with ``graph_steps`` the first step runs eagerly (allocations, one-time
        kernel set-up, per-clip caches), the second is captured in a hipGraph on the loop's own stream -- its first node copies
        row ``counter`` of the table into a fixed ``cur`` row and increments the counter, every other node reads ``cur`` -- and
        steps 1 .. n-1 are replays: the host issues one graph launch per step instead of ~1 400 kernel launches."""
        sch, dev, unet = self.scheduler, self.device, self.unet
        Tl, Bl, half, fpar = sh.Tl, sh.Bl, sh.half, sh.fpar
        rows = Tl * h * w
        in_ld = max([unet.in_ld] + [a[0].in_ld for a in adapters])
        x_in = torch.zeros((2 * rows, in_ld), dtype=torch.float16, device=dev)
        x_loc = x_in if Bl == 2 else x_in[half * rows:(half + 1) * rows]
        timesteps = sch.timesteps
        n = len(timesteps)
        self._num_timesteps = n
        tab = torch.from_numpy(sch.step_table()).to(dev)

        def step(scal):
            ops.prepare_model_input_dev(lat, il, x_in, scal)
            noise = self._denoise_forward(x_loc, scal[2:2 + Bl], emb, added_time_ids, Bl, Tl, half, fpar, h, w, adapters, c_un, masks)
            if Bl == 1:
                noise = sh.par.gather_cfg(noise)                          # both halves of this frame shard
            ops.cfg_euler_step_dev_(lat, noise, scal, g0, g1)
            if self.round_latents_to_fp16:
                ops.cast_f16_to_f32(ops.cast_f32_to_f16(lat), out=lat)

        if not (self.graph_steps and callback is None and sh.par is None and ops.TIMER is None and n >= 3):
            for i, t in enumerate(timesteps):
                step(tab[i])
                if callback is not None:
                    new = self._callback(callback, i, t, lat, (1, Tl, 4, h, w))
                    if new is not lat:
                        lat.copy_(new)
            return lat
        caller = torch.cuda.current_stream(dev)
        if self._loop_stream is None:
            self._loop_stream = torch.cuda.Stream(device=dev)
        gs = self._loop_stream
        gs.wait_stream(caller)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        cur = torch.empty(ops.STEP_SCALARS, dtype=torch.float32, device=dev)

        def body():
            ops.step_select(tab, counter, cur)
            step(cur)
        with torch.cuda.stream(gs):
            body()                                                        # step 0, eager
            graph = torch.cuda.CUDAGraph()
            graph.capture_begin()
            try:
                body()                                                    # (recorded, not executed)
            finally:
                graph.capture_end()
            for _ in range(1, n):
                graph.replay()
        caller.wait_stream(gs)
        lat.record_stream(gs)
        self._last_graph = graph                                          # keeps the graph's memory pool until the next clip
        return lat

    def _callback(self, cb, i, t, lat, shape):
        """callback_on_step_end(self, i, t, {"latents": ...}) may return replacement latents (pipeline.py:503-508)"""
        if cb is None:
            return lat
        out = cb(self, i, t, {"latents": lat.reshape(shape)})
        if out and "latents" in out:
            lat = out["latents"].to(self.device, torch.float32).reshape(lat.shape).contiguous()
        return lat

    def _decode(self, latents_out, T, decode_chunk_size, output_type, sh, stream_chunks=None):
        """decode_latents + tensor2vid (pipeline.py:513-518).  With several ranks the independent VAE chunks (:204-213) are
        dealt round-robin (or as ``sh.owner`` says): the result is then the list of (first_frame, frames) chunks this rank
        decoded.  stream_chunks: {chunk index: raw fp32 [n,3,H,W]} already decoded on a side stream (Keypoint loop)."""
        if output_type == "latent":
            return latents_out
        if sh.world == 1:
            if stream_chunks:
                parts = []
                for ci, s0 in enumerate(range(0, T, decode_chunk_size)):
                    z = latents_out[0, s0:s0 + decode_chunk_size]
                    parts.append(stream_chunks[ci] if ci in stream_chunks else
                                 self.vae.decode(z, num_frames=z.shape[0], _prescale=1.0 / self.vae.config.scaling_factor))
                fr = torch.cat(parts, 0)
                frames = fr.reshape(-1, T, *fr.shape[1:]).permute(0, 2, 1, 3, 4).float()
            else:
                frames = decode_latents(self.vae, latents_out, T, decode_chunk_size)   # fp32 [1,3,T,H,W]
            return frames if output_type == "raw" else tensor2vid(frames, None, output_type=output_type)
        frames = []
        sf = 1.0 / self.vae.config.scaling_factor
        owner = getattr(sh, "owner", None)                  # Keypoint loop: chunks decoded early belong to a stated rank
        for ci, s0 in enumerate(range(0, T, decode_chunk_size)):
            if (owner[ci] if owner else ci % sh.world) == sh.rank:
                z = latents_out[0, s0:s0 + decode_chunk_size]
                fr = stream_chunks[ci] if (stream_chunks and ci in stream_chunks) else \
                    self.vae.decode(z, num_frames=z.shape[0], _prescale=sf)            # fp32 [n,3,H,W]
                if output_type != "raw":
                    fr = tensor2vid(fr.permute(1, 0, 2, 3).unsqueeze(0), None, output_type=output_type)[0]
                frames.append((s0, fr))
        return frames

    # ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, height: int = 576,
                 width: int = 1024, num_frames: Optional[int] = None, num_inference_steps: int = 25,
                 min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7,
                 motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1,
                 generator=None, latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], return_dict: bool = True,
                 controlnet_cond_scale=1.0, batch_size=1, *, image_embeddings=None, image_latents=None):
        unet, cn, sch, dev = self.unet, self.controlnet, self.scheduler, self.device
        num_frames = num_frames if num_frames is not None else unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        self.check_inputs(image, height, width)
        self._check_call(batch_size, num_videos_per_prompt, max_guidance_scale, num_frames)
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
        cond = self._condition_image(controlnet_condition, height, width)
        flow = controlnet_flow.to(dev, torch.float32)
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)   # :430-440

        sh = _Shard(self.parallel, T)
        f0, f1, Tl, Bl, half, fpar = sh.f0, sh.f1, sh.Tl, sh.Bl, sh.half, sh.fpar
        warped = cn.prepare_condition(cond[:1], flow[:1], frames=(f0, f1))
        lat = lat[f0:f1].contiguous()
        gspan = (max_guidance_scale - min_guidance_scale) / max(T - 1, 1)  # per-frame guidance is linear in the frame
        g0, g1 = min_guidance_scale + gspan * f0, min_guidance_scale + gspan * (f1 - 1)

        c_cn, c_un = Ctx(Bl, Tl), Ctx(Bl, Tl)                             # hold the per-clip invariant caches
        lat = self._clip_loop(lat, il, emb, added_time_ids, [(cn, c_cn, warped, controlnet_cond_scale)], c_un, None, sh, h, w,
                              g0, g1, callback_on_step_end)                # :447-511

        if fpar is not None:                                              # reassemble the clip's latents on every rank
            lat = fpar.gather_frames(lat.reshape(Tl, 4 * h * w), 1).reshape(T, 4, h, w)
        frames = self._decode(lat.reshape(1, T, 4, h, w), T, decode_chunk_size, output_type, sh)
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)


# =========================================================================================================
# Hybrid: face (landmark) adapter + drag (trajectory) adapter, residuals blended by the user mask
# (MOFA-Video-Hybrid/pipeline/pipeline.py:293-320 signature, :443-507 loop, :479-489 blend)
# =========================================================================================================
def _resized_masks(mask, height, width, h, w, dev):
    """user mask [1,1,H,W] nearest-resized to the four residual resolutions (:481, :488); timestep-invariant"""
    m = mask.to(dev, torch.float32).reshape(1, height, width)
    masks, hh, ww = {}, h, w
    for _ in range(4):
        masks[hh * ww] = ops.resize_nearest_f32(m, hh, ww).reshape(-1).contiguous()
        hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
    return masks


def _blend_residuals(df, mf, dd, md, masks, nframes):
    """face * m + drag * (1 - m) on all 12 + 1 residuals (:479-489)"""
    down = []
    for a, b in zip(df, dd):
        hw = a.shape[0] // nframes
        down.append(ops.mask_blend(a, b, masks[hw], hw))
    hw = mf.shape[0] // nframes
    return down, ops.mask_blend(mf, md, masks[hw], hw)


class HybridFlowControlNetPipeline(FlowControlNetPipeline):
    def __init__(self, vae=None, image_encoder=None, unet=None, face_controlnet=None, drag_controlnet=None,
                 scheduler=None, feature_extractor=None, parallel=None, round_latents_to_fp16=False):
        super().__init__(vae, image_encoder, unet, face_controlnet, scheduler, feature_extractor, parallel=parallel,
                         round_latents_to_fp16=round_latents_to_fp16)
        self.face_controlnet, self.drag_controlnet = face_controlnet, drag_controlnet

    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, landmarks=None, drag_flow=None,
                 mask=None, height: int = 576, width: int = 1024, num_frames: Optional[int] = None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1, generator=None,
                 latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 return_dict: bool = True, ctrl_scale_traj=1.0, ctrl_scale_ldmk=1.0, batch_size=1, *,
                 image_embeddings=None, image_latents=None):
        unet, face, drag, sch, dev = self.unet, self.face_controlnet, self.drag_controlnet, self.scheduler, self.device
        T = num_frames if num_frames is not None else unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else T
        self.check_inputs(image, height, width)
        self._check_call(batch_size, num_videos_per_prompt, max_guidance_scale, T)
        h, w = height // 8, width // 8
        emb, il = self._conditioning(image, image_embeddings, image_latents, height, width, noise_aug_strength, generator)
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps
        lat = self.prepare_latents(1, T, unet.config.in_channels, height, width, generator, latents).reshape(T, 4, h, w).contiguous()
        cond = self._condition_image(controlnet_condition, height, width)
        # frame sharding exactly as in FlowControlNetPipeline.__call__ (2-way CFG x frame shards)
        sh = _Shard(self.parallel, T)
        f0, f1, Tl, Bl, half, fpar = sh.f0, sh.f1, sh.Tl, sh.Bl, sh.half, sh.fpar
        cf = face.prepare_condition(cond[:1], controlnet_flow.to(dev, torch.float32)[:1], landmarks[:1], frames=(f0, f1))
        cd = drag.prepare_condition(cond[:1], drag_flow.to(dev, torch.float32)[:1], frames=(f0, f1))
        lat = lat[f0:f1].contiguous()
        gspan = (max_guidance_scale - min_guidance_scale) / max(T - 1, 1)
        g0, g1 = min_guidance_scale + gspan * f0, min_guidance_scale + gspan * (f1 - 1)
        masks = _resized_masks(mask, height, width, h, w, dev)
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
        c_f, c_d, c_u = Ctx(Bl, Tl), Ctx(Bl, Tl), Ctx(Bl, Tl)
        lat = self._clip_loop(lat, il, emb, added_time_ids, [(face, c_f, cf, ctrl_scale_ldmk), (drag, c_d, cd, ctrl_scale_traj)],
                              c_u, masks, sh, h, w, g0, g1, callback_on_step_end)
        if fpar is not None:
            lat = fpar.gather_frames(lat.reshape(Tl, 4 * h * w), 1).reshape(T, 4, h, w)
        frames = self._decode(lat.reshape(1, T, 4, h, w), T, decode_chunk_size, output_type, sh)
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)


# =========================================================================================================
# Keypoint long video ("periodic sampling"): overlapping temporal windows with frame 0 prepended, one Euler step per
# window, overlap average (MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:289-294 signature,
# :426-429 views, :445-511 loop)
# =========================================================================================================
def _deal_ready_chunks(ready, world, busy_next, idle_cap=3, busy_cap=2):
    """Which rank decodes the VAE chunks that became final before the last round(s) of the last denoise step: a rank with no
    window in the next round takes up to ``idle_cap`` of them (a chunk decode is about a third of a window step), a rank that
    is stepping a window up to ``busy_cap`` on its second stream (a window finalises 1.5 chunks per round at stride 12 and
    chunks of 8); the rest wait for the next round.  Pure function of its arguments: every rank computes the same table."""
    idle = [r for r in range(world) if r not in busy_next]
    slots = [r for k in range(idle_cap) for r in idle] + [r for k in range(busy_cap) for r in busy_next]
    return list(zip(ready, slots))


def window_views(num_frames, window_size, stride):
    window_num = (num_frames - window_size) // stride + 1
    views = [(1 + i * stride, i * stride + window_size) for i in range(window_num)]
    return views + [(num_frames - window_size + 1, num_frames)]


class KeypointFlowControlNetPipeline(FlowControlNetPipeline):
    """The window loop of the reference's long-video pipeline.  Two extensions beyond it (new work, BASELINE config 5):
    * hybrid control inside the windows: built with a ``drag_controlnet`` and called with ``drag_flow`` / ``mask``, every
      window runs the landmark adapter AND the trajectory adapter and blends their residuals by the mask, exactly as
      ``HybridFlowControlNetPipeline`` does per clip (Hybrid/pipeline/pipeline.py:479-489);
    * the VAE decode overlaps the last denoise step: in that step the windows finish in order, so every decode chunk whose
      frames are final is decoded on a second HIP stream while the remaining windows are still being stepped.  With several
      ranks the same holds per round (``parallel.WindowParallel``: a round = one window per rank; ranks without a window in
      the next round take the finished chunks first) or per window (``parallel.FrameParallel``: every window on all ranks),
      and the chunks not decoded early are dealt evenly after the loop."""

    def __init__(self, vae=None, image_encoder=None, unet=None, controlnet=None, scheduler=None, feature_extractor=None,
                 parallel=None, round_latents_to_fp16=False, drag_controlnet=None, overlap_decode=True):
        super().__init__(vae, image_encoder, unet, controlnet, scheduler, feature_extractor, parallel=parallel,
                         round_latents_to_fp16=round_latents_to_fp16)
        self.drag_controlnet = drag_controlnet
        self.overlap_decode = overlap_decode

    def _single_window_sharded(self, lat, il, emb, cond, flow, dflow, landmarks, mask, timesteps, h, w, height, width, gmin, gmax,
                               cn_scale, traj_scale, callback):
        """the window loop for ONE window == the clip, on this rank's CFG half / frame shard (see __call__)"""
        unet, cn, drag, sch, dev = self.unet, self.controlnet, self.drag_controlnet, self.scheduler, self.device
        T = lat.shape[0]
        sh = _Shard(self.parallel, T)
        f0, f1, Tl, Bl, half, fpar = sh.f0, sh.f1, sh.Tl, sh.Bl, sh.half, sh.fpar
        hybrid = dflow is not None
        # window = frame 0 + frames 1 .. T-1 with the flows of frames 1 .. T-1 (svdxt_pipeline_ctrlnet_loop.py:470-476)
        cf = cn.prepare_condition(cond[:1], flow[:1, 0:T - 1], landmarks[:, 0:T], frames=(f0, f1))
        cd = drag.prepare_condition(cond[:1], dflow[:1, 0:T - 1], frames=(f0, f1)) if hybrid else None
        masks = _resized_masks(mask, height, width, h, w, dev) if hybrid else None
        lat = lat[f0:f1].contiguous()
        gspan = (gmax - gmin) / max(T - 1, 1)
        g0, g1 = gmin + gspan * f0, gmin + gspan * (f1 - 1)
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
        c_f, c_d, c_u = Ctx(Bl, Tl), Ctx(Bl, Tl), Ctx(Bl, Tl)
        adapters = [(cn, c_f, cf, cn_scale)] + ([(drag, c_d, cd, traj_scale)] if hybrid else [])
        lat = self._clip_loop(lat, il, emb, added_time_ids, adapters, c_u, masks, sh, h, w, g0, g1, callback)
        if fpar is not None:
            lat = fpar.gather_frames(lat.reshape(Tl, 4 * h * w), 1).reshape(T, 4, h, w)
        return lat

    @torch.no_grad()
    def __call__(self, image=None, controlnet_condition=None, controlnet_flow=None, landmarks=None, window_size: int = 25,
                 stride: int = 12, height: int = 576, width: int = 1024, num_frames: Optional[int] = None,
                 num_inference_steps: int = 25, min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0,
                 fps: int = 7, motion_bucket_id: int = 127, noise_aug_strength: float = 0.02,
                 decode_chunk_size: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1, generator=None,
                 latents: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 return_dict: bool = True, controlnet_cond_scale=1.0, batch_size=1, *, image_embeddings=None,
                 image_latents=None, drag_flow=None, mask=None, ctrl_scale_traj=1.0):
        unet, cn, drag, sch, dev = self.unet, self.controlnet, self.drag_controlnet, self.scheduler, self.device
        N = num_frames if num_frames is not None else unet.config.num_frames
        Tw = window_size
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else N
        self.check_inputs(image, height, width)
        self._check_call(batch_size, num_videos_per_prompt, max_guidance_scale, Tw)
        if N < Tw:
            raise ValueError(f"num_frames ({N}) must be at least window_size ({Tw})")
        if not 0 < stride < Tw:
            raise ValueError(f"stride ({stride}) must lie in (0, window_size = {Tw}): consecutive windows have to overlap")
        hybrid = drag_flow is not None
        if hybrid and (drag is None or mask is None):
            raise ValueError("hybrid control needs a drag_controlnet (constructor) and a mask")
        h, w = height // 8, width // 8
        emb, il = self._conditioning(image, image_embeddings, image_latents, height, width, noise_aug_strength, generator)
        sch.set_timesteps(num_inference_steps)
        timesteps = sch.timesteps
        lat = self.prepare_latents(1, N, unet.config.in_channels, height, width, generator, latents).reshape(N, 4, h, w).contiguous()
        cond = self._condition_image(controlnet_condition, height, width)
        flow = controlnet_flow.to(dev, torch.float32)
        dflow = drag_flow.to(dev, torch.float32) if hybrid else None
        views = window_views(N, Tw, stride)
        from .parallel import FrameParallel, GroupedWindowParallel
        grouped = self.parallel if isinstance(self.parallel, GroupedWindowParallel) else None
        framepar = self.parallel if isinstance(self.parallel, FrameParallel) else (grouped.frame if grouped is not None else None)
        if framepar is not None and grouped is None and len(set(views)) == 1 and N == Tw:
            # ONE window that is the whole clip (N == window_size: every view is frames 1 .. N-1 behind frame 0): the loop
            # degenerates to the plain denoise loop -- value = k * stepped window, count = k -- so the clip is frame-sharded
            # exactly like FlowControlNetPipeline / HybridFlowControlNetPipeline (2-way CFG x frame shards) and the latents
            # stay sharded between the steps.
            lat = self._single_window_sharded(lat, il, emb, cond, flow, dflow, landmarks, mask, timesteps, h, w, height, width,
                                              min_guidance_scale, max_guidance_scale, controlnet_cond_scale, ctrl_scale_traj,
                                              callback_on_step_end)
            sh = _Shard(self.parallel, N)
            frames = self._decode(lat.reshape(1, N, 4, h, w), N, decode_chunk_size, output_type, sh)
            if not return_dict:
                return frames, controlnet_flow
            return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)
        # Several windows, four ways to spread them (self.parallel; parallel.window_layout_costs prices them):
        #   None            every window on this GPU, one after the other;
        #   WindowParallel  the distinct windows of a step dealt to the ranks, one all-gather per round;
        #   FrameParallel   (Layout of window_size frames) every window on ALL ranks, 2-way CFG x frame shards inside the
        #                   window, the stepped window gathered before the overlap average -- the layout for more ranks than
        #                   windows (on 8 ranks and 7 windows WindowParallel's single round is the faster of the two);
        #   GroupedWindowParallel  G groups of g ranks: windows dealt to the groups, frame-parallel inside a group.
        # In all of them every rank holds all N latent frames and applies the same overlap average in view order.
        sh_w = _Shard(framepar, Tw)
        f0, f1, Tl, Bl, half, fpar = sh_w.f0, sh_w.f1, sh_w.Tl, sh_w.Bl, sh_w.half, sh_w.fpar
        # adapter state per DISTINCT window is timestep-invariant: computed once per clip (the reference recomputes it
        # every step; its last view often repeats the previous one -- SURVEY 3.5 -- and is computed once here)
        conds, dconds = {}, {}
        fr = dict(frames=(f0, f1)) if framepar is not None else {}
        for (t0, t1) in views:
            if (t0, t1) not in conds:
                lm = torch.cat([landmarks[:, 0:1], landmarks[:, t0:t1]], dim=1)
                conds[(t0, t1)] = cn.prepare_condition(cond[:1], flow[:1, t0 - 1:t1 - 1], lm, **fr)
                if hybrid:
                    dconds[(t0, t1)] = drag.prepare_condition(cond[:1], dflow[:1, t0 - 1:t1 - 1], **fr)
        masks = _resized_masks(mask, height, width, h, w, dev) if hybrid else None
        added_time_ids = torch.tensor([[6.0, 128.0, 0.02]] * 2, dtype=torch.float32, device=dev)
        ctxs = {v: (Ctx(Bl, Tl), Ctx(Bl, Tl), Ctx(Bl, Tl)) for v in conds}
        distinct = list(conds)                                                     # distinct windows, in view order
        wpar = grouped if grouped is not None else (self.parallel if framepar is None else None)   # deals windows, or None
        if wpar is not None:
            world, rank = wpar.world, wpar.rank                                    # (global: VAE chunks go to all ranks)
        else:
            world, rank = (sh_w.world, sh_w.rank) if framepar is not None else (1, 0)
        nslots, slot = (len(wpar.rounds(list(conds))[0]), wpar.slot) if wpar is not None else (1, 0)
        rows = Tl * h * w
        x_in = torch.zeros((2 * rows, unet.in_ld), dtype=torch.float16, device=dev)
        x_loc = x_in if Bl == 2 else x_in[half * rows:(half + 1) * rows]
        g0, g1 = min_guidance_scale, max_guidance_scale
        if framepar is not None:                                                   # guidance of this rank's frames of a window
            gspan = (max_guidance_scale - min_guidance_scale) / max(Tw - 1, 1)
            g0, g1 = min_guidance_scale + gspan * f0, min_guidance_scale + gspan * (f1 - 1)
        value = torch.empty_like(lat)
        # the views that cover each frame, and the last of them in processing order: a frame is final once that one is merged
        last_view_of = [max(idx for idx, (t0, t1) in enumerate(views) if (0 if idx == 0 else t0) <= f < t1) for f in range(N)]
        nchunks = -(-N // decode_chunk_size)
        rounds = wpar.rounds(distinct) if wpar is not None else [[k] for k in distinct]
        # the decode of finished frames overlaps the rest of the LAST step on a second HIP stream wherever that step has more
        # than one round; a chunk belongs to ``owner[ci]`` (the same table on every rank)
        side = torch.cuda.Stream(device=dev) if (self.overlap_decode and output_type != "latent" and len(rounds) > 1) else None
        stream_chunks, owner = {}, {}
        self._num_timesteps = len(timesteps)
        for i, t in enumerate(timesteps):
            sigma, sigma_next = sch.sigma_pair(i)
            overlap = i == len(timesteps) - 1 and side is not None
            count = [0] * N
            touched = [False] * N
            # every window of a step reads the latents of the PREVIOUS step; when frames are finalised while later windows
            # of the (last) step still run, those windows read a snapshot
            lat_in = lat.clone() if overlap else lat

            def step_window(t0, t1):
                lw = torch.cat([lat_in[0:1], lat_in[t0:t1]], dim=0)                   # frame 0 + window frames
                lw = lw[f0:f1].contiguous()                                           # (this rank's frames of the window)
                ops.prepare_model_input(lw, il, x_in, sigma)
                c_cn, c_dr, c_un = ctxs[(t0, t1)]
                adapters = [(cn, c_cn, conds[(t0, t1)], controlnet_cond_scale)]
                if hybrid:
                    adapters.append((drag, c_dr, dconds[(t0, t1)], ctrl_scale_traj))
                noise = self._denoise_forward(x_loc, t, emb, added_time_ids, Bl, Tl, half, fpar, h, w, adapters, c_un, masks)
                if Bl == 1:
                    noise = framepar.gather_cfg(noise)
                ops.cfg_euler_step_(lw, noise, sigma, sigma_next, g0, g1)
                if fpar is not None:
                    lw = fpar.gather_frames(lw.reshape(Tl, 4 * h * w), 1).reshape(Tw, 4, h, w)
                return lw

            def merge(idx, lw):
                """value[0:t1] += lw (first view) / value[t0:t1] += lw[1:] (others)   (:502-507)"""
                t0, t1 = views[idx]
                if idx == 0 and t0 != 1:
                    raise ValueError("the first window must start at frame 1")
                dst0, src0 = (0, 0) if idx == 0 else (t0, 1)
                for k in range(t1 - dst0):
                    f = dst0 + k
                    ops.axpby_f32_(lw[src0 + k].reshape(-1), value[f].reshape(-1), 1.0, 1.0 if touched[f] else 0.0)
                    touched[f] = True
                    count[f] += 1

            def finalize(frames):
                """latents = where(count > 0, value / count, value) (:511; value is 0 where no window landed)"""
                for f in frames:
                    if count[f]:
                        ops.axpby_f32_(value[f].reshape(-1), lat[f].reshape(-1), 1.0 / count[f], 0.0)
                    else:
                        lat[f].zero_()

            done = {}
            merged, frontier = 0, 0                               # views [0, merged) are merged, frames [0, frontier) final
            for ri, rnd in enumerate(rounds):
                if wpar is None:
                    done[rnd[0]] = step_window(*rnd[0])
                else:                                             # window-parallel: one window per rank and round
                    mine = rnd[slot]
                    lw = step_window(*mine) if mine is not None else torch.zeros((Tw,) + tuple(lat.shape[1:]),
                                                                                dtype=lat.dtype, device=dev)
                    for key, got in zip(rnd, wpar.gather(lw)):
                        if key is not None:
                            done[key] = got
                if not overlap or ri == len(rounds) - 1:
                    continue
                # views are merged in view order as soon as their window is stepped (the same summation order as after the
                # loop); every frame whose last covering view is merged can be averaged now, and whole decode chunks below
                # the frontier go to the second stream while the next round runs
                while merged < len(views) and views[merged] in done:
                    merge(merged, done[views[merged]])
                    merged += 1
                newf = frontier
                while newf < N and last_view_of[newf] < merged:
                    newf += 1
                finalize(range(frontier, newf))
                frontier = newf
                ready = [ci for ci in range(nchunks) if ci not in owner and min((ci + 1) * decode_chunk_size, N) <= frontier]
                busy_next = [r for r in range(world) if wpar is None or rounds[ri + 1][r * nslots // world] is not None]
                for ci, r in _deal_ready_chunks(ready, world, busy_next):
                    owner[ci] = r
                mine = [ci for ci in ready if owner.get(ci) == rank]
                if mine:
                    lat_r = self._round(lat)
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        lat_r.record_stream(side)                 # (lat_r may be a temporary of the main stream: keep its
                        for ci in mine:                           #  block out of the allocator until the decode ran)
                            s0 = ci * decode_chunk_size
                            z = lat_r[s0:min(s0 + decode_chunk_size, N)]
                            stream_chunks[ci] = self.vae.decode(z, num_frames=z.shape[0],
                                                                _prescale=1.0 / self.vae.config.scaling_factor)
            while merged < len(views):
                merge(merged, done[views[merged]])
                merged += 1
            finalize(range(frontier, N))
            lat = self._round(lat)
            lat = self._callback(callback_on_step_end, i, t, lat, (1, N, 4, h, w))
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            if callback_on_step_end is not None:
                stream_chunks, owner = {}, {}             # a callback may have replaced the latents after the last step
        rest = [ci for ci in range(nchunks) if ci not in owner]   # the chunks still to decode: dealt evenly
        for j, ci in enumerate(rest):
            owner[ci] = j % world
        sh = _Shard(None, N)
        sh.world, sh.rank = world, rank
        sh.owner = owner
        frames = self._decode(lat.reshape(1, N, 4, h, w), N, decode_chunk_size, output_type, sh, stream_chunks)
        if not return_dict:
            return frames, controlnet_flow
        return FlowControlNetPipelineOutput(frames=frames, controlnet_flow=controlnet_flow)
