"""Oracle (test infrastructure): the denoise loop + decode of
``FlowControlNetPipeline.__call__`` (MOFA-Video-Traj/pipeline/pipeline.py:283-527)
restated for CPU fp32 on *pre-encoded* inputs (image embeddings / image latents are
produced before the hot path by CLIP / VAE-encode).
"""
import torch

from .vae import decode_latents


def make_added_time_ids(dtype=torch.float32, device=None):
    """pipeline.py:430-440 -- overwritten to fps=6, motion_bucket_id=128, noise_aug=0.02, x2 for CFG."""
    ids = torch.tensor([[6, 128, 0.02]], dtype=dtype, device=device)
    return torch.cat([ids] * 2)


@torch.no_grad()
def denoise(unet, controlnet, scheduler, latents, image_latents, image_embeddings, controlnet_condition,
            controlnet_flow, num_inference_steps=25, min_guidance_scale=1.0, max_guidance_scale=3.0,
            controlnet_cond_scale=1.0, return_trace=False):
    """latents [1,T,4,h,w] ~ N(0,1) (unscaled); image_latents [2,4,h,w] (uncond zeros first);
    image_embeddings [2,1,1024] (uncond zeros first); controlnet_condition [1,3,H,W];
    controlnet_flow [1,T-1,2,H,W].  Returns final latents [1,T,4,h,w]."""
    num_frames = latents.shape[1]
    scheduler.set_timesteps(num_inference_steps)                                  # :372
    timesteps = scheduler.timesteps
    latents = latents * scheduler.init_noise_sigma                                # :272
    image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)     # :356
    controlnet_condition = torch.cat([controlnet_condition] * 2)                  # :393
    controlnet_flow = torch.cat([controlnet_flow] * 2)                            # :396
    guidance_scale = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0)
    guidance_scale = guidance_scale.to(latents.device, latents.dtype)[(...,) + (None,) * 3]       # :423-426
    added_time_ids = make_added_time_ids(latents.dtype, latents.device)
    trace = []
    for t in timesteps:                                                           # :447-511
        latent_model_input = torch.cat([latents] * 2)
        latent_model_input = scheduler.scale_model_input(latent_model_input, t)
        latent_model_input = torch.cat([latent_model_input, image_latents], dim=2)
        down_res, mid_res, _, _ = controlnet(
            latent_model_input, t, encoder_hidden_states=image_embeddings, controlnet_cond=controlnet_condition,
            controlnet_flow=controlnet_flow, added_time_ids=added_time_ids,
            conditioning_scale=controlnet_cond_scale, guess_mode=False, return_dict=False)
        noise_pred = unet(latent_model_input, t, encoder_hidden_states=image_embeddings,
                          down_block_additional_residuals=down_res, mid_block_additional_residual=mid_res,
                          added_time_ids=added_time_ids, return_dict=False)[0]
        uncond, cond = noise_pred.chunk(2)
        noise_pred = uncond + guidance_scale * (cond - uncond)                   # :495-497
        latents = scheduler.step(noise_pred, t, latents)                          # :500
        if return_trace:
            trace.append(latents.clone())
    return (latents, trace) if return_trace else latents


@torch.no_grad()
def denoise_and_decode(unet, controlnet, vae, scheduler, *args, decode_chunk_size=8, **kw):
    latents = denoise(unet, controlnet, scheduler, *args, **kw)
    frames = decode_latents(vae, latents, latents.shape[1], decode_chunk_size)   # :517
    return latents, frames


# ---------------------------------------------------------------------------------------------------------
# Hybrid: face (landmark) adapter + drag (trajectory) adapter, residuals blended by a user mask
# (MOFA-Video-Hybrid/pipeline/pipeline.py:443-507; blend :479-489)
# ---------------------------------------------------------------------------------------------------------
@torch.no_grad()
def denoise_hybrid(unet, face_controlnet, drag_controlnet, scheduler, latents, image_latents, image_embeddings,
                   controlnet_condition, controlnet_flow, landmarks, drag_flow, mask, num_inference_steps=25,
                   min_guidance_scale=1.0, max_guidance_scale=3.0, ctrl_scale_traj=1.0, ctrl_scale_ldmk=1.0,
                   return_trace=False):
    """landmarks [1,T,3,H,W]; mask [1,1,H,W] (1 = face adapter, 0 = drag adapter); other args as ``denoise``.
    return_trace: also the latents after every step (test bookkeeping, not in the reference)."""
    import torch.nn.functional as F
    num_frames = latents.shape[1]
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    latents = latents * scheduler.init_noise_sigma
    image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
    controlnet_condition = torch.cat([controlnet_condition] * 2)          # :405-418
    controlnet_flow = torch.cat([controlnet_flow] * 2)
    drag_flow = torch.cat([drag_flow] * 2)
    landmarks = torch.cat([landmarks] * 2)
    guidance_scale = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).unsqueeze(0)
    guidance_scale = guidance_scale.to(latents.device, latents.dtype)[(...,) + (None,) * 3]
    added_time_ids = make_added_time_ids(latents.dtype, latents.device)
    trace = []
    for t in timesteps:
        x = torch.cat([latents] * 2)
        x = scheduler.scale_model_input(x, t)
        x = torch.cat([x, image_latents], dim=2)
        df, mf, _, _ = face_controlnet(x, t, encoder_hidden_states=image_embeddings,
                                       controlnet_cond=controlnet_condition, controlnet_flow=controlnet_flow,
                                       landmarks=landmarks, added_time_ids=added_time_ids,
                                       conditioning_scale=ctrl_scale_ldmk, return_dict=False)
        dd, md, _, _ = drag_controlnet(x, t, encoder_hidden_states=image_embeddings,
                                       controlnet_cond=controlnet_condition, controlnet_flow=drag_flow,
                                       added_time_ids=added_time_ids, conditioning_scale=ctrl_scale_traj,
                                       return_dict=False)
        down = []
        for a, b in zip(df, dd):                                          # :479-485
            m = F.interpolate(mask, a.shape[-2:], mode='nearest')
            down.append(a * m + b * (1 - m))
        m = F.interpolate(mask, mf.shape[-2:], mode='nearest')            # :487-489
        mid = mf * m + md * (1 - m)
        noise_pred = unet(x, t, encoder_hidden_states=image_embeddings, down_block_additional_residuals=down,
                          mid_block_additional_residual=mid, added_time_ids=added_time_ids, return_dict=False)[0]
        u, c = noise_pred.chunk(2)
        noise_pred = u + guidance_scale * (c - u)
        latents = scheduler.step(noise_pred, t, latents)
        if return_trace:
            trace.append(latents.clone())
    return (latents, trace) if return_trace else latents


# ---------------------------------------------------------------------------------------------------------
# Keypoint long video ("periodic sampling"): overlapping temporal windows, frame 0 prepended to every window,
# one Euler step per window, overlap-average (MOFA-Video-Keypoint/pipeline/svdxt_pipeline_ctrlnet_loop.py:426-511)
# ---------------------------------------------------------------------------------------------------------
def window_views(num_frames, window_size, stride):
    """:426-429"""
    window_num = (num_frames - window_size) // stride + 1
    views = [(1 + i * stride, i * stride + window_size) for i in range(window_num)]
    return views + [(num_frames - window_size + 1, num_frames)]


@torch.no_grad()
def denoise_keypoint_loop(unet, controlnet, scheduler, latents, image_latents, image_embeddings, controlnet_condition,
                          controlnet_flow, landmarks, window_size=25, stride=12, num_inference_steps=25,
                          min_guidance_scale=1.0, max_guidance_scale=3.0, controlnet_cond_scale=1.0,
                          drag_controlnet=None, drag_flow=None, mask=None, ctrl_scale_traj=1.0, return_trace=False,
                          reuse_identical_views=False):
    """latents [1,N,4,h,w]; controlnet_flow [1,N-1,2,H,W]; landmarks [1,N,3,H,W]; ``controlnet`` = landmark adapter.
    drag_controlnet / drag_flow / mask: BASELINE config 5's "hybrid control" inside the windows -- no reference file runs
    it; it composes the window loop above with the Hybrid step's residual blend (Hybrid/pipeline/pipeline.py:479-489).
    return_trace: also the merged latents after every step.  reuse_identical_views (test bookkeeping, result-identical):
    the reference's last view often repeats the one before it (:426-429, e.g. N = window_size gives (1,N) twice); every
    window of a step reads the PREVIOUS step's latents and the networks are deterministic functions of their inputs, so a
    repeated view's noise prediction IS the earlier one's -- it is taken from there instead of being recomputed; the Euler
    step, the ``_step_index`` rewind and the overlap average still run once per view exactly as in the reference."""
    import torch.nn.functional as F
    num_frames = latents.shape[1]
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    latents = latents * scheduler.init_noise_sigma
    image_latents = image_latents.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
    controlnet_condition = torch.cat([controlnet_condition] * 2)
    controlnet_flow = torch.cat([controlnet_flow] * 2)
    landmarks = torch.cat([landmarks] * 2)
    if drag_controlnet is not None:
        drag_flow = torch.cat([drag_flow] * 2)
    guidance_scale = torch.linspace(min_guidance_scale, max_guidance_scale, window_size).unsqueeze(0)
    guidance_scale = guidance_scale.to(latents.device, latents.dtype)[(...,) + (None,) * 3]
    added_time_ids = make_added_time_ids(latents.dtype, latents.device)
    views = window_views(num_frames, window_size, stride)
    count = torch.zeros_like(latents)
    value = torch.zeros_like(latents)
    trace = []
    for t in timesteps:
        count.zero_()
        value.zero_()
        seen = {}
        for idx, (t0, t1) in enumerate(views):                                        # :449-509
            lt = torch.cat([latents[:, 0:1], latents[:, t0:t1]], dim=1)
            if reuse_identical_views and (t0, t1) in seen:
                lt = scheduler.step(seen[(t0, t1)], t, lt)
                if idx != len(views) - 1:
                    scheduler._step_index -= 1
                value[:, t0:t1] += lt[:, 1:]
                count[:, t0:t1] += 1
                continue
            il = torch.cat([image_latents[:, 0:1], image_latents[:, t0:t1]], dim=1)
            fl = controlnet_flow[:, (t0 - 1):(t1 - 1)]
            lm = torch.cat([landmarks[:, 0:1], landmarks[:, t0:t1]], dim=1)
            x = torch.cat([lt] * 2)
            x = scheduler.scale_model_input(x, t)
            x = torch.cat([x, il], dim=2)
            down, mid, _, _ = controlnet(x, t, encoder_hidden_states=image_embeddings,
                                         controlnet_cond=controlnet_condition, controlnet_flow=fl, landmarks=lm,
                                         added_time_ids=added_time_ids, conditioning_scale=controlnet_cond_scale,
                                         return_dict=False)
            if drag_controlnet is not None:
                dd, md, _, _ = drag_controlnet(x, t, encoder_hidden_states=image_embeddings,
                                               controlnet_cond=controlnet_condition,
                                               controlnet_flow=drag_flow[:, (t0 - 1):(t1 - 1)], added_time_ids=added_time_ids,
                                               conditioning_scale=ctrl_scale_traj, return_dict=False)
                blended = []
                for a_, b_ in zip(down, dd):
                    m_ = F.interpolate(mask, a_.shape[-2:], mode='nearest')
                    blended.append(a_ * m_ + b_ * (1 - m_))
                m_ = F.interpolate(mask, mid.shape[-2:], mode='nearest')
                down, mid = blended, mid * m_ + md * (1 - m_)
            noise_pred = unet(x, t, encoder_hidden_states=image_embeddings, down_block_additional_residuals=down,
                              mid_block_additional_residual=mid, added_time_ids=added_time_ids, return_dict=False)[0]
            u, c = noise_pred.chunk(2)
            noise_pred = u + guidance_scale * (c - u)
            seen[(t0, t1)] = noise_pred
            lt = scheduler.step(noise_pred, t, lt)
            if idx != len(views) - 1:
                scheduler._step_index -= 1                                            # :499-500
            if idx == 0:
                value[:, 0:t1] += lt
                count[:, 0:t1] += 1
            else:
                value[:, t0:t1] += lt[:, 1:]
                count[:, t0:t1] += 1
        latents = torch.where(count > 0, value / count, value)                        # :511
        if return_trace:
            trace.append(latents.clone())
    return (latents, trace) if return_trace else latents
