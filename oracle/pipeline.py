"""Oracle (test infrastructure): the denoise loop + decode of
``FlowControlNetPipeline.__call__`` (MOFA-Video-Traj/pipeline/pipeline.py:283-527)
restated for CPU fp32 on *pre-encoded* inputs (image embeddings / image latents are
produced before the hot path by CLIP / VAE-encode).
"""
import torch

from .vae import decode_latents


def make_added_time_ids(dtype=torch.float32):
    """pipeline.py:430-440 -- overwritten to fps=6, motion_bucket_id=128, noise_aug=0.02, x2 for CFG."""
    ids = torch.tensor([[6, 128, 0.02]], dtype=dtype)
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
    guidance_scale = guidance_scale.to(latents.dtype)[(...,) + (None,) * 3]       # :423-426
    added_time_ids = make_added_time_ids(latents.dtype)
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
