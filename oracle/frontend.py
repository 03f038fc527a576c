"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the image-conditioning front end that runs once per
clip before the denoising loop, MOFA-Video-Traj/pipeline/pipeline.py:

* ``resize_with_antialiasing``  :531-562 (+ ``_compute_padding`` :565-584, ``_filter2d`` :587-610, ``_gaussian`` :613-629,
  ``_gaussian_blur2d`` :632-645): Gaussian blur (sigma = max((factor - 1) / 2, 0.001), kernel = 2 * 2 * sigma rounded to
  odd, at least 3, reflect padding, x pass then y pass) followed by a bicubic, align-corners resize.  The reference feeds
  the [0, 1] image straight to CLIP: no (x+1)/2, no CLIP mean / std normalisation (:114-125).
* ``encode_image``      :114-139: resize to 224x224 -> image encoder -> [uncond = zeros, cond]
* ``encode_vae_image``  :141-162 and the call site :338-352: image in [-1, 1] + noise_aug_strength * noise ->
  ``vae.encode(x).latent_dist.mode()`` (the mean half of quant_conv's output) -> [zeros, latents]

Pinned by tests/golden/reference_golden_frontend.pt: the reference's own functions, executed in place
(tests/golden/make_golden_frontend.py)."""
import torch
import torch.nn.functional as F


def _gaussian(window_size, sigma, dtype=torch.float32):
    x = torch.arange(window_size, dtype=dtype) - window_size // 2
    if window_size % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(sigma, dtype=dtype).pow(2.0)))
    return g / g.sum()


def blur_geometry(h, w, size):
    """-> ((ky, kx), (sigma_y, sigma_x)) of pipeline.py:538-556"""
    fy, fx = h / size[0], w / size[1]
    sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
    ky, kx = int(max(2.0 * 2 * sy, 3)), int(max(2.0 * 2 * sx, 3))
    return (ky + (ky % 2 == 0), kx + (kx % 2 == 0)), (sy, sx)


def _filter_axis(x, taps, axis):
    k = taps.numel()
    front = (k - 1) // 2
    pad = (front, k - 1 - front, 0, 0) if axis == 1 else (0, 0, front, k - 1 - front)
    b, c, h, w = x.shape
    x = F.pad(x, pad, mode="reflect")
    kern = taps.view(1, 1, 1, k) if axis == 1 else taps.view(1, 1, k, 1)
    return F.conv2d(x.reshape(b * c, 1, *x.shape[-2:]), kern.to(x.dtype)).view(b, c, h, w)


def resize_with_antialiasing(x, size, interpolation="bicubic", align_corners=True):
    if x.ndim == 3:
        x = x.unsqueeze(0)
    (ky, kx), (sy, sx) = blur_geometry(x.shape[-2], x.shape[-1], size)
    x = _filter_axis(x, _gaussian(kx, sx, x.dtype), 1)
    x = _filter_axis(x, _gaussian(ky, sy, x.dtype), 0)
    return F.interpolate(x, size=size, mode=interpolation, align_corners=align_corners)


@torch.no_grad()
def encode_image(image_encoder, image01, do_classifier_free_guidance=True):
    """image01: fp32 [B, 3, H, W] in [0, 1] (the numpy_to_pt form of the PIL input) -> [2B, 1, D]"""
    emb = image_encoder(resize_with_antialiasing(image01, (224, 224))).image_embeds.unsqueeze(1)
    return torch.cat([torch.zeros_like(emb), emb]) if do_classifier_free_guidance else emb


@torch.no_grad()
def encode_vae_image(vae, image, noise=None, noise_aug_strength=0.02, do_classifier_free_guidance=True):
    """image: fp32 [B, 3, H, W] in [-1, 1] (VaeImageProcessor.preprocess); noise: the randn_tensor draw of :340"""
    if noise is not None:
        image = image + noise_aug_strength * noise
    lat = vae.encode(image).latent_dist.mode()
    return torch.cat([torch.zeros_like(lat), lat]) if do_classifier_free_guidance else lat
