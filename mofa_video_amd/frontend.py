"""Image conditioning front end: what the reference pipeline does once per clip before the denoising loop,
MOFA-Video-Traj/pipeline/pipeline.py ``_resize_with_antialiasing`` (:531-562), ``_encode_image`` (:114-139),
``_encode_vae_image`` (:141-162) and the call site (:330-352).  Launch sequencing only; the arithmetic is in
libmofa_hip.so (csrc/frontend.hip, igemm, attention, norm)."""
import numpy as np
import torch

from . import ops


def _gaussian_taps(window_size, sigma, device):
    """pipeline.py:613-629 on the host (a handful of fp32 values), uploaded as the filter taps"""
    x = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    if window_size % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
    return (g / g.sum()).to(device)


def blur_geometry(h, w, size):
    """((ky, kx), (sigma_y, sigma_x)) of pipeline.py:538-556: sigma = max((factor - 1) / 2, 0.001), kernel 2 * 2 * sigma
    truncated, at least 3, made odd"""
    fy, fx = h / size[0], w / size[1]
    sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
    ky, kx = int(max(2.0 * 2 * sy, 3)), int(max(2.0 * 2 * sx, 3))
    return (ky + (ky % 2 == 0), kx + (kx % 2 == 0)), (sy, sx)


def _resize_with_antialiasing(input, size, interpolation="bicubic", align_corners=True):
    if interpolation != "bicubic" or not align_corners:
        raise ValueError("only the reference's call (bicubic, align_corners=True) is implemented")
    x = input.to(torch.float32)
    if x.ndim == 3:
        x = x.unsqueeze(0)
    x = x.contiguous()
    (ky, kx), (sy, sx) = blur_geometry(x.shape[-2], x.shape[-1], size)
    x = ops.filter1d_reflect(x, _gaussian_taps(kx, sx, x.device), axis=1)      # x pass, then y pass (:641-642)
    x = ops.filter1d_reflect(x, _gaussian_taps(ky, sy, x.device), axis=0)
    return ops.resize_bicubic_ac(x, size[0], size[1])


def image_to_01(image, height, width, device):
    """PIL image(s) / numpy / tensor -> fp32 [1, 3, h, w] in [0, 1] on ``device`` (VaeImageProcessor pil_to_numpy +
    numpy_to_pt, pipeline.py:118-119).  ``height`` / ``width`` None: the image keeps its own size (what the reference's
    ``_encode_image`` feeds to the 224 x 224 antialiased resize).  Otherwise the size ``image_processor.preprocess(image,
    height, width)`` produces (:338, :391): PIL inputs are resized on the host with PIL's lanczos filter, numpy / tensor
    inputs by nearest-neighbour interpolation (``F.interpolate`` default), as VaeImageProcessor.resize does."""
    if isinstance(image, (list, tuple)):
        if len(image) != 1:
            raise ValueError("one clip per call: pass a single conditioning image")
        image = image[0]
    if torch.is_tensor(image):
        t = image.to(torch.float32)
        if t.dim() == 3:
            t = t.unsqueeze(0)
    else:
        if hasattr(image, "resize") and hasattr(image, "size") and not isinstance(image, np.ndarray):
            if height is not None and tuple(image.size) != (width, height):
                from PIL import Image
                image = image.resize((width, height), resample=Image.LANCZOS)
            image = np.array(image.convert("RGB")).astype(np.float32) / 255.0
        t = torch.from_numpy(np.asarray(image, dtype=np.float32))
        if t.dim() == 3:
            t = t.unsqueeze(0)
        t = t.permute(0, 3, 1, 2)
    if t.shape[0] != 1 or t.shape[1] != 3:
        raise ValueError(f"conditioning image is {tuple(t.shape)}, expected (1, 3, H, W)")
    t = t.to(device).contiguous()
    if height is not None and tuple(t.shape[-2:]) != (height, width):
        t = ops.resize_nearest_f32(t[0], height, width).unsqueeze(0)
    return t.contiguous()


def preprocess(image, height, width, device):
    """``VaeImageProcessor.preprocess(image, height, width)`` (diffusers 0.24.0 image_processor.py, called at pipeline.py:338
    and :391): resize as in ``image_to_01``, then ``normalize`` to [-1, 1] -- unless a tensor input already holds negative
    values, in which case diffusers warns and skips the normalisation.  fp32 [1, 3, height, width]."""
    already = torch.is_tensor(image) and bool((image.min() < 0).item())
    x = image_to_01(image, height, width, device)
    if already:
        return x
    y = x.clone()
    ops.axpby_f32_(torch.ones_like(x), y, a=-1.0, b=2.0)               # 2 x - 1
    return y


@torch.no_grad()
def encode_image(image_encoder, image01, do_classifier_free_guidance=True):
    """pipeline.py:114-139.  image01 fp32 [1, 3, H, W] in [0, 1] (the reference feeds the resized [0, 1] image straight to
    CLIP) -> fp32 [2, 1, D] = [zeros, image_embeds]"""
    emb = image_encoder(_resize_with_antialiasing(image01, (224, 224))).image_embeds
    emb = emb.to(torch.float32).unsqueeze(1)
    return torch.cat([torch.zeros_like(emb), emb]) if do_classifier_free_guidance else emb


@torch.no_grad()
def encode_vae_image(vae, image01, noise_aug_strength=0.02, generator=None, do_classifier_free_guidance=True, noise=None):
    """pipeline.py:338-352 + :141-162: x = 2*image - 1 (VaeImageProcessor.normalize), + noise_aug_strength * randn, VAE
    encoder mode, [zeros, latents].  ``noise`` overrides the draw (tests)."""
    img = image01.to(torch.float32).contiguous()
    if noise is None:
        gdev = generator.device if generator is not None else img.device
        noise = torch.randn(img.shape, generator=generator, device=gdev, dtype=torch.float32)
    x = noise.to(img.device, torch.float32).clone().contiguous()
    ops.axpby_f32_(img, x, a=2.0, b=float(noise_aug_strength))         # x = 2*image + strength*noise
    ops.axpby_f32_(torch.ones_like(img), x, a=-1.0, b=1.0)             #     - 1
    lat = vae.encode(x).latent_dist.mode()
    return torch.cat([torch.zeros_like(lat), lat]) if do_classifier_free_guidance else lat
