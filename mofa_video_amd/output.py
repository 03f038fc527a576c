"""Output stage with the reference's names: ``tensor2vid`` (Traj/pipeline/pipeline.py:57-69; the postprocess of
diffusers' VaeImageProcessor it calls) and ``flow_to_image`` (Traj/utils/flow_viz.py:241-277), on the HIP library."""
import torch

from . import ops

_MODES = {"pt": 0, "np": 1, "pil": 2}


def tensor2vid(video: torch.Tensor, processor=None, output_type: str = "np"):
    """video fp32 [B, 3, T, H, W] (decode_latents' result).  Returns a list with one entry per batch element:
    "pt": fp32 tensor [T,3,H,W] in [0,1]; "np": float32 ndarray [T,H,W,3]; "pil": list of T PIL images.
    ``processor`` is accepted for signature compatibility (the reference passes its VaeImageProcessor)."""
    if output_type not in _MODES:
        raise ValueError(f"output_type {output_type!r} is not one of {sorted(_MODES)}")
    outputs = []
    for b in range(video.shape[0]):
        out = ops.frames_postprocess(video[b].permute(1, 0, 2, 3).contiguous().float(), _MODES[output_type])
        if output_type == "np":
            out = out.cpu().numpy()
        elif output_type == "pil":
            from PIL import Image
            out = [Image.fromarray(f) for f in out.cpu().numpy()]
        outputs.append(out)
    return outputs


def flow_to_image(flow):
    """flow [H, W, 2] (tensor) -> uint8 ndarray [H, W, 3], Middlebury colour code."""
    return ops.flow_to_image(flow.to("cuda", torch.float32)).cpu().numpy()
