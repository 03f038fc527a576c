"""Checkpoint directories in the layout the reference loads (diffusers ``save_pretrained``: ``config.json`` +
``diffusion_pytorch_model[.fp16].safetensors`` / ``.bin``; transformers: ``config.json`` + ``model[.fp16].safetensors`` /
``pytorch_model.bin``), for the ``from_pretrained`` class methods of the host mirrors -- the call sites are
MOFA-Video-Traj/run_gradio.py:98-110 (``UNetSpatioTemporalConditionControlNetModel.from_pretrained(path, subfolder="unet")``,
``FlowControlNet.from_pretrained(ckpt_dir/controlnet)``, ``AutoencoderKLTemporalDecoder`` / ``CLIPVisionModelWithProjection``
inside ``FlowControlNetPipeline.from_pretrained``).  Host-side file I/O only; tensors go to the classes' constructors, which
repack them for the HIP kernels.  ``from_unet`` mirrors MOFA-Video-Traj/models/controlnet_sdv.py:572-628.
"""
import json
import math
import os

import torch

_WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "model.safetensors",
                 "model.fp16.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin")


def resolve_dir(pretrained_model_name_or_path, subfolder=None):
    path = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
    if not os.path.isdir(path):
        raise OSError(f"{path} is not a directory (there is no hub download here: pass a local checkpoint directory)")
    return path


def load_config(path):
    """config.json as a dict without the bookkeeping keys ("_class_name", "_diffusers_version", "_name_or_path", ...)"""
    fn = os.path.join(path, "config.json")
    if not os.path.exists(fn):
        raise OSError(f"no config.json in {path}")
    with open(fn) as f:
        cfg = json.load(f)
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if not k.startswith("_")}


def load_state_dict(path, variant=None):
    """the first weight file found; ``variant="fp16"`` prefers the ``.fp16.`` files (diffusers convention)"""
    names = list(_WEIGHT_FILES)
    if variant:
        names.sort(key=lambda n: 0 if f".{variant}." in n else 1)
    for n in names:
        fn = os.path.join(path, n)
        if os.path.exists(fn):
            if fn.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(fn, device="cpu")
            sd = torch.load(fn, map_location="cpu", weights_only=True)
            return sd.get("state_dict", sd)
    raise OSError(f"no weight file in {path} (looked for {', '.join(_WEIGHT_FILES)})")


def save_pretrained(path, state_dict, config, class_name=None, safe_serialization=True):
    """write ``config.json`` + weights in the layout ``from_pretrained`` reads (tests and checkpoint conversion)"""
    os.makedirs(path, exist_ok=True)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(config or {}).items()}
    if class_name:
        cfg["_class_name"] = class_name
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    sd = {k: v.detach().cpu().contiguous() for k, v in state_dict.items()}
    if safe_serialization:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(path, "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, os.path.join(path, "diffusion_pytorch_model.bin"))


def controlnet_state_dict_from_unet(unet_state_dict, schema, load_weights_from_unet=True, seed=0):
    """``ControlNetSDVModel.from_unet`` (controlnet_sdv.py:572-628) on state dicts.  The reference copies exactly
    ``conv_in``, ``time_proj`` (no parameters), ``time_embedding``, ``down_blocks`` and ``mid_block`` (:610-619;
    ``add_embedding`` keeps its fresh initialisation).  Every other adapter
    parameter is initialised as the reference constructors do: ``zero_module`` sites (the 13 ControlNet output convs, the
    last conv of the condition embedding, the flow encoder's zero convs) are zeros, the rest PyTorch's default
    kaiming-uniform(a = sqrt 5) / uniform(+-1/sqrt(fan_in)) draws (seeded here)."""
    copied = ("conv_in.", "time_embedding.", "down_blocks.", "mid_block.")
    zero_sites = ("controlnet_down_blocks.", "controlnet_mid_block.", "controlnet_cond_embedding.conv_out.",
                  "flow_encoder.zeroconvs.", "zero_outs.")
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shape in schema.items():
        if load_weights_from_unet and k.startswith(copied):
            if k not in unet_state_dict or tuple(unet_state_dict[k].shape) != tuple(shape):
                raise ValueError(f"from_unet: the UNet holds no parameter {k} of shape {tuple(shape)}")
            out[k] = unet_state_dict[k].detach().clone()
        elif k.startswith(zero_sites):
            out[k] = torch.zeros(shape, dtype=torch.float16)
        elif k.endswith("mix_factor"):
            out[k] = torch.full(shape, 0.5, dtype=torch.float16)
        elif len(shape) == 1 and k.endswith(".weight") and ("norm" in k):
            out[k] = torch.ones(shape, dtype=torch.float16)
        elif len(shape) == 1 and ("norm" in k):
            out[k] = torch.zeros(shape, dtype=torch.float16)
        else:
            if len(shape) > 1:
                fan_in = int(math.prod(shape[1:]))
            else:                                   # a bias: the fan-in of its layer's weight
                wk = k[:-len("bias")] + "weight"
                fan_in = int(math.prod(schema[wk][1:])) if wk in schema else int(shape[0])
            bound = 1.0 / math.sqrt(max(fan_in, 1))
            out[k] = ((torch.rand(shape, generator=g) * 2 - 1) * bound).to(torch.float16)
    return out
