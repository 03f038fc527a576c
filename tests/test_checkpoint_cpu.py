"""Host logic of the checkpoint-directory loaders (mofa_video_amd/checkpoint.py): the diffusers / transformers directory
layout the reference loads (MOFA-Video-Traj/run_gradio.py:98-116) and ``ControlNetSDVModel.from_unet`` on state dicts
(models/controlnet_sdv.py:572-628).  No GPU: files and tensors only."""
import json
import os

import pytest
import torch

from helpers import TINY, TINY_CN


def test_save_load_round_trip_safetensors_and_bin(tmp_path):
    from mofa_video_amd import checkpoint, schema
    sd = schema.synthetic_state_dict(schema.unet_schema(TINY), seed=3)
    for safe in (True, False):
        d = str(tmp_path / f"unet_{safe}")
        checkpoint.save_pretrained(d, sd, TINY, "UNetSpatioTemporalConditionModel", safe_serialization=safe)
        assert os.path.exists(os.path.join(d, "config.json"))
        cfg = checkpoint.load_config(d)
        assert cfg["block_out_channels"] == tuple(TINY["block_out_channels"]) and "_class_name" not in cfg
        back = checkpoint.load_state_dict(d)
        assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    with pytest.raises(OSError):
        checkpoint.resolve_dir(str(tmp_path), "missing")
    empty = tmp_path / "empty"
    empty.mkdir()
    (empty / "config.json").write_text(json.dumps({}))
    with pytest.raises(OSError):
        checkpoint.load_state_dict(str(empty))


def test_variant_prefers_fp16_file(tmp_path):
    from safetensors.torch import save_file
    from mofa_video_amd import checkpoint
    d = tmp_path / "m"
    d.mkdir()
    save_file({"w": torch.zeros(2)}, str(d / "diffusion_pytorch_model.safetensors"))
    save_file({"w": torch.ones(2)}, str(d / "diffusion_pytorch_model.fp16.safetensors"))
    assert checkpoint.load_state_dict(str(d))["w"].sum() == 0
    assert checkpoint.load_state_dict(str(d), variant="fp16")["w"].sum() == 2


def test_controlnet_state_dict_from_unet():
    from mofa_video_amd import checkpoint, schema
    usd = schema.synthetic_state_dict(schema.unet_schema(TINY), seed=5)
    sch = schema.controlnet_schema(TINY_CN)
    sd = checkpoint.controlnet_state_dict_from_unet(usd, sch)
    assert set(sd) == set(sch) and all(tuple(sd[k].shape) == tuple(sch[k]) for k in sch)
    copied = [k for k in sch if k.startswith(("conv_in.", "time_embedding.", "down_blocks.", "mid_block."))]
    assert copied and all(torch.equal(sd[k], usd[k]) for k in copied)
    zeros = [k for k in sch if k.startswith(("controlnet_down_blocks.", "controlnet_mid_block.", "flow_encoder.zeroconvs.",
                                             "controlnet_cond_embedding.conv_out."))]
    assert len([k for k in zeros if k.endswith("weight")]) == 12 + 1 + 3 + 1
    assert all(float(sd[k].abs().max()) == 0.0 for k in zeros)
    fresh = [k for k in sch if k.startswith(("add_embedding.", "controlnet_cond_embedding.conv_in."))]
    assert fresh and all(float(sd[k].float().abs().max()) > 0 for k in fresh)      # not copied: default initialisation
    k = "add_embedding.linear_1.weight"
    assert not torch.equal(sd[k], usd[k])
    bound = 1.0 / (sch[k][1] ** 0.5)
    assert float(sd[k].float().abs().max()) <= bound * 1.001
    fresh2 = checkpoint.controlnet_state_dict_from_unet(usd, sch, load_weights_from_unet=False)
    assert not torch.equal(fresh2["conv_in.weight"], usd["conv_in.weight"])
    bad = dict(usd)
    bad.pop("conv_in.weight")
    with pytest.raises(ValueError):
        checkpoint.controlnet_state_dict_from_unet(bad, sch)
