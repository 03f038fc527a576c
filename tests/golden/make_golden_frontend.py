"""Golden vectors for the image-conditioning front end.

* ``resize``: the reference's OWN ``_resize_with_antialiasing`` (+ ``_compute_padding``, ``_filter2d``, ``_gaussian``,
  ``_gaussian_blur2d``; MOFA-Video-Traj/pipeline/pipeline.py:531-645).  The module imports diffusers at import time, so
  the FunctionDef nodes are taken from the source file in place (ast) and executed in a namespace that only holds torch.
* ``encode_image``: the reference's OWN ``FlowControlNetPipeline._encode_image`` method (pipeline.py:114-139), executed the
  same way on a stand-in ``self`` whose ``image_encoder`` is transformers' CLIPVisionModelWithProjection (the class the
  reference loads, run_gradio.py:98) at a 2-layer, head-dim-80 configuration.
Weights: schema.synthetic_state_dict(schema.clip_vision_schema(CFG), seed=31, dtype=float32) -- not stored, regenerated
by the tests.      python tests/golden/make_golden_frontend.py      (needs /root/reference and transformers)"""
import ast
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from mofa_video_amd import schema  # noqa: E402

REF = "/root/reference/MOFA-Video-Traj/pipeline/pipeline.py"
CFG = dict(hidden_size=320, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4, projection_dim=64)
tree = ast.parse(open(REF).read())
ns = {"torch": torch}
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in ("_resize_with_antialiasing", "_compute_padding", "_filter2d",
                                                           "_gaussian", "_gaussian_blur2d"):
        exec(compile(ast.Module([node], []), REF, "exec"), ns)
    if isinstance(node, ast.ClassDef) and node.name == "FlowControlNetPipeline":
        for m in node.body:
            if isinstance(m, ast.FunctionDef) and m.name == "_encode_image":
                m.decorator_list = []
                exec(compile(ast.Module([m], []), REF, "exec"), ns)
resize = ns["_resize_with_antialiasing"]

G = {"resize": {}}
g = torch.Generator().manual_seed(11)
cases = {"down_96x160_to_32": ((1, 3, 96, 160), (32, 32)), "down_72x128_to_24x40": ((2, 3, 72, 128), (24, 40)),
         "up_40x56_to_64": ((1, 3, 40, 56), (64, 64)), "chw_50x70_to_20x30": ((3, 50, 70), (20, 30)),
         "clip_144x256_to_224": ((1, 3, 144, 256), (224, 224))}
for name, (shape, size) in cases.items():
    x = torch.rand(shape, generator=g)
    G["resize"][name] = dict(x=x, size=size, out=resize(x.clone(), size))

from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection  # noqa: E402

enc = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_act="gelu", patch_size=14, image_size=224, **CFG)).eval()
sd = schema.synthetic_state_dict(schema.clip_vision_schema(CFG), seed=31, dtype=torch.float32)
missing, unexpected = enc.load_state_dict(sd, strict=False)
assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)


class _Self:                       # what _encode_image touches for tensor input: image_encoder only
    image_encoder = enc


image = torch.rand(1, 3, 144, 256, generator=g)
with torch.no_grad():
    emb = ns["_encode_image"](_Self(), image.clone(), "cpu", 1, True)
    last = enc.vision_model(resize(image.clone(), (224, 224))).last_hidden_state
G["encode_image"] = dict(cfg=CFG, seed=31, image=image, image_embeddings=emb, last_hidden_state_cls=last[:, 0].clone(),
                         last_hidden_state_tok200=last[:, 200].clone())
torch.save(G, os.path.join(HERE, "reference_golden_frontend.pt"))
print({k: tuple(v["out"].shape) for k, v in G["resize"].items()}, tuple(emb.shape),
      os.path.getsize(os.path.join(HERE, "reference_golden_frontend.pt")))
