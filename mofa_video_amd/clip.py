"""MI355X host mirror of the reference's ``image_encoder``: transformers ``CLIPVisionModelWithProjection``
(MOFA-Video-Traj/run_gradio.py:23, :98-100; called once per clip at pipeline/pipeline.py:114-139).
``state_dict`` keys are transformers' (``vision_model.embeddings.*``, ``vision_model.pre_layrnorm`` (sic),
``vision_model.encoder.layers.N.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1, mlp.fc2, layer_norm2}``,
``vision_model.post_layernorm``, ``visual_projection``), so the SVD checkpoint's image_encoder loads unchanged.

Launch sequence per image (token-major fp16 [257][hidden]):
  patchify -> igemm (+ position embedding as the row-vector epilogue) -> LayerNorm
  32 x { LayerNorm -> igemm QKV (+bias) -> flash attention -> igemm out_proj (+bias, +residual)
         LayerNorm -> igemm fc1 (+bias, exact GELU epilogue) -> igemm fc2 (+bias, +residual) }
  LayerNorm(CLS) -> igemm visual_projection

Head dim 80 on a kernel that tiles 64 / 128: every head is laid out in a 128-column slot (weights / bias rows of the
unused 48 columns are zero, so Q.K and the out_proj input are unchanged).  The sequence (257) is padded to a multiple of
8 rows; the padded keys are masked *through the spare column*: column 80 of every query is the constant 1 (bias), column
80 of a padded key row is -30000, so its score is -30000 * scale and its softmax weight underflows to exactly 0.
"""
import math

import torch

from . import lib as L
from . import ops
from .blocks import LayerNorm, Linear, Sub
from .weights import f32, pack_linear

MASK = -30000.0
BIG = 1 << 30


def _slot(hd):
    s = 64 if hd < 64 else 128
    assert hd < s <= 128, f"head dim {hd}: no spare column for the key-padding mask"
    return s


class _Layer:
    def __init__(self, s, heads):
        a = s.sub("self_attn")
        d = a.get("q_proj.weight").shape[0]
        hd = d // heads
        sl = _slot(hd)
        self.heads, self.hd, self.slot, self.d = heads, hd, sl, d

        def spread(w, b):                                    # [d, K], [d] -> rows of head h at [h*slot, h*slot + hd)
            wp = w.new_zeros((heads, sl, w.shape[1]))
            wp[:, :hd] = w.reshape(heads, hd, -1)
            bp = b.new_zeros((heads, sl))
            bp[:, :hd] = b.reshape(heads, hd)
            return wp, bp
        ws, bs = [], []
        for n in ("q_proj", "k_proj", "v_proj"):
            wp, bp = spread(a.get(n + ".weight").float(), a.get(n + ".bias").float())
            if n == "q_proj":
                bp[:, hd] = 1.0                              # the mask column of every query
            ws.append(wp.reshape(heads * sl, -1))
            bs.append(bp.reshape(-1))
        self.wqkv = s.dev(pack_linear(torch.cat(ws, 0)))
        self.bqkv = s.dev(f32(torch.cat(bs, 0)))
        wo = a.get("out_proj.weight").float()                # [d, d]: input columns head-major
        wop = wo.new_zeros((d, heads, sl))
        wop[:, :, :hd] = wo.reshape(d, heads, hd)
        self.wo = s.dev(pack_linear(wop.reshape(d, heads * sl)))
        self.bo = s.dev(f32(a.get("out_proj.bias")))
        self.ln1, self.ln2 = LayerNorm(s.sub("layer_norm1")), LayerNorm(s.sub("layer_norm2"))
        self.fc1, self.fc2 = Linear(s.sub("mlp.fc1")), Linear(s.sub("mlp.fc2"))

    def __call__(self, x, qkv, S):
        """x fp16 [S, d]; qkv: the persistent [Spad, 3*heads*slot] buffer whose padded key rows carry the mask"""
        C = self.heads * self.slot
        ops.igemm(self.ln1(x), self.wqkv, self.bqkv, out=qkv[:S])
        o = ops.attn_spatial(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], 1, self.heads, qkv.shape[0], head_dim=self.slot,
                             scale=self.hd ** -0.5)
        x = ops.igemm(o[:S], self.wo, self.bo, r1=x, s1=1.0)
        h = self.fc1(self.ln2(x), act=L.ACT_GELU)
        return self.fc2(h, r1=x, s1=1.0)


class _Out:
    def __init__(self, image_embeds):
        self.image_embeds = image_embeds


class CLIPVisionModelWithProjection:
    def __init__(self, state_dict, config=None, device="cuda"):
        from .schema import CLIP_VIT_H
        c = dict(CLIP_VIT_H)
        c.update(config or {})
        self.config = type("Cfg", (dict,), {"__getattr__": dict.__getitem__})(c)
        self.device, self.dtype = torch.device(device), torch.float16
        s = Sub(state_dict, "vision_model.", device)
        e = s.sub("embeddings")
        wp = e.get("patch_embedding.weight").float()
        self.patch = wp.shape[-1]
        self.wpatch = s.dev(pack_linear(wp.reshape(wp.shape[0], -1)))                       # [d, 3*p*p -> multiple of 64]
        pos = e.get("position_embedding.weight").float()
        self.S = pos.shape[0]
        self.pos_patches = s.dev(f32(pos[1:]))                                                # row vector of patch m
        self.cls16 = s.dev((e.get("class_embedding").float() + pos[0]).to(torch.float16).reshape(1, -1))
        eps = c["layer_norm_eps"]
        self.pre_ln = LayerNorm(s.sub("pre_layrnorm"), eps)
        self.layers = [_Layer(s.sub(f"encoder.layers.{i}"), c["num_attention_heads"]) for i in range(c["num_hidden_layers"])]
        for l in self.layers:
            l.ln1.eps = l.ln2.eps = eps
        self.post_ln = LayerNorm(s.sub("post_layernorm"), eps)
        self.proj = Linear(Sub(state_dict, "visual_projection.", device))
        lay = self.layers[0]
        Sp = (self.S + 7) // 8 * 8
        C = lay.heads * lay.slot
        self.qkv = torch.zeros((Sp, 3 * C), dtype=torch.float16, device=self.device)
        if Sp > self.S:                                                                       # mask column of the padded keys
            self.qkv[self.S:, C:2 * C].view(Sp - self.S, lay.heads, lay.slot)[:, :, lay.hd] = MASK

    @classmethod
    def from_module(cls, module, device="cuda"):
        cfg = module.config
        keys = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size", "patch_size",
                "projection_dim", "layer_norm_eps")
        get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        return cls(module.state_dict(), {k: get(k) for k in keys}, device)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", variant=None, **unused):
        """``image_encoder/`` of the SVD-XT checkpoint directory (transformers layout: ``config.json`` + ``model.safetensors``)"""
        from . import checkpoint
        path = checkpoint.resolve_dir(pretrained_model_name_or_path, subfolder)
        raw = checkpoint.load_config(path)
        raw = dict(raw.get("vision_config") or {}, **{k: v for k, v in raw.items() if k != "vision_config"})
        keys = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size", "patch_size",
                "projection_dim", "layer_norm_eps")
        return cls(checkpoint.load_state_dict(path, variant), {k: raw[k] for k in keys if k in raw}, device)

    def parameters(self):                                                                    # dtype probe of pipeline.py:115
        yield self.wpatch

    def hidden_states(self, pixel_values):
        """fp32 [1, 3, image_size, image_size] -> fp16 [S, d] after the last encoder layer"""
        d = self.wpatch.shape[0]
        x = torch.empty((self.S, d), dtype=torch.float16, device=self.device)
        x[0:1].copy_(self.cls16)
        patches = ops.patchify(pixel_values, self.patch, self.wpatch.shape[1])
        assert patches.shape[0] == self.S - 1, "image size does not match the position embedding"
        ops.igemm(patches, self.wpatch, rowvec=self.pos_patches, rv=(BIG, 1, self.S - 1, BIG), out=x[1:])
        x = self.pre_ln(x)
        for lay in self.layers:
            x = lay(x, self.qkv, self.S)
        return x

    @torch.no_grad()
    def __call__(self, pixel_values):
        pv = pixel_values.to(self.device, torch.float32).contiguous()
        if pv.dim() == 3:
            pv = pv.unsqueeze(0)
        embeds = []
        for b in range(pv.shape[0]):                                                          # one clip = one image
            x = self.hidden_states(pv[b:b + 1])
            embeds.append(self.proj(self.post_ln(x[0:1])))
        return _Out(torch.cat(embeds, 0))
