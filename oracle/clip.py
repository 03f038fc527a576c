"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU fp32 restatement of the CLIP vision tower with projection that the
reference loads as ``image_encoder`` (MOFA-Video-Traj/run_gradio.py:23, :98-100; called at pipeline/pipeline.py:114-139
``_encode_image``): transformers ``CLIPVisionModelWithProjection`` (models/clip/modeling_clip.py), a third-party
dependency that is not in the reference tree.  SVD's image encoder is the OpenCLIP ViT-H/14 configuration: hidden 1280,
32 layers, 16 heads (head dim 80), MLP 5120 with exact GELU, 224x224 / patch 14 -> 257 tokens, projection 1024,
layer_norm_eps 1e-5.  Parameter names are transformers' (including its ``pre_layrnorm`` spelling) so the checkpoint's
``state_dict`` loads unchanged.

Pinned by tests/golden/reference_golden_frontend.pt: transformers' own class (4.x, importable in the build container)
run on the same seeded state_dict (tests/golden/make_golden_frontend.py)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

VIT_H = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
             patch_size=14, projection_dim=1024, layer_norm_eps=1e-5)


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, p = c["hidden_size"], c["patch_size"]
        self.class_embedding = nn.Parameter(torch.randn(d))
        self.patch_embedding = nn.Conv2d(3, d, kernel_size=p, stride=p, bias=False)
        self.position_embedding = nn.Embedding((c["image_size"] // p) ** 2 + 1, d)

    def forward(self, pixel_values):
        x = self.patch_embedding(pixel_values).flatten(2).transpose(1, 2)                  # [B, patches, d]
        x = torch.cat([self.class_embedding.expand(x.shape[0], 1, -1), x], dim=1)
        return x + self.position_embedding.weight[None]


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        d = c["hidden_size"]
        self.heads = c["num_attention_heads"]
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))

    def forward(self, x):
        B, S, d = x.shape
        hd = d // self.heads

        def split(t):
            return t.view(B, S, self.heads, hd).transpose(1, 2)
        q, k, v = split(self.q_proj(x)), split(self.k_proj(x)), split(self.v_proj(x))
        p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
        return self.out_proj((p @ v).transpose(1, 2).reshape(B, S, d))


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c["hidden_size"], c["intermediate_size"])
        self.fc2 = nn.Linear(c["intermediate_size"], c["hidden_size"])

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, eps = c["hidden_size"], c["layer_norm_eps"]
        self.self_attn, self.mlp = _Attention(c), _MLP(c)
        self.layer_norm1, self.layer_norm2 = nn.LayerNorm(d, eps=eps), nn.LayerNorm(d, eps=eps)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c["num_hidden_layers"])])


class _VisionTransformer(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, eps = c["hidden_size"], c["layer_norm_eps"]
        self.embeddings = _Embeddings(c)
        self.pre_layrnorm = nn.LayerNorm(d, eps=eps)
        self.encoder = _Encoder(c)
        self.post_layernorm = nn.LayerNorm(d, eps=eps)

    def forward(self, pixel_values):
        x = self.pre_layrnorm(self.embeddings(pixel_values))
        for layer in self.encoder.layers:
            x = layer(x)
        return self.post_layernorm(x[:, 0])                                               # pooled CLS token


class _Out:
    def __init__(self, image_embeds):
        self.image_embeds = image_embeds


class CLIPVisionModelWithProjection(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        c = dict(VIT_H)
        c.update(config or {})
        self.config = c
        self.vision_model = _VisionTransformer(c)
        self.visual_projection = nn.Linear(c["hidden_size"], c["projection_dim"], bias=False)

    @torch.no_grad()
    def forward(self, pixel_values):
        return _Out(self.visual_projection(self.vision_model(pixel_values)))
