"""Repacking of reference-layout parameters (diffusers ``state_dict`` tensors) into the kernel layouts of
libmofa_hip.so.  Pure data movement done once at load time (host side, torch as a container only):
weights become fp16 ``[N][taps*Cin]`` with the reduction axis contiguous, vectors become fp32.
"""
import torch

KPAD = 64  # the MFMA implicit-GEMM walks K in chunks of 64 channels per tap


def _pad_cin(w, dim):
    cin = w.shape[dim]
    pad = (-cin) % KPAD
    if pad == 0:
        return w
    shape = list(w.shape)
    shape[dim] = pad
    return torch.cat([w, w.new_zeros(shape)], dim=dim)


def pack_linear(w):
    """nn.Linear / 1x1 conv weight [N, K(,1,1)] -> fp16 [N, Kpad]."""
    w = w.reshape(w.shape[0], -1)
    return _pad_cin(w, 1).to(torch.float16).contiguous()


def pack_conv3x3(w):
    """nn.Conv2d weight [N, Cin, k, k] (k = 3 or 7) -> fp16 [N, k*k*Cinpad], tap = ky*k + kx major."""
    w = _pad_cin(w, 1)
    n, cin, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(n, kh * kw * cin).to(torch.float16).contiguous()


def pack_conv3d_t3(w):
    """nn.Conv3d weight [N, Cin, 3, 1, 1] -> fp16 [N, 3*Cinpad], tap = kt major."""
    w = _pad_cin(w[..., 0, 0], 1)          # [N, Cin, 3]
    n, cin = w.shape[:2]
    return w.permute(0, 2, 1).reshape(n, 3 * cin).to(torch.float16).contiguous()


def pad_rows(w, mult=4):
    """pad the output-channel axis (N % 4 == 0 is required by the epilogue)."""
    pad = (-w.shape[0]) % mult
    if pad == 0:
        return w
    return torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)


def interleave_geglu(w, b):
    """GEGLU projection [8C, K]: rows [0,4C) = value, [4C,8C) = gate.  Interleave in blocks of 16 rows (value block j, gate
    block j): every 32-row MFMA tile of the GEMM then holds 16 value columns and the 16 matching gate columns, and in the
    32x32 accumulator layout (register r <-> column 8 (r >> 2) + 4 (lane >> 5) + (r & 3)) registers r and r + 8 of ONE lane are
    a value / gate pair -- the x * gelu(gate) product is formed lane-locally in the epilogue (MOFA_ACT_GEGLU_PAIR) on every
    tile shape, whatever the number of column tiles a wave owns (the 256x320 tile gives a wave five)."""
    n2 = w.shape[0]
    ch = n2 // 2
    assert ch % 16 == 0
    idx = torch.arange(n2).reshape(2, ch // 16, 16).permute(1, 0, 2).reshape(-1)
    return w[idx].contiguous(), (b[idx].contiguous() if b is not None else None)


def pack_ff320(w1, b1, w2, gamma, beta):
    """Operands of the fused level-0 feed-forward (csrc/ff320.hip, ``mofa_ff320_f16``; layout in include/mofa_hip.h) from the
    diffusers parameters of FeedForward(320, activation_fn="geglu") and the LayerNorm in front of it:
    ``w1`` = net.0.proj.weight [2560, 320] (rows [0, 1280) value, [1280, 2560) gate), ``b1`` = net.0.proj.bias, ``w2`` =
    net.2.weight [320, 1280], ``gamma`` / ``beta`` = the norm's weight / bias.  The norm's affine part is folded in fp32:
    W1 . (g * xhat + b) + b1 = (W1 * g) . xhat + (W1 . b + b1); weights are rounded to fp16 once, afterwards.
    Returns (w1p fp16 [40, 2, 20, 64, 8], b1 fp32 [2560], w2p fp16 [40, 10, 2, 64, 8])."""
    C, H = 320, 1280
    w1, w2 = w1.detach().float().cpu().reshape(2 * H, C), w2.detach().float().cpu().reshape(C, H)   # (packed on the host)
    gamma, beta = gamma.detach().float().cpu(), beta.detach().float().cpu()
    b1f = (b1.detach().float().cpu() if b1 is not None else torch.zeros(2 * H)) + w1 @ beta
    w1g = (w1 * gamma[None, :]).to(torch.float16)
    lane = torch.arange(64)
    n, lh = lane % 32, lane // 32
    # w1p[c, t, s, l, e] = w1g[t * 1280 + 32 c + n(l), 16 s + 8 lh(l) + e]
    c, t, s_, e = torch.arange(40), torch.arange(2), torch.arange(20), torch.arange(8)
    rows = (t[None, :, None, None, None] * H + 32 * c[:, None, None, None, None] + n[None, None, None, :, None])
    cols = 16 * s_[None, None, :, None, None] + 8 * lh[None, None, None, :, None] + e[None, None, None, None, :]
    w1p = w1g[rows.expand(40, 2, 20, 64, 8), cols.expand(40, 2, 20, 64, 8)].contiguous()
    # w2p[c, j, u, l, jj] = w2[32 j + n(l), 32 c + 16 u + 4 lh(l) + (jj & 3) + 8 (jj >> 2)]
    j, u, jj = torch.arange(10), torch.arange(2), torch.arange(8)
    rows2 = 32 * j[None, :, None, None, None] + n[None, None, None, :, None]
    cols2 = (32 * c[:, None, None, None, None] + 16 * u[None, None, :, None, None] + 4 * lh[None, None, None, :, None] +
             (jj & 3)[None, None, None, None, :] + 8 * (jj >> 2)[None, None, None, None, :])
    w2p = w2.to(torch.float16)[rows2.expand(40, 10, 2, 64, 8), cols2.expand(40, 10, 2, 64, 8)].contiguous()
    return w1p, b1f.contiguous(), w2p


def pack_lin320(w, b=None, gamma=None, beta=None):
    """Operands of ``mofa_lin320_f16`` (csrc/lin320.hip; layout in include/mofa_hip.h) from an nn.Linear weight [N, 320] (N % 64 == 0)
    and, when a LayerNorm precedes it, the norm's weight / bias, folded in fp32: W (g xhat + beta) + b = (W g) xhat + (W beta + b).
    Returns (wp fp16 [N / 64, 2, 20, 64, 8], bias fp32 [N] or None)."""
    w = w.detach().float().cpu().reshape(w.shape[0], -1)          # (packed on the host whatever device the checkpoint tensors are on)
    N, K = w.shape
    assert K == 320 and N % 64 == 0, (N, K)
    bias = b.detach().float().cpu().clone() if b is not None else None
    if beta is not None:
        bias = (bias if bias is not None else torch.zeros(N)) + w @ beta.detach().float().cpu()
    if gamma is not None:
        w = w * gamma.detach().float().cpu()[None, :]
    wh = w.to(torch.float16)
    lane = torch.arange(64)
    n, lh = lane % 32, lane // 32
    c, t, s_, e = torch.arange(N // 64), torch.arange(2), torch.arange(20), torch.arange(8)
    rows = 64 * c[:, None, None, None, None] + 32 * t[None, :, None, None, None] + n[None, None, None, :, None]
    cols = 16 * s_[None, None, :, None, None] + 8 * lh[None, None, None, :, None] + e[None, None, None, None, :]
    shp = (N // 64, 2, 20, 64, 8)
    return wh[rows.expand(shp), cols.expand(shp)].contiguous(), (bias.contiguous() if bias is not None else None)


def f32(t):
    return t.detach().to(torch.float32).contiguous()
