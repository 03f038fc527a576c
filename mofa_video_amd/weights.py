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


def f32(t):
    return t.detach().to(torch.float32).contiguous()
