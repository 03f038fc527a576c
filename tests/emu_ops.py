"""TEST INFRASTRUCTURE ONLY: a torch-CPU stand-in for the subset of ``mofa_video_amd.ops`` that the front-end host code
calls, so that the *host logic* (weight repacking into 128-column head slots, the key-padding mask column, quant_conv
folded into conv_out, trailing-pad geometry, buffer views) can be exercised without a GPU in the ``-m "not gpu"`` suite.
It is installed by monkeypatching inside tests/test_frontend_host_cpu.py and nowhere else; the product has no CPU path
(mofa_video_amd/lib.py raises without libmofa_hip.so, and these functions are never importable from the package)."""
import torch
import torch.nn.functional as F

from mofa_video_amd import lib as L

F16, F32 = torch.float16, torch.float32


def igemm(x, w, bias=None, geom=None, M=None, rowvec=None, rv=(1, 1, 1, 1 << 30), r1=None, s1=1.0, r2=None, s2=1.0,
          act=L.ACT_NONE, s_acc=1.0, out=None):
    from mofa_video_amd import ops
    geom = geom or ops.PLAIN
    N, Ktot = w.shape
    wf = w.float()
    if geom.mode == L.MODE_PLAIN:
        y = x[:, :Ktot].float() @ wf.T
    elif geom.mode == L.MODE_CONV3X3:
        k = geom.ksize
        Cin = Ktot // (k * k)
        nimg = x.shape[0] // (geom.Hin * geom.Win)
        xi = x[:, :Cin].float().reshape(nimg, geom.Hin, geom.Win, Cin).permute(0, 3, 1, 2)
        if geom.up == 2:
            xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
        wk = wf.reshape(N, k, k, Cin).permute(0, 3, 1, 2)
        p = geom.dil * (k // 2)
        xi = F.pad(xi, (0, p, 0, p) if geom.pad == L.PAD_TRAILING else (p, p, p, p))
        y = F.conv2d(xi, wk, stride=geom.stride, dilation=geom.dil)
        assert tuple(y.shape[-2:]) == (geom.Hout, geom.Wout), (y.shape, geom.Hout, geom.Wout)
        y = y.permute(0, 2, 3, 1).reshape(-1, N)
    else:
        raise NotImplementedError("temporal conv is not used by the front end")
    if M is not None:
        y = y[:M]
    M = y.shape[0]
    if bias is not None:
        y = y + bias.float()
    if rowvec is not None:
        m = torch.arange(M)
        idx = ((m // rv[0]) * rv[1] + (m % rv[2])) % rv[3]
        # the kernel addresses row idx at base + idx*N whatever the view's own row stride is
        rows = torch.as_strided(rowvec, (int(idx.max()) + 1, N), (N, 1))
        y = y + rows.float()[idx]
    y = y * s_acc
    if r1 is not None:
        y = y + s1 * r1[:, :N].float()
    if r2 is not None:
        y = y + s2 * r2[:, :N].float()
    if act == L.ACT_SILU:
        y = F.silu(y)
    elif act == L.ACT_RELU:
        y = F.relu(y)
    elif act == L.ACT_GELU:
        y = F.gelu(y)
    elif act != L.ACT_NONE:
        raise NotImplementedError
    if out is None:
        out = torch.empty((M, N), dtype=F16)
    assert out.shape[0] == M and out.shape[1] >= N
    out[:, :N] = y.to(F16)
    return out


def layer_norm(x, gamma, beta, eps=1e-5, rowvec=None, rv_div=1, rv_mod=1, out=None):
    assert rowvec is None
    return F.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps).to(F16)


def group_norm(x, gamma, beta, nframes, HW, eps, frames_per_stat=1, silu=False, out=None, C_=None, reduce_fn=None,
               frames_total=None):
    assert frames_per_stat == 1 and reduce_fn is None
    C = gamma.numel()
    y = F.group_norm(x[:, :C].float().reshape(nframes, HW, C).permute(0, 2, 1), 32, gamma, beta, eps)
    y = F.silu(y) if silu else y
    return y.permute(0, 2, 1).reshape(nframes * HW, C).to(F16)


def attn_spatial(q, k, v, nframes, heads, S, head_dim=64, scale=None, out=None, prescaled=False):
    assert head_dim in (64, 128) and S % 8 == 0
    scale = head_dim ** -0.5 if scale is None else scale
    if prescaled:                       # q holds Q * head_dim^-0.5 * log2(e): softmax in base 2
        scale = 0.6931471805599453

    def split(t):
        return t.float().reshape(nframes, S, heads, head_dim).permute(0, 2, 1, 3)
    p = torch.softmax(split(q) @ split(k).transpose(-1, -2) * scale, dim=-1)
    return (p @ split(v)).permute(0, 2, 1, 3).reshape(nframes * S, heads * head_dim).to(F16)


def transpose_v(v, nframes, ncb, S):
    return v.reshape(nframes, S, ncb * 64).permute(0, 2, 1).reshape(nframes * ncb * 64, S).contiguous()


def softmax_rows_(x):
    x.copy_(torch.softmax(x.float(), dim=-1).to(F16))
    return x


def nchw_to_tokens(x, ld=None, scale=1.0, out=None):
    n, C, H, W = x.shape
    y = torch.zeros((n * H * W, ld or C), dtype=F16)
    y[:, :C] = (x * scale).permute(0, 2, 3, 1).reshape(-1, C).to(F16)
    return y


def tokens_to_nchw(x, n, Cc, H, W):
    return x[:, :Cc].float().reshape(n, H, W, Cc).permute(0, 3, 1, 2).contiguous()


def patchify(x, p, ld):
    n, C, H, W = x.shape
    y = torch.zeros((n * (H // p) * (W // p), ld), dtype=F16)
    y[:, :C * p * p] = F.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, C * p * p).to(F16)
    return y


def filter1d_reflect(x, taps, axis):
    k = taps.numel()
    front = (k - 1) // 2
    shape = x.shape
    x4 = x.reshape(-1, 1, *shape[-2:])
    if axis == 1:
        y = F.conv2d(F.pad(x4, (front, k - 1 - front, 0, 0), mode="reflect"), taps.view(1, 1, 1, k))
    else:
        y = F.conv2d(F.pad(x4, (0, 0, front, k - 1 - front), mode="reflect"), taps.view(1, 1, k, 1))
    return y.reshape(shape)


def resize_bicubic_ac(x, Ho, Wo):
    shape = x.shape
    return F.interpolate(x.reshape(-1, 1, *shape[-2:]), size=(Ho, Wo), mode="bicubic",
                         align_corners=True).reshape(*shape[:-2], Ho, Wo)


def axpby_f32_(x, y, a=1.0, b=1.0):
    y.copy_(a * x + b * y)
    return y


def cast_f16_to_f32(x):
    return x.float()


def cast_f32_to_f16(x):
    return x.half()


def resize_nearest_f32(x, h, w):
    return torch.nn.functional.interpolate(x[None].float(), size=(h, w), mode="nearest")[0]


NAMES = ["resize_nearest_f32", "axpby_f32_", "cast_f16_to_f32", "cast_f32_to_f16", "igemm", "layer_norm", "group_norm", "attn_spatial", "transpose_v", "softmax_rows_", "nchw_to_tokens",
         "tokens_to_nchw", "patchify", "filter1d_reflect", "resize_bicubic_ac"]


def install(monkeypatch):
    from mofa_video_amd import ops
    for n in NAMES:
        monkeypatch.setattr(ops, n, globals()[n])
