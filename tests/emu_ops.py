"""TEST INFRASTRUCTURE ONLY: a torch-CPU stand-in for the subset of ``mofa_video_amd.ops`` that the front-end host code
and the frame-sharded temporal block call, so that the *host logic* (the exchange protocol of parallel.FrameParallel, weight repacking into 128-column head slots, the key-padding mask column, quant_conv
folded into conv_out, trailing-pad geometry, buffer views) can be exercised without a GPU in the ``-m "not gpu"`` suite.
It is installed by monkeypatching inside tests/test_frontend_host_cpu.py and the worker processes of tests/test_parallel_gloo.py only; the product has no CPU path
(mofa_video_amd/lib.py raises without libmofa_hip.so, and these functions are never importable from the package)."""
import torch
import torch.nn.functional as F

from mofa_video_amd import lib as L

F16, F32 = torch.float16, torch.float32


def igemm(x, w, bias=None, geom=None, M=None, rowvec=None, rv=(1, 1, 1, 1 << 30), r1=None, s1=1.0, r2=None, s2=1.0,
          act=L.ACT_NONE, s_acc=1.0, out=None, stats=False):
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
        # (3,1,1) convolution over frames of HW rows.  T > 0: clips of T frames, zero padding at the clip ends; T == 0: unclipped,
        # the caller supplies one halo frame before and after a.x (the frame-sharded path: x is a view into a longer buffer)
        Cin, HW = Ktot // 3, geom.HW
        Mr = M if M is not None else x.shape[0]
        wk = wf.reshape(N, 3, Cin)
        if geom.T == 0:
            xe = torch.as_strided(x, (Mr + 2 * HW, Cin), (x.stride(0), 1), x.storage_offset() - HW * x.stride(0)).float()
            y = sum(xe[k * HW:k * HW + Mr] @ wk[:, k].T for k in range(3))
        else:
            xc = x[:Mr, :Cin].float().reshape(-1, geom.T, HW, Cin)
            xp = F.pad(xc, (0, 0, 0, 0, 1, 1))
            y = sum(xp[:, k:k + geom.T].reshape(Mr, Cin) @ wk[:, k].T for k in range(3))
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
    elif act == L.ACT_GEGLU_PAIR:
        # value / gate rows interleaved in blocks of 16 (weights.interleave_geglu): out[:, 16 b + c] = val * gelu(gate)
        yb = y.reshape(M, N // 32, 2, 16)
        y = (yb[:, :, 0] * F.gelu(yb[:, :, 1])).reshape(M, N // 2)
        N = N // 2
    elif act != L.ACT_NONE:
        raise NotImplementedError
    if out is None:
        out = torch.empty((M, N), dtype=F16)
    assert out.shape[0] == M and out.shape[1] >= N
    out[:, :N] = y.to(F16)
    return out


def layer_norm(x, gamma, beta, eps=1e-5, rowvec=None, rv_div=1, rv_mod=1, out=None):
    xf = x.float()
    if rowvec is not None:
        m = torch.arange(x.shape[0])
        xf = xf + rowvec.float()[(m // rv_div) % rv_mod]
    y = F.layer_norm(xf, (x.shape[1],), gamma, beta, eps).to(F16)
    if out is not None:
        out.copy_(y)
        return out
    return y


_UNPACKED = {}


def unpack_ff320(w1p, w2p):
    key = (w1p.data_ptr(), w2p.data_ptr())
    if key not in _UNPACKED:
        _UNPACKED[key] = (_unpack_ff320(w1p, w2p), w1p, w2p)       # (the packed tensors are kept alive with the entry)
    return _UNPACKED[key][0]


def _unpack_ff320(w1p, w2p):
    """inverse of weights.pack_ff320, written from the layout include/mofa_hip.h documents for mofa_ff320_f16 (not from the
    packer): -> (W1g fp16 [2560, 320], W2 fp16 [320, 1280])"""
    w1p, w2p = w1p.reshape(40, 2, 20, 64, 8).cpu(), w2p.reshape(40, 10, 2, 64, 8).cpu()
    w1 = torch.zeros(2560, 320, dtype=F16)
    w2 = torch.zeros(320, 1280, dtype=F16)
    for l in range(64):
        n, lh = l & 31, l >> 5
        for t in range(2):
            # element e of lane l, chunk c, k-step s = W1g[t * 1280 + 32 c + n][16 s + 8 lh + e]
            blk = w1p[:, t, :, l, :]                                     # [40 c, 20 s, 8 e]
            rows = t * 1280 + 32 * torch.arange(40) + n
            cols = (16 * torch.arange(20)[:, None] + 8 * lh + torch.arange(8)[None, :]).reshape(-1)
            w1[rows[:, None], cols[None, :]] = blk.reshape(40, 160)
        for jj in range(8):
            # element jj of lane l, chunk c, out tile j, k-step u = W2[32 j + n][32 c + 16 u + 4 lh + (jj & 3) + 8 (jj >> 2)]
            blk = w2p[:, :, :, l, jj]                                    # [40 c, 10 j, 2 u]
            rows = 32 * torch.arange(10) + n
            cols = 32 * torch.arange(40)[:, None] + 16 * torch.arange(2)[None, :] + 4 * lh + (jj & 3) + 8 * (jj >> 2)
            w2[rows[None, :, None], cols[:, None, :]] = blk
    return w1, w2


def ff320(x, w1p, b1, w2p, b2, eps=1e-5, pos=None, HW=1, T=1, r2=None, s_acc=1.0, s1=1.0, s2=0.0, out=None, ln_out=None,
          ln_eps=1e-5):
    w1, w2 = unpack_ff320(w1p, w2p)
    xf = x[:, :320].float()
    if pos is not None:
        xf = xf + pos.float()[(torch.arange(x.shape[0]) // HW) % T]
    xn = F.layer_norm(xf, (320,), None, None, eps).to(F16).float()
    p = xn @ w1.float().T + b1.float()
    h = (p[:, :1280] * F.gelu(p[:, 1280:])).to(F16).float()
    y = (s_acc * (h @ w2.float().T + b2.float())).to(F16).float() + s1 * xf
    if r2 is not None:
        y = y + s2 * r2[:, :320].float()
    y = y.to(F16)
    if out is not None:
        out[:, :320].copy_(y)
        y = out
    if ln_out is None:
        return y
    yl = F.layer_norm(y[:, :320].float(), (320,), ln_out[0], ln_out[1], ln_eps).to(F16)
    if len(ln_out) > 2 and ln_out[2] is not None:
        ln_out[2][:, :320].copy_(yl)
        yl = ln_out[2]
    return y, yl


def unpack_lin320(wp):
    """inverse of weights.pack_lin320 from the layout include/mofa_hip.h documents for mofa_lin320_f16: -> W' fp16 [N, 320]"""
    key = ("lin", wp.data_ptr())
    if key not in _UNPACKED:
        nch = wp.shape[0]
        p = wp.reshape(nch, 2, 20, 64, 8).cpu()
        w = torch.zeros(nch * 64, 320, dtype=F16)
        for l in range(64):
            n, lh = l & 31, l >> 5
            for t in range(2):
                rows = 64 * torch.arange(nch) + 32 * t + n
                cols = (16 * torch.arange(20)[:, None] + 8 * lh + torch.arange(8)[None, :]).reshape(-1)
                w[rows[:, None], cols[None, :]] = p[:, t, :, l, :].reshape(nch, 160)
        _UNPACKED[key] = (w, wp)
    return _UNPACKED[key][0]


def lin320(x, wp, bias=None, norm=False, eps=1e-5, rowvec=None, rv=(1, 1, 1, 1 << 30), r1=None, s1=1.0, s_acc=1.0, out=None):
    w = unpack_lin320(wp)
    N, M = w.shape[0], x.shape[0]
    xf = x[:, :320].float()
    if norm:
        xf = F.layer_norm(xf, (320,), None, None, eps).to(F16).float()
    y = xf @ w.float().T
    if bias is not None:
        y = y + bias.float()
    if rowvec is not None:
        m = torch.arange(M)
        idx = ((m // rv[0]) * rv[1] + (m % rv[2])) % rv[3]
        rows = torch.as_strided(rowvec, (int(idx.max()) + 1, N), (N, 1))
        y = y + rows.float()[idx]
    y = (s_acc * y).to(F16).float()
    if r1 is not None:
        y = y + s1 * r1[:, :N].float()
    y = y.to(F16)
    if out is not None:
        out[:, :N].copy_(y)
        return out
    return y


def group_norm(x, gamma, beta, nframes, HW, eps, frames_per_stat=1, silu=False, out=None, C_=None):
    C = gamma.numel()
    fps = frames_per_stat
    y = F.group_norm(x[:, :C].float().reshape(nframes // fps, fps * HW, C).permute(0, 2, 1), 32, gamma, beta, eps)
    y = F.silu(y) if silu else y
    y = y.permute(0, 2, 1).reshape(nframes * HW, C).to(F16)
    if out is not None:
        out.copy_(y)
        return out
    return y


def gn_nparts(HW, Cc):
    """csrc/norm.hip gn_nchunks"""
    n, cap = (-(-HW // 576), 128) if HW >= 9216 else (-(-HW // 144), 16)
    return max(1, min(n, cap))


def gn_partial_into(x, part_rows, nframes, HW):
    C = x.shape[1]
    nparts = gn_nparts(HW, C)
    rpc = -(-HW // nparts)
    xf = x.float().reshape(nframes, HW, 32, C // 32)
    for f in range(nframes):
        for ch in range(nparts):
            blk = xf[f, ch * rpc:(ch + 1) * rpc]
            part_rows[f * nparts + ch] = torch.stack([blk.sum((0, 2)), (blk * blk).sum((0, 2))], -1).reshape(64)


def gn_apply_gathered(x, part_all, count_per_group, gamma, beta, eps, out, nframes, HW, silu=False):
    C = x.shape[1]
    tot = part_all.double().reshape(-1, 32, 2).sum(0)
    mean = tot[:, 0] / count_per_group
    var = (tot[:, 1] / count_per_group - mean * mean).clamp_min(0.0)
    rstd = 1.0 / torch.sqrt(var + eps)
    cpg = C // 32
    sc = rstd.float().repeat_interleave(cpg) * gamma
    y = x.float() * sc + (beta - mean.float().repeat_interleave(cpg) * sc)
    out.copy_((F.silu(y) if silu else y).to(F16))
    return out


def attn_spatial(q, k, v, nframes, heads, S, head_dim=64, scale=None, out=None, prescaled=False):
    assert head_dim in (64, 128) and S % 8 == 0
    scale = head_dim ** -0.5 if scale is None else scale
    if prescaled:                       # q holds Q * head_dim^-0.5 * log2(e): softmax in base 2
        scale = 0.6931471805599453

    def split(t):
        return t.float().reshape(nframes, S, heads, head_dim).permute(0, 2, 1, 3)
    p = torch.softmax(split(q) @ split(k).transpose(-1, -2) * scale, dim=-1)
    return (p @ split(v)).permute(0, 2, 1, 3).reshape(nframes * S, heads * head_dim).to(F16)


def attn_temporal(q, k, v, nclips, T, HW, heads, head_dim=64, scale=None, out=None, Tq=None, key_mask=None):
    """k / v rows (clip, frame t < T, pixel); q / out rows (clip, frame i < Tq, pixel); key_mask bit j = key frame j exists"""
    Tq = T if Tq is None else Tq
    scale = head_dim ** -0.5 if scale is None else scale
    C = heads * head_dim

    def split(t, n):                       # -> [clip, pixel, head, frame, d]
        return t[:, :C].float().reshape(nclips, n, HW, heads, head_dim).permute(0, 2, 3, 1, 4)
    kk, vv = split(k, T), split(v, T)
    dead = None
    if key_mask is not None:               # (the kernel never reads a masked frame's rows: they may hold anything)
        dead = torch.tensor([not ((key_mask >> j) & 1) for j in range(T)])
        kk = kk.masked_fill(dead.view(1, 1, 1, T, 1), 0.0)
        vv = vv.masked_fill(dead.view(1, 1, 1, T, 1), 0.0)
    s = split(q, Tq) @ kk.transpose(-1, -2) * scale
    if dead is not None:
        s = s.masked_fill(dead, float("-inf"))
    o = torch.softmax(s, dim=-1) @ vv
    return o.permute(0, 3, 1, 2, 4).reshape(nclips * Tq * HW, C).to(F16)


def timestep_embedding(t, dim):
    """csrc/elementwise.hip timestep_embedding_kernel: [cos | sin], frequencies exp(-ln(1e4) k / half)"""
    half = dim // 2
    freq = torch.exp(-9.210340371976184 * torch.arange(half, dtype=torch.float32) / half)
    a = t.reshape(-1, 1).float() * freq
    return torch.cat([torch.cos(a), torch.sin(a)], 1)


def silu_f32(x):
    return F.silu(x)


def axpby_(x, y, a=1.0, b=1.0):
    y.copy_((a * x.float() + b * y.float()).to(F16))
    return y


def axpby_out(x, y, a, b, out):
    out.copy_((a * x.float() + b * y.float()).to(F16))
    return out


def copy2d(src, dst):
    dst.copy_(src)
    return dst


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


NAMES = ["attn_temporal", "timestep_embedding", "silu_f32", "axpby_", "axpby_out", "copy2d", "gn_nparts", "gn_partial_into", "gn_apply_gathered", "resize_nearest_f32", "axpby_f32_", "cast_f16_to_f32", "cast_f32_to_f16", "igemm", "ff320", "lin320", "layer_norm", "group_norm", "attn_spatial", "transpose_v", "softmax_rows_", "nchw_to_tokens",
         "tokens_to_nchw", "patchify", "filter1d_reflect", "resize_bicubic_ac"]


def install(monkeypatch=None):
    """monkeypatch=None: plain setattr (spawned test worker processes, which exit afterwards)"""
    from mofa_video_amd import ops
    for n in NAMES:
        if monkeypatch is not None:
            monkeypatch.setattr(ops, n, globals()[n])
        else:
            setattr(ops, n, globals()[n])
