"""Host-side block graph of the SVD spatio-temporal UNet / ControlNet trunk / temporal VAE decoder.

These classes only *sequence* launches of libmofa_hip.so (``ops``) over token-major fp16 activations;
they hold repacked weights and no torch arithmetic.  Parameter names are the diffusers==0.24.0 names
the reference checkpoints use, so ``from_state_dict`` consumes an unmodified checkpoint
``state_dict``.

Fusions relative to the reference module graph (all results-identical in exact arithmetic):
  * conv/linear bias, "+ time-embedding" broadcast, residual adds, AlphaBlender and the ControlNet
    ``conditioning_scale`` are igemm epilogues;
  * cross-attention to the single image-embedding token: softmax over one key == 1, so
    attn2(x) = to_out(to_v(ctx)) is a per-clip row vector added in the attn1.to_out epilogue
    (to_q / to_k / norm2 are mathematically dead);
  * GEGLU is formed in the projection GEMM's epilogue (value/gate weight rows interleaved at load);
  * the frame-position embedding add is folded into norm_in and the ff_in output epilogue.
"""
import math

import torch

from . import lib as L
from . import ops
from .weights import (f32, interleave_geglu, pack_conv3d_t3, pack_conv3x3, pack_ff320, pack_lin320, pack_linear, pad_rows)

BIG = 1 << 30


class Sub:
    """prefix view of a flat state_dict"""

    def __init__(self, sd, prefix="", device="cuda"):
        self.sd, self.prefix, self.device = sd, prefix, device

    def sub(self, name):
        return Sub(self.sd, f"{self.prefix}{name}.", self.device)

    def has(self, name):
        return f"{self.prefix}{name}" in self.sd

    def get(self, name):
        return self.sd[f"{self.prefix}{name}"].detach()

    def dev(self, t):
        return t.to(self.device)


class Ctx:
    """Per-forward geometry + conditioning vectors shared by all blocks."""

    def __init__(self, B, T):
        self.B, self.T = B, T
        self.N = B * T
        self.temb_act = None      # fp16 [B, temb_dim] = silu(emb); None for the VAE
        self.temb_all = None      # fp32 [B, total]: every time_emb_proj of the network in one GEMM (TembBatch)
        self.ctx16 = None         # fp16 [B, cross_dim] image embedding (first frame's == every frame's)
        self.cache = {}           # timestep-invariant per-layer vectors (cross-attention, frame-position emb)
        self.time_context_hw_major = True
        self.par = None           # parallel.FrameParallel when this rank holds only frames [f0, f1) of the clip
        self.ctx16_all = None     # fp16 [B_global, cross_dim]: both CFG halves' embeddings (temporal-context quirk)
        self.half = 0             # global CFG-half index of local batch row 0
        self.halves = None        # [Ctx(1, T), Ctx(1, T)] of a B = 2 context whose CFG halves are evaluated separately (pipeline.py)


# ---------------------------------------------------------------------------------------------------------
class Linear:
    def __init__(self, s, pad_n=False):
        w = s.get("weight")
        w = w.reshape(w.shape[0], -1)
        self.n_real = w.shape[0]
        b = s.get("bias") if s.has("bias") else None
        if pad_n:
            w = pad_rows(w)
            b = pad_rows(b) if b is not None else None
        self.w = s.dev(pack_linear(w))
        self.b = s.dev(f32(b)) if b is not None else None

    def __call__(self, x, **kw):
        return ops.igemm(x, self.w, self.b, **kw)


class Conv3x3:
    def __init__(self, s, stride=1, up=1, pad_n=False, pad=L.PAD_SAME):
        w, b = s.get("weight"), s.get("bias")
        self.n_real = w.shape[0]
        if pad_n:
            w, b = pad_rows(w), pad_rows(b)
        self.w = s.dev(pack_conv3x3(w))
        self.b = s.dev(f32(b))
        self.stride, self.up, self.pad = stride, up, pad

    def __call__(self, x, H, W, **kw):
        return ops.igemm(x, self.w, self.b, geom=ops.conv3x3_geom(H, W, self.stride, self.up, pad=self.pad), **kw)


class ConvT3:
    def __init__(self, s, pad_n=False):
        w, b = s.get("weight"), s.get("bias")
        if pad_n:
            w, b = pad_rows(w), pad_rows(b)
        self.w = s.dev(pack_conv3d_t3(w))
        self.b = s.dev(f32(b))

    def __call__(self, x, T, HW, **kw):
        return ops.igemm(x, self.w, self.b, geom=ops.convt3_geom(T, HW), **kw)


class GroupNorm:
    def __init__(self, s, eps):
        self.g, self.b, self.eps = s.dev(f32(s.get("weight"))), s.dev(f32(s.get("bias"))), eps

    def __call__(self, x, nframes, HW, frames_per_stat=1, silu=False):
        return ops.group_norm(x, self.g, self.b, nframes, HW, self.eps, frames_per_stat, silu)


class LayerNorm:
    def __init__(self, s, eps=1e-5):
        self.g, self.b, self.eps = s.dev(f32(s.get("weight"))), s.dev(f32(s.get("bias"))), eps

    def __call__(self, x, **kw):
        return ops.layer_norm(x, self.g, self.b, self.eps, **kw)


class GegluFF:
    """diffusers FeedForward(dim, activation_fn='geglu'): net.0.proj (-> 8C), net.2 (4C -> dim_out).

    ``norm``: the Sub of the LayerNorm that always precedes this feed-forward (BasicTransformerBlock.norm3,
    TemporalBasicTransformerBlock.norm_in / norm3).  At C = 320 (level 0) norm + projection + GELU gate + output projection +
    residuals then run as ONE launch (``fused``, csrc/ff320.hip) whose packed operands carry the norm's gain / bias."""

    def __init__(self, s, norm=None, norm_eps=1e-5):
        w, b = interleave_geglu(s.get("net.0.proj.weight"), s.get("net.0.proj.bias"))
        self.w1, self.b1 = s.dev(pack_linear(w)), s.dev(f32(b))
        self.out = Linear(s.sub("net.2"))
        self.pk = None
        if norm is not None and tuple(s.get("net.0.proj.weight").shape) == (2560, 320) and tuple(s.get("net.2.weight").shape) == (320, 1280):
            w1p, b1f, w2p = pack_ff320(s.get("net.0.proj.weight"), s.get("net.0.proj.bias"), s.get("net.2.weight"),
                                       norm.get("weight"), norm.get("bias"))
            self.pk = (s.dev(w1p), s.dev(b1f), s.dev(w2p), norm_eps)

    def __call__(self, x, **epilogue):
        h = ops.igemm(x, self.w1, self.b1, act=L.ACT_GEGLU_PAIR)
        return self.out(h, **epilogue)

    @property
    def can_fuse(self):
        return self.pk is not None and ops.FF_FUSED

    def fused(self, x, **kw):
        """x: the UN-normalised tokens; returns f16(f16(s_acc * ff(norm(x'))) + s1 * x' + s2 * r2) (ops.ff320)"""
        w1p, b1f, w2p, eps = self.pk
        return ops.ff320(x, w1p, b1f, w2p, self.out.b, eps=eps, **kw)


def drive(gen):
    """run a layer generator to its end and return its value"""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


def run_lockstep(gens, enter):
    """Advance several layer generators in turn, one layer each (``enter[i]()`` = the context -- HIP stream, communicator lane --
    network i is enqueued in), until all are exhausted; returns their values.  ONE host thread issues everything, so the order
    of the networks' launches and collectives is the program order: the same on every rank by construction."""
    vals, live = [None] * len(gens), list(range(len(gens)))
    while live:
        for i in list(live):
            with enter[i]():
                try:
                    next(gens[i])
                except StopIteration as stop:
                    vals[i] = stop.value
                    live.remove(i)
    return vals


def _sigmoid(v):
    return 1.0 / (1.0 + math.exp(-float(v)))


# ---------------------------------------------------------------------------------------------------------
class TembBatch:
    """All ``time_emb_proj`` layers of one network (44 in the SVD UNet, 20 in the ControlNet trunk; each a
    [Cout, 1280] x [B, 1280] product) as ONE implicit-GEMM launch + one cast per denoise step instead of two launches
    per layer.  Blocks built inside ``with TembBatch() as tb:`` register their projection and remember the column
    offset; the row width is padded to a common multiple of every Cout so that a layer's [B, Cout] slice of the
    [B, total] result is addressable as the igemm row vector with ``rv_mul = total / Cout`` (include/mofa_hip.h)."""
    _active = None

    def __init__(self):
        self.items, self.total, self.w, self.b = [], 0, None, None

    def __enter__(self):
        self._prev, TembBatch._active = TembBatch._active, self
        return self

    def __exit__(self, *exc):
        TembBatch._active = self._prev
        if exc[0] is None and self.items:
            mult = 1
            for lin in self.items:
                mult = mult * lin.n_real // math.gcd(mult, lin.n_real)
            pad = (-self.total) % mult
            ws, bs = [lin.w for lin in self.items], [lin.b for lin in self.items]
            if pad:
                ws.append(ws[0].new_zeros((pad, ws[0].shape[1])))
                bs.append(bs[0].new_zeros((pad,)))
            self.w, self.b = torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous()
            for lin in self.items:                      # the per-layer copies are no longer needed
                lin.w = lin.b = None
        return False

    def add(self, lin):
        assert lin.w.shape[0] == lin.n_real and lin.n_real % 64 == 0 and lin.b is not None
        off = self.total
        self.items.append(lin)
        self.total += lin.n_real
        return off

    def run(self, temb_act):
        return ops.cast_f16_to_f32(ops.igemm(temb_act, self.w, self.b)) if self.w is not None else None


class SpatioTemporalResBlock:
    """diffusers SpatioTemporalResBlock = ResnetBlock2D -> TemporalResnetBlock -> AlphaBlender."""

    def __init__(self, s, eps, temporal_eps=None, switch=False):
        sp, tp = s.sub("spatial_res_block"), s.sub("temporal_res_block")
        self.norm1, self.norm2 = GroupNorm(sp.sub("norm1"), eps), GroupNorm(sp.sub("norm2"), eps)
        self.conv1, self.conv2 = Conv3x3(sp.sub("conv1")), Conv3x3(sp.sub("conv2"))
        self.temb = Linear(sp.sub("time_emb_proj")) if sp.has("time_emb_proj.weight") else None
        tb = TembBatch._active
        self.temb_off = tb.add(self.temb) if (tb is not None and self.temb is not None) else None
        self.shortcut = Linear(sp.sub("conv_shortcut")) if sp.has("conv_shortcut.weight") else None
        te = temporal_eps if temporal_eps is not None else eps
        self.tnorm1, self.tnorm2 = GroupNorm(tp.sub("norm1"), te), GroupNorm(tp.sub("norm2"), te)
        self.tconv1, self.tconv2 = ConvT3(tp.sub("conv1")), ConvT3(tp.sub("conv2"))
        self.ttemb = Linear(tp.sub("time_emb_proj")) if tp.has("time_emb_proj.weight") else None
        self.ttemb_off = tb.add(self.ttemb) if (tb is not None and self.ttemb is not None) else None
        alpha = _sigmoid(s.get("time_mixer.mix_factor").reshape(-1)[0])
        self.alpha = (1.0 - alpha) if switch else alpha   # weight of x_spatial

    def __call__(self, x, c, H, W, out_stats=False, out=None):
        """out_stats: the caller's next op on the result is a GroupNorm (a transformer's ``norm`` / the next block's ``norm1``): the last
        convolution's epilogue emits its partial sums (``ops.igemm(stats=True)``), as the block's inner convolutions always do.
        out: where the block's result goes (a column slice of the decoder's next concat buffer) instead of a fresh tensor"""
        HW, N, T = H * W, c.N, c.T

        def tvec(lin, off):
            """fp32 [B, Cout] time-embedding vector of this layer + the row-vector index mapping that reads it"""
            if off is not None and c.temb_all is not None:
                return dict(rowvec=c.temb_all[:, off:off + lin.n_real], rv=(T * HW, c.temb_all.shape[1] // lin.n_real, 1, BIG))
            return dict(rowvec=ops.cast_f16_to_f32(lin(c.temb_act)), rv=(T * HW, 1, 1, BIG))
        h = self.norm1(x, N, HW, silu=True)
        if self.temb is not None:
            h = self.conv1(h, H, W, stats=True, **tvec(self.temb, self.temb_off))
        else:
            h = self.conv1(h, H, W, stats=True)
        h = self.norm2(h, N, HW, silu=True)
        xs = self.shortcut(x) if self.shortcut is not None else x
        par = c.par
        xs = self.conv2(h, H, W, r1=xs, s1=1.0, stats=par is None)       # ResnetBlock2D output
        if par is None:
            g = self.tnorm1(xs, N, HW, frames_per_stat=T, silu=True)
            if self.ttemb is not None:
                g = self.tconv1(g, T, HW, stats=True, **tvec(self.ttemb, self.ttemb_off))
            else:
                g = self.tconv1(g, T, HW, stats=True)
            g = self.tnorm2(g, N, HW, frames_per_stat=T, silu=True)
            # alpha*xs + (1-alpha)*(xs + conv) = xs + (1-alpha)*conv
            return self.tconv2(g, T, HW, s_acc=1.0 - self.alpha, r1=xs, s1=1.0, stats=out_stats, out=out)
        # ---- frames of the clip sharded over ranks (one CFG half per rank; mofa_video_amd/parallel.py) ----
        assert c.B == 1
        g = _sharded_norm_convt3(self.tnorm1, self.tconv1, xs, c, HW,
                                 **(tvec(self.ttemb, self.ttemb_off) if self.ttemb is not None else {}))
        return _sharded_norm_convt3(self.tnorm2, self.tconv2, g, c, HW, s_acc=1.0 - self.alpha, r1=xs, s1=1.0, out=out)


def _sharded_norm_convt3(norm, conv, x, c, HW, r1=None, out=None, **epi):
    """GroupNorm over the WHOLE clip (+ SiLU) -> (3,1,1) convolution, on the T = c.T frames of the clip this rank holds.
    One exchange group (parallel.FrameParallel): the raw boundary frames of x leave for the neighbour shards at once; the
    GroupNorm partials are all-gathered (the only wait of the compute stream) and every rank combines them itself; the own
    frames are normalised straight into the halo-extended buffer the convolution reads; the interior frames are convolved
    while the boundary frames travel, the received frames are normalised with the same statistics on arrival, and the first
    and last frame are convolved last.  Zero frames at the clip ends = the reference convolution's zero padding."""
    par, T = c.par, c.T
    M, Cc = T * HW, x.shape[1]
    nparts = ops.gn_nparts(HW, Cc)
    buf, own = par.part_buffer(nparts, x.device)
    ops.gn_partial_into(x, own, T, HW)
    halo = par.halo_begin(x, HW)
    par.gather_partials(buf, nparts)
    cnt = float(par.T_full) * HW * (Cc // 32)
    ext = torch.empty(((T + 2) * HW, Cc), dtype=x.dtype, device=x.device)
    ops.gn_apply_gathered(x, buf, cnt, norm.g, norm.b, norm.eps, ext[HW:(T + 1) * HW], T, HW, silu=True)
    if out is None:
        out = torch.empty((M, conv.w.shape[0]), dtype=x.dtype, device=x.device)
    geom = ops.convt3_geom(0, HW)                                   # unclipped: the halo rows lie before / after a.x

    def launch(f0, f1):
        """output frames [f0, f1) of the shard"""
        rr = r1[f0 * HW:f1 * HW] if r1 is not None else None
        ops.igemm(ext[(f0 + 1) * HW:], conv.w, conv.b, geom=geom, M=(f1 - f0) * HW, out=out[f0 * HW:f1 * HW], r1=rr, **epi)
    split = par.split_convs and T >= 4
    if split:
        launch(1, T - 1)
    fp, fn = halo.wait()
    for src, dst in ((fp, ext[:HW]), (fn, ext[(T + 1) * HW:])):
        if src is None:
            dst.zero_()
        else:
            ops.gn_apply_gathered(src, buf, cnt, norm.g, norm.b, norm.eps, dst, 1, HW, silu=True)
    if split:
        launch(0, 1)
        launch(T - 1, T)
    else:
        launch(0, T)
    return out


class CrossAttnVec:
    """attn2 with a single key token: output row vector to_out(to_v(ctx)) + bias per clip."""

    def __init__(self, s):
        self.to_v = Linear(s.sub("to_v"))
        self.to_out = Linear(s.sub("to_out.0"))

    def __call__(self, ctx16):
        return ops.cast_f16_to_f32(self.to_out(self.to_v(ctx16)))       # fp32 [B, C]


class SelfAttn:
    def __init__(self, s, heads, fold_q_scale=False, norm=None, norm_eps=1e-5):
        """fold_q_scale: multiply the Q projection by head_dim^-0.5 * log2(e) at load time (fp32, rounded to fp16 once,
        each weight independently -- the effect on Q averages over the C input channels and is far below Q's own fp16
        rounding), so the spatial attention kernel takes Q as is (``ops.attn_spatial(prescaled=True)``)."""
        wq = s.get("to_q.weight")
        self.heads = heads
        self.C = wq.shape[0]
        self.head_dim = self.C // heads      # diffusers: attention_head_dim = out_channels // num_attention_heads
        assert self.head_dim in (64, 128), f"unsupported head dim {self.head_dim}"
        self.q_prescaled = bool(fold_q_scale)
        if fold_q_scale:
            wq = (wq.float() * (self.head_dim ** -0.5 * ops.Q_FOLD_LOG2E)).to(wq.dtype)
        w = torch.cat([wq, s.get("to_k.weight"), s.get("to_v.weight")], 0)
        self.wqkv = s.dev(pack_linear(w))
        self.to_out = Linear(s.sub("to_out.0"))
        # level 0 (C = 320), spatial block: LayerNorm + the three projections as ONE launch (ops.lin320 with the norm's gain / bias
        # folded in): 491 against 555 us for mofa_layernorm_f16 + mofa_igemm_f16 at 460 800 tokens, 58 against 71 at a rank-of-8's
        # 64 512.  The kernel's other uses measured SLOWER than the 256x320 implicit-GEMM tile and are not wired: plain q | k | v 427
        # against 403 us, to_out + vector + residual 278 against 216, proj_in 180 against 154 (profiles/r06_lin320.log)
        self.qkv_pk = None
        if norm is not None and self.C == 320:
            wp, bp = pack_lin320(w, None, norm.get("weight"), norm.get("bias"))
            self.qkv_pk = (s.dev(wp), s.dev(bp), norm_eps)

    def qkv_normed(self, x):
        """q, k, v of LayerNorm(x) from the un-normalised tokens (level 0, one launch)"""
        wp, bp, eps = self.qkv_pk
        qkv = ops.lin320(x, wp, bp, norm=True, eps=eps)
        Cc = self.C
        return qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]

    @property
    def can_fuse_norm(self):
        return self.qkv_pk is not None and ops.LIN320

    def qkv(self, x):
        qkv = ops.igemm(x, self.wqkv)
        Cc = self.C
        return qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]

    def kv_into(self, x, out):
        """the K|V projection alone, written in place into ``out`` [rows, 2C] (a frame shard's slot of the all-gather
        buffer)"""
        return ops.igemm(x, self.wqkv[self.C:], out=out)

    def q(self, x):
        return ops.igemm(x, self.wqkv[:self.C])


class TransformerSpatioTemporal:
    """diffusers TransformerSpatioTemporalModel (one spatial + one temporal transformer block)."""
    _uid = 0

    def __init__(self, s, heads):
        TransformerSpatioTemporal._uid += 1
        self.uid = TransformerSpatioTemporal._uid
        self.heads = heads
        self.norm = GroupNorm(s.sub("norm"), 1e-6)
        self.proj_in, self.proj_out = Linear(s.sub("proj_in")), Linear(s.sub("proj_out"))
        b = s.sub("transformer_blocks.0")
        self.norm1, self.norm3 = LayerNorm(b.sub("norm1")), LayerNorm(b.sub("norm3"))
        self.attn1, self.attn2, self.ff = (SelfAttn(b.sub("attn1"), heads, fold_q_scale=True, norm=b.sub("norm1")), CrossAttnVec(b.sub("attn2")),
                                           GegluFF(b.sub("ff"), norm=b.sub("norm3")))
        t = s.sub("temporal_transformer_blocks.0")
        self.norm_in, self.tnorm1, self.tnorm3 = LayerNorm(t.sub("norm_in")), LayerNorm(t.sub("norm1")), LayerNorm(t.sub("norm3"))
        self.ff_in, self.tattn1, self.tattn2, self.tff = (GegluFF(t.sub("ff_in"), norm=t.sub("norm_in")), SelfAttn(t.sub("attn1"), heads),
                                                          CrossAttnVec(t.sub("attn2")), GegluFF(t.sub("ff"), norm=t.sub("norm3")))
        self.pos1, self.pos2 = Linear(s.sub("time_pos_embed.linear_1")), Linear(s.sub("time_pos_embed.linear_2"))
        self.alpha = _sigmoid(s.get("time_mixer.mix_factor").reshape(-1)[0])
        self.C = self.proj_in.w.shape[1]

    def _invariants(self, c):
        key = ("xf", self.uid)
        if key not in c.cache:
            dev = self.proj_in.w.device
            f0 = c.par.f0 if c.par is not None else 0
            tpos = ops.timestep_embedding(torch.arange(f0, f0 + c.T, dtype=torch.float32, device=dev), self.C)
            e = self.pos2(self.pos1(ops.cast_f32_to_f16(tpos), act=L.ACT_SILU))
            v_tm = self.tattn2(c.ctx16_all if c.ctx16_all is not None else c.ctx16)
            c.cache[key] = (self.attn2(c.ctx16), v_tm, ops.cast_f16_to_f32(e))
        return c.cache[key]

    def __call__(self, x, c, H, W, out_stats=False, out=None):
        HW, N, T, B = H * W, c.N, c.T, c.B
        v_sp, v_tm, pos = self._invariants(c)
        h = self.norm(x, N, HW)
        h = self.proj_in(h)
        # --- spatial BasicTransformerBlock ---
        fuse_qkv = self.attn1.can_fuse_norm and ops.lin320_fits(h.shape[0], 3 * self.C)
        q, k, v = self.attn1.qkv_normed(h) if fuse_qkv else self.attn1.qkv(self.norm1(h))
        a = ops.attn_spatial(q, k, v, N, self.heads, HW, head_dim=self.attn1.head_dim, prescaled=self.attn1.q_prescaled)
        h = self.attn1.to_out(a, r1=h, s1=1.0, rowvec=v_sp, rv=(T * HW, 1, 1, BIG))      # + attn1 + attn2
        # level 0 (C = 320): norm + feed-forward + residual(s) as one launch (GegluFF.fused); ff_in's launch also writes
        # norm1(f), the input of the temporal attention's projections (the lane pair that owns a token holds its whole row)
        h = self.ff.fused(h) if self.ff.can_fuse else self.ff(self.norm3(h), r1=h, s1=1.0)  # x_spatial
        # --- TemporalBasicTransformerBlock on h + pos[t] ---
        pos_rv = (HW, 1, 1, T)
        fn = None
        if self.ff_in.can_fuse:
            f_own = None
            if c.par is not None and c.par.kv_slots <= 32 and c.par.kv_inplace and c.par.gather_hidden:
                hid, f_own = c.par.kv_buffer(HW, self.C, h.device)            # norm1(f) straight into this shard's gather slot
            f, fn = self.ff_in.fused(h, pos=pos, HW=HW, T=T, ln_out=(self.tnorm1.g, self.tnorm1.b, f_own), ln_eps=self.tnorm1.eps)
        else:
            f = self.ff_in(self.norm_in(h, rowvec=pos, rv_div=HW, rv_mod=T), r1=h, s1=1.0, rowvec=pos, rv=pos_rv)
        if c.par is None:
            q, k, v = self.tattn1.qkv(fn if fn is not None else self.tnorm1(f))
            a = ops.attn_temporal(q, k, v, B, T, HW, self.heads, head_dim=self.tattn1.head_dim)
        else:
            # this rank holds T of the clip's T_full frames.  The K|V projection writes this shard's slot of the gather
            # buffer, one all_gather_into_tensor (asynchronous, RCCL's stream) fills the other slots while the Q
            # projection runs, and the attention kernel masks the padding frames of the shorter shards.
            Cc, par = self.C, c.par
            if par.kv_slots <= 32 and par.kv_inplace and par.gather_hidden:
                # gather the NORMED HIDDEN tokens (C columns) instead of K|V (2C): half the bytes over xGMI; every rank then
                # projects K|V for all key slots itself (4 x the rows of a 2C x C GEMM at 4 frame shards: ~0.1 ms at level 0
                # against ~1 ms of transfer saved).  LayerNorm writes this shard's slot of the gather buffer in place; the
                # padding slots of shorter shards hold whatever the allocator left -- their K|V rows are computed but masked,
                # never read by the attention kernel.
                if fn is None:
                    hid, own = par.kv_buffer(HW, Cc, f.device)
                    fn = self.tnorm1(f, out=own)
                work = par.kv_gather_begin(hid, HW)
                q = self.tattn1.q(fn)
                work.wait()
                kv = ops.igemm(hid, self.tattn1.wqkv[Cc:])
                a = ops.attn_temporal(q, kv[:, :Cc], kv[:, Cc:], 1, par.kv_slots, HW, self.heads,
                                      head_dim=self.tattn1.head_dim, Tq=T, key_mask=par.kv_mask)
            elif par.kv_slots <= 32 and par.kv_inplace:
                fn = fn if fn is not None else self.tnorm1(f)
                kv, own = par.kv_buffer(HW, 2 * Cc, fn.device)
                self.tattn1.kv_into(fn, own)
                work = par.kv_gather_begin(kv, HW)
                q = self.tattn1.q(fn)
                work.wait()
                a = ops.attn_temporal(q, kv[:, :Cc], kv[:, Cc:], 1, par.kv_slots, HW, self.heads,
                                      head_dim=self.tattn1.head_dim, Tq=T, key_mask=par.kv_mask)
            else:   # more than 32 key slots after padding (e.g. 31 frames over 3 shards), or the in-place path failed
                    # its self-check on this transport (parallel.FrameParallel.self_check): compact to T_full frames
                fn = fn if fn is not None else self.tnorm1(f)
                q, k, v = self.tattn1.qkv(fn)
                kv = torch.empty((q.shape[0], 2 * Cc), dtype=torch.float16, device=q.device)
                ops.copy2d(k, kv[:, :Cc])
                ops.copy2d(v, kv[:, Cc:])
                kv = par.gather_frames(kv, HW)
                a = ops.attn_temporal(q, kv[:, :Cc], kv[:, Cc:], 1, par.T_full, HW, self.heads,
                                      head_dim=self.tattn1.head_dim, Tq=T)
        # temporal cross-attention row vector.  diffusers 0.24.0 quirk: token row (b, s) of the GLOBAL batch takes the
        # context of batch (b*HW + s) mod B_global; v_tm has one row per global batch element.
        Bg = v_tm.shape[0]
        if not c.time_context_hw_major:
            quirk, tab = (T * HW, 1, 1, BIG), (v_tm if Bg == B else v_tm[c.half:c.half + B].contiguous())
        elif Bg == B:
            quirk, tab = (T * HW, HW, HW, B), v_tm
        else:   # one half per rank: local b = 0 is global batch c.half -> rotate the table by (half*HW) mod Bg
            rot = (c.half * HW) % Bg
            tab = torch.cat([v_tm[rot:], v_tm[:rot]], 0).contiguous() if rot else v_tm
            quirk = (T * HW, 0, HW, Bg)
        f = self.tattn1.to_out(a, r1=f, s1=1.0, rowvec=tab, rv=quirk)
        al = self.alpha
        if self.tff.can_fuse:
            m = self.tff.fused(f, s_acc=1.0 - al, s1=1.0 - al, r2=h, s2=al)                # AlphaBlender
        else:
            m = self.tff(self.tnorm3(f), s_acc=1.0 - al, r1=f, s1=1.0 - al, r2=h, s2=al)
        return self.proj_out(m, r1=x, s1=1.0, stats=out_stats, out=out)


# ---------------------------------------------------------------------------------------------------------
class DownBlock:
    def __init__(self, s, num_layers, heads, cross, downsample):
        eps = 1e-6 if cross else 1e-5
        self.resnets = [SpatioTemporalResBlock(s.sub(f"resnets.{i}"), eps) for i in range(num_layers)]
        self.attns = [TransformerSpatioTemporal(s.sub(f"attentions.{i}"), heads) for i in range(num_layers)] if cross else None
        self.down = Conv3x3(s.sub("downsamplers.0.conv"), stride=2) if downsample else None

    def layers(self, x, c, H, W):
        """generator: yields after every res (+ transformer) layer -- the points where ``run_lockstep`` switches to the other
        network of the step; the generator's return value is what ``__call__`` returns"""
        outs = []
        more = lambda i: i + 1 < len(self.resnets)               # the next layer's norm1 reads this layer's output
        for i, r in enumerate(self.resnets):
            x = r(x, c, H, W, out_stats=self.attns is not None or more(i))
            if self.attns is not None:
                x = self.attns[i](x, c, H, W, out_stats=more(i))
            outs.append((x, H, W))
            yield
        if self.down is not None:
            x = self.down(x, H, W)
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            outs.append((x, H, W))
        return x, H, W, outs

    def __call__(self, x, c, H, W):
        return drive(self.layers(x, c, H, W))


class MidBlock:
    def __init__(self, s, heads):
        self.res0 = SpatioTemporalResBlock(s.sub("resnets.0"), 1e-5)
        self.attn = TransformerSpatioTemporal(s.sub("attentions.0"), heads)
        self.res1 = SpatioTemporalResBlock(s.sub("resnets.1"), 1e-5)

    def layers(self, x, c, H, W):
        x = self.res0(x, c, H, W, out_stats=True)
        yield
        x = self.attn(x, c, H, W, out_stats=True)
        yield
        return self.res1(x, c, H, W)

    def __call__(self, x, c, H, W):
        return drive(self.layers(x, c, H, W))


class UpBlock:
    def __init__(self, s, num_layers, heads, cross, upsample):
        self.resnets = [SpatioTemporalResBlock(s.sub(f"resnets.{i}"), 1e-6) for i in range(num_layers)]
        self.attns = [TransformerSpatioTemporal(s.sub(f"attentions.{i}"), heads) for i in range(num_layers)] if cross else None
        self.up = Conv3x3(s.sub("upsamplers.0.conv"), up=2) if upsample else None

    def __call__(self, cat, skips, c, H, W):
        """cat: fp16 [rows, Cx + Cs] whose first Cx columns already hold the block's input -- its producer wrote them in place
        (``concat_target``); skips: list of (skip, ControlNet residual or None, multiplicity), consumed from the end.  The skip +
        multiplicity x residual sum of the reference (unet_..._controlnet.py:447-459) goes straight into the other Cs columns,
        and every layer writes its result into the columns of the NEXT concat buffer: no copy of either operand of a
        ``torch.cat`` (:478-483 of the reference's up-block loop in diffusers).  Returns that next buffer (or, when no skip is
        left, the plain result), H, W."""
        x = cat
        for i, r in enumerate(self.resnets):
            sk, res, m = skips.pop()
            Cx = cat.shape[1] - sk.shape[1]
            if res is not None and m:
                ops.axpby_out(res, sk, float(m), 1.0, out=cat[:, Cx:])
            else:
                ops.copy2d(sk, cat[:, Cx:])
            direct = i + 1 < len(self.resnets) or self.up is None    # this layer's result is the next concat's first operand
            nxt, tgt = concat_target(cat.shape[0], r.tconv2.w.shape[0], skips, cat) if direct else (None, None)
            x = r(cat, c, H, W, out_stats=self.attns is not None, out=tgt if self.attns is None else None)
            if self.attns is not None:
                x = self.attns[i](x, c, H, W, out=tgt)
            if nxt is not None:
                cat = x = nxt
        if self.up is not None:
            nxt, tgt = concat_target(cat.shape[0] * 4, self.up.w.shape[0], skips, cat)
            x = self.up(x, H, W, out=tgt)
            H, W = H * 2, W * 2
            if nxt is not None:
                x = nxt
        return x, H, W


def concat_target(rows, Cx, skips, like):
    """the buffer of the NEXT channel concat -- [rows, Cx + width of the skip on top of ``skips``] -- and its first Cx columns,
    which the producer of the concat's first operand writes in place; (None, None) when no skip is left"""
    if not skips:
        return None, None
    nxt = torch.empty((rows, Cx + skips[-1][0].shape[1]), dtype=like.dtype, device=like.device)
    return nxt, nxt[:, :Cx]


class TimeEmbedding:
    """time_proj/time_embedding + add_time_proj/add_embedding (unet_..._controlnet.py:386-417)."""

    def __init__(self, s, c0, add_dim):
        self.c0, self.add_dim = c0, add_dim
        self.l1, self.l2 = Linear(s.sub("time_embedding.linear_1")), Linear(s.sub("time_embedding.linear_2"))
        self.a1, self.a2 = Linear(s.sub("add_embedding.linear_1")), Linear(s.sub("add_embedding.linear_2"))

    def __call__(self, timesteps, added_time_ids):
        """timesteps fp32 [B]; added_time_ids fp32 [B,3] -> fp16 silu(emb) [B, 4*c0]"""
        B = timesteps.numel()
        t = ops.cast_f32_to_f16(ops.timestep_embedding(timesteps, self.c0))
        e = self.l2(self.l1(t, act=L.ACT_SILU))
        a = ops.timestep_embedding(added_time_ids.reshape(-1).contiguous(), self.add_dim).reshape(B, -1)
        a = ops.cast_f32_to_f16(a)
        e = self.a2(self.a1(a, act=L.ACT_SILU), r1=e, s1=1.0)
        return ops.cast_f32_to_f16(ops.silu_f32(ops.cast_f16_to_f32(e)))
