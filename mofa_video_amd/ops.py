"""Tensor-level wrappers over the C ABI (include/mofa_hip.h).

Activations are 2-D fp16 CUDA tensors, token-major: rows = frames*H*W tokens, columns =
channels, ``stride(1) == 1`` and ``stride(0)`` = leading dimension (column-sliced views are
fine).  PyTorch only provides device memory and the stream here; all arithmetic runs in
libmofa_hip.so.
"""
import ctypes as C
import struct
import threading

import torch

from . import lib as L

F16 = torch.float16
F32 = torch.float32


class LaunchTimer:
    """Optional per-launch HIP-event timing on the launch stream (bench.py's roofline leg).  When
    ``ops.TIMER`` is set, every timed entry point brackets its launch with two events recorded on the
    current stream and logs the launch's algorithmic work; ``summary()`` reads the events after a sync."""

    def __init__(self):
        self.rec = []
        self.tags = []

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, name, e0, flops=0.0, nbytes=0.0, tag=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.rec.append((name, e0, e1, flops, nbytes))
        if tag is not None:
            self.tags.append((tag, len(self.rec) - 1))

    def by_tag(self):
        torch.cuda.synchronize()
        out = {}
        for tag, i in self.tags:
            _, e0, e1, fl, _ = self.rec[i]
            d = out.setdefault(tag, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += fl
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, fl, nb in self.rec:
            d = out.setdefault(name, dict(launches=0, seconds=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["seconds"] += e0.elapsed_time(e1) * 1e-3
            d["flops"] += fl
            d["bytes"] += nb
        return out


TIMER = None
FORCE_TILE = L.TILE_AUTO     # tests set this to run whole models through one igemm tile (lib.TILE_*)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "token-major 2-D tensor with unit channel stride expected"
    return t.stride(0)


def _chk(t, dtype):
    assert t.is_cuda and t.dtype == dtype, f"expected cuda {dtype}, got {t.device} {t.dtype}"


class ConvGeom:
    """Geometry of an implicit-GEMM launch."""
    __slots__ = ("mode", "Hin", "Win", "Hout", "Wout", "stride", "up", "T", "HW", "ksize", "dil", "pad")

    def __init__(self, mode=L.MODE_PLAIN, Hin=0, Win=0, Hout=0, Wout=0, stride=1, up=1, T=0, HW=0, ksize=3, dil=1,
                 pad=L.PAD_SAME):
        self.mode, self.Hin, self.Win, self.Hout, self.Wout = mode, Hin, Win, Hout, Wout
        self.stride, self.up, self.T, self.HW, self.ksize, self.dil, self.pad = stride, up, T, HW, ksize, dil, pad


PLAIN = ConvGeom()


def conv3x3_geom(H, W, stride=1, up=1, ksize=3, dil=1, pad=L.PAD_SAME):
    """k x k convolution (k = 1, 3, 5, 7) with dilation dil and padding dil * (k // 2) on every side (PAD_SAME) or only
    after the last row / column (PAD_TRAILING: diffusers Downsample2D(padding=0), F.pad(x, (0, 1, 0, 1)) for k = 3)"""
    Hv, Wv = H * up, W * up
    if pad == L.PAD_TRAILING:
        span = dil * (ksize - 1) - dil * (ksize // 2)            # taps beyond the first that must fit without padding
        Ho = (Hv - 1 - span) // stride + 1
        Wo = (Wv - 1 - span) // stride + 1
    else:
        Ho = (Hv - 1) // stride + 1
        Wo = (Wv - 1) // stride + 1
    return ConvGeom(L.MODE_CONV3X3, H, W, Ho, Wo, stride, up, ksize=ksize, dil=dil, pad=pad)


def convt3_geom(T, HW):
    return ConvGeom(L.MODE_CONVT3, T=T, HW=HW)


IGEMM_WS_BYTES = 256 * 256 * 320 * 4     # split-K scratch: at most 256 partial tiles of 256 x 320 fp32 (84 MB)
_igemm_ws = {}
_igemm_ws_lock = threading.Lock()         # threads that share a stream (the virtual-rank tests) share its scratch: the split
                                          # launch and its fix-up must be enqueued back to back


def igemm_workspace(device):
    """the split-K scratch of the CURRENT stream on ``device`` (one per stream: launches of different streams overlap).  At most
    8 are kept (torch hands out side streams from a pool of 32): a buffer was allocated while its stream was current and is
    only ever used on it, so dropping it is stream-ordered by the caching allocator like any other temporary.  Look-up, insert
    and eviction happen under ``_igemm_ws_lock`` (virtual-rank threads share a stream and this dict); key and launch stream
    (``lib.stream_ptr``) are both taken from the CURRENT device, which must be ``device``."""
    cur = torch.cuda.current_device()
    assert device.index is None or device.index == cur, f"tensor on {device}, current device {cur}"
    key = (cur, torch.cuda.current_stream().cuda_stream)
    with _igemm_ws_lock:
        ws = _igemm_ws.pop(key, None)
        if ws is None:
            ws = torch.empty(IGEMM_WS_BYTES, dtype=torch.uint8, device=device)
            while len(_igemm_ws) >= 8:
                _igemm_ws.pop(next(iter(_igemm_ws)))
        _igemm_ws[key] = ws                                   # (re-inserted last: least recently used first)
    return ws


# mofa_igemm_args (include/mofa_hip.h, 192 bytes) packed in one call: the ~40 ctypes field stores of a Structure cost more
# host time than everything else in this wrapper, and the host's time per launch is what bounds a frame-sharded rank
_IGEMM_ARGS = struct.Struct("@7P22i3f3iPqP")
assert _IGEMM_ARGS.size == C.sizeof(L.IgemmArgs)
_TAPS_FIXED = {L.MODE_PLAIN: 1, L.MODE_CONVT3: 3}


GN_STATS = True          # A/B switch: False = every GroupNorm takes its partial sums with mofa_gn_partial_f16 (three passes)


def igemm(x, w, bias=None, geom=PLAIN, M=None, rowvec=None, rv=(1, 1, 1, 1 << 30), r1=None, s1=1.0, r2=None, s2=1.0,
          act=L.ACT_NONE, s_acc=1.0, out=None, tile=None, split_k=True, stats=False):
    """out[m,n] = act(s_acc*(conv/gemm + bias + rowvec[idx(m)]) + s1*r1 + s2*r2).  See include/mofa_hip.h.
    tile: one of lib.TILE_* to force the output tile (parity tests); default = ops.FORCE_TILE = the launcher's model.
    split_k: hand the launcher this stream's scratch buffer so that it may split a partial last round of tiles along K.
    stats: the caller's next op on the result is a GroupNorm: where the launch can (``mofa_igemm_stats_ok``) its epilogue also emits
    the pair sums of the outputs (``mofa_igemm_args.stats``); they travel as the attribute ``gn_stats`` of the FRESH output tensor
    (never of a caller-supplied ``out``) and ``group_norm`` consumes them instead of reading the activations for its partial sums."""
    lib = L.load()
    assert x.is_cuda and x.dtype is F16 and w.dtype is F16 and w.is_contiguous() and x.stride(1) == 1, "fp16 cuda, unit channel stride"
    N, Ktot = w.shape
    mode = geom.mode
    taps = geom.ksize * geom.ksize if mode == L.MODE_CONV3X3 else _TAPS_FIXED[mode]
    Cin = Ktot // taps
    assert Cin * taps == Ktot and x.shape[1] >= Cin, (x.shape, w.shape, taps)
    if M is None:
        if mode == L.MODE_CONV3X3:
            nimg = x.shape[0] // (geom.Hin * geom.Win)
            assert nimg * geom.Hin * geom.Win == x.shape[0]
            M = nimg * geom.Hout * geom.Wout
        else:
            M = x.shape[0]
    n_out = N // 2 if act == L.ACT_GEGLU_PAIR else N
    fresh = out is None
    if fresh:
        out = torch.empty((M, n_out), dtype=F16, device=x.device)
    else:
        assert out.dtype is F16 and out.is_cuda and out.shape[0] == M and out.shape[1] >= n_out and out.stride(1) == 1
    pb = pv = p1 = p2 = 0
    ld1 = ld2 = 0
    if bias is not None:
        assert bias.dtype is F32 and bias.numel() == N
        pb = bias.data_ptr()
    if rowvec is not None:
        # rows of a wider fp32 matrix are allowed: the kernel addresses row idx at idx*N, so the caller's rv_mul must
        # carry the row stride (TembBatch: stride = total, rv_mul = total / N)
        assert rowvec.dtype is F32 and rowvec.dim() == 2 and rowvec.shape[1] == N and rowvec.stride(1) == 1
        assert rowvec.is_contiguous() or (rowvec.stride(0) % N == 0 and rv[1] == rowvec.stride(0) // N), (rowvec.stride(), N, rv)
        pv = rowvec.data_ptr()
    if r1 is not None:
        assert r1.dtype is F16 and r1.shape[0] == M and r1.stride(1) == 1
        p1, ld1 = r1.data_ptr(), r1.stride(0)
    if r2 is not None:
        assert r2.dtype is F16 and r2.shape[0] == M and r2.stride(1) == 1
        p2, ld2 = r2.data_ptr(), r2.stride(0)
    if split_k:
        ws = igemm_workspace(x.device)
        pw, nw = ws.data_ptr(), ws.numel()
    else:
        pw = nw = 0
    tile_ = FORCE_TILE if tile is None else tile
    st = None
    if stats and GN_STATS and fresh and M % 64 == 0 and N % 320 == 0 and act == L.ACT_NONE and r2 is None and (r1 is None or s1 == 1.0):
        st = torch.empty((M // 64, N), dtype=F32, device=x.device)
    while True:
        args = _IGEMM_ARGS.pack(x.data_ptr(), w.data_ptr(), pb, pv, p1, p2, out.data_ptr(),
                                M, N, Cin, x.stride(0), out.stride(0), ld1, ld2, mode,
                                geom.Hin, geom.Win, geom.Hout, geom.Wout, geom.stride, geom.up, geom.ksize, geom.T, geom.HW,
                                rv[0], rv[1], rv[2], rv[3], act, s_acc, s1, s2, geom.dil, geom.pad,
                                tile_, pw, nw, st.data_ptr() if st is not None else 0)
        if st is None or lib.mofa_igemm_stats_ok(args) == 1:
            break
        st = None                                             # (alignment / geometry the 256x320 tile does not take: plain launch)
    t0 = TIMER.start() if TIMER is not None else None
    if split_k:
        with _igemm_ws_lock:
            rc = lib.mofa_igemm_f16(args, L.stream_ptr())
    else:
        rc = lib.mofa_igemm_f16(args, L.stream_ptr())
    if rc != 0:
        L.check(rc, "mofa_igemm_f16")
    if t0 is not None:
        TIMER.stop("igemm_f16_kernel", t0, flops=2.0 * M * N * Ktot,
                   tag=(mode, geom.stride, geom.up, M, N, Ktot, act))
    if st is not None:
        out.gn_stats = (st, _version(out), out.data_ptr())    # (group_norm ignores them if torch saw an in-place write since)
    elif not fresh:
        _written(out)
    return out


def _version(t):
    """torch's in-place write counter, or -1 for a tensor that keeps none (created under ``torch.inference_mode()``: reading
    ``_version`` raises there) -- such a tensor is guarded by its data pointer and this module's ``_written`` calls only"""
    return -1 if t.is_inference() else t._version


def _written(t):
    """an entry point of this module is about to write ``t`` in place: pair sums a producer attached describe the old values"""
    if t is not None and getattr(t, "gn_stats", None) is not None:
        del t.gn_stats


# ---- normalisation -------------------------------------------------------------------------------------
GN_FUSED_MAX_ENTRIES = 512     # partial entries (256 B each) a statistics set may have for the two-launch GroupNorm


def group_norm(x, gamma, beta, nframes, HW, eps, frames_per_stat=1, silu=False, out=None, C_=None):
    lib = L.load()
    _chk(x, F16)
    Cc = C_ if C_ is not None else x.shape[1]
    assert x.shape[0] == nframes * HW
    nparts = lib.mofa_gn_nparts(HW, Cc)
    part = torch.empty((nframes, nparts, 32, 2), dtype=F32, device=x.device)
    st = L.stream_ptr()
    pairs = getattr(x, "gn_stats", None)                      # pair sums the producing igemm epilogue emitted (igemm(stats=True))
    if pairs is not None:                                     # ... valid only for the very tensor object / values the producer left
        pairs, ver, dptr = pairs
        if ver != _version(x) or dptr != x.data_ptr():
            pairs = None
    if pairs is not None and HW % 64 == 0 and Cc % 64 == 0 and tuple(pairs.shape) == (nframes * HW // 64, Cc) == (x.shape[0] // 64, x.shape[1]):
        L.check(lib.mofa_gn_partial_from_stats(L.ptr(pairs), L.ptr(part), nframes, HW, Cc, st), "mofa_gn_partial_from_stats")
        del x.gn_stats                                            # one shot: they describe x as the producer left it
    else:
        L.check(lib.mofa_gn_partial_f16(L.ptr(x), L.ptr(part), nframes, HW, Cc, _ld(x), st), "mofa_gn_partial_f16")
    if out is None:
        out = torch.empty((x.shape[0], Cc), dtype=F16, device=x.device)
    else:
        _written(out)
    if frames_per_stat * nparts <= GN_FUSED_MAX_ENTRIES:
        # two launches: the applying kernel combines the partial sums of its statistics set itself
        L.check(lib.mofa_gn_apply_f16(L.ptr(x), L.ptr(part), L.ptr(gamma), L.ptr(beta), L.ptr(out), nframes, HW, Cc, _ld(x),
                                      _ld(out), frames_per_stat, eps, 1 if silu else 0, st), "mofa_gn_apply_f16")
        return out
    scale = torch.empty((nframes, Cc), dtype=F32, device=x.device)
    shift = torch.empty((nframes, Cc), dtype=F32, device=x.device)
    L.check(lib.mofa_gn_finalize(L.ptr(part), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift), nframes, HW,
                                 Cc, frames_per_stat, eps, st), "mofa_gn_finalize")
    L.check(lib.mofa_affine_act_f16(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(out), nframes, HW, Cc, _ld(x), _ld(out),
                                    1 if silu else 0, st), "mofa_affine_act_f16")
    return out


# GroupNorm whose statistics set spans frames on other ranks (parallel.FrameParallel): partials into the rank's rows of the
# gather buffer, [all-gather by the caller], then one applying launch per group of frames (own frames, received halo frames)
def gn_nparts(HW, Cc):
    return L.load().mofa_gn_nparts(HW, Cc)


def gn_partial_into(x, part_rows, nframes, HW):
    """part_rows: fp32 [nframes * nparts, 64] rows of the gather buffer (contiguous)"""
    lib = L.load()
    _chk(x, F16); _chk(part_rows, F32)
    assert x.shape[0] == nframes * HW and part_rows.is_contiguous()
    assert part_rows.shape[0] == nframes * lib.mofa_gn_nparts(HW, x.shape[1]) and part_rows.shape[1] == 64
    L.check(lib.mofa_gn_partial_f16(L.ptr(x), L.ptr(part_rows), nframes, HW, x.shape[1], _ld(x), L.stream_ptr()),
            "mofa_gn_partial_f16")


def gn_apply_gathered(x, part_all, count_per_group, gamma, beta, eps, out, nframes, HW, silu=False):
    """normalise ``nframes`` frames of x with the ONE statistics set whose gathered partial entries are ``part_all``"""
    lib = L.load()
    _chk(x, F16); _chk(out, F16); _chk(part_all, F32)
    assert x.shape[0] == nframes * HW == out.shape[0] and part_all.is_contiguous()
    _written(out)
    L.check(lib.mofa_gn_apply_gathered_f16(L.ptr(x), L.ptr(part_all), part_all.shape[0], float(count_per_group), L.ptr(gamma),
                                           L.ptr(beta), L.ptr(out), nframes, HW, x.shape[1], _ld(x), _ld(out), eps,
                                           1 if silu else 0, L.stream_ptr()), "mofa_gn_apply_gathered_f16")
    return out


def layer_norm(x, gamma, beta, eps=1e-5, rowvec=None, rv_div=1, rv_mod=1, out=None):
    lib = L.load()
    _chk(x, F16)
    M, Cc = x.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=F16, device=x.device)
    else:
        _written(out)
    L.check(lib.mofa_layernorm_f16(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(out), M, Cc, _ld(x), _ld(out), eps,
                                   L.ptr(rowvec), rv_div, rv_mod, L.stream_ptr()), "mofa_layernorm_f16")
    return out


# ---- fused level-0 feed-forward -------------------------------------------------------------------------
_FF320_ARGS = struct.Struct("@11P7i5f4i")
assert _FF320_ARGS.size == C.sizeof(L.Ff320Args)
FF_FUSED = True          # A/B switch: False = LayerNorm + two implicit GEMMs for every feed-forward (blocks.GegluFF)


def ff320(x, w1p, b1, w2p, b2, eps=1e-5, pos=None, HW=1, T=1, r2=None, s_acc=1.0, s1=1.0, s2=0.0, out=None,
          ln_out=None, ln_eps=1e-5):
    """out = f16(f16(s_acc * FF(LayerNorm(x'))) + s1 * x' + s2 * r2), x' = x + pos[(m / HW) % T]; FF = GEGLU feed-forward
    320 -> 1280 -> 320 with the norm's affine part folded into the packed projection (weights.pack_ff320).  ln_out = (gamma,
    beta[, buffer]): also returns LayerNorm(out) * gamma + beta (the norm in front of the next projection).  See
    mofa_ff320_f16 in include/mofa_hip.h."""
    lib = L.load()
    _chk(x, F16)
    M = x.shape[0]
    assert x.shape[1] == 320 and w1p.dtype is F16 and w2p.dtype is F16 and w1p.numel() == 2560 * 320 and w2p.numel() == 320 * 1280
    assert b1.dtype is F32 and b1.numel() == 2560 and b2.dtype is F32 and b2.numel() == 320
    if out is None:
        out = torch.empty((M, 320), dtype=F16, device=x.device)
    else:
        assert out.dtype is F16 and out.shape[0] == M and out.shape[1] >= 320
        _written(out)
    if pos is not None:
        assert pos.dtype is F32 and pos.is_contiguous() and pos.shape == (T, 320)
    if r2 is not None:
        assert r2.dtype is F16 and r2.shape[0] == M and r2.shape[1] >= 320
    yl = pg = pb = None
    if ln_out is not None:
        pg, pb = ln_out[0], ln_out[1]
        yl = ln_out[2] if len(ln_out) > 2 and ln_out[2] is not None else torch.empty((M, 320), dtype=F16, device=x.device)
        assert yl.dtype is F16 and yl.shape[0] == M and pg.dtype is F32 and pb.dtype is F32
        _written(yl)
    args = _FF320_ARGS.pack(x.data_ptr(), L.ptr(pos) or 0, w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                            L.ptr(r2) or 0, out.data_ptr(), L.ptr(yl) or 0, L.ptr(pg) or 0, L.ptr(pb) or 0,
                            M, _ld(x), _ld(out), _ld(r2) if r2 is not None else 0, _ld(yl) if yl is not None else 0, HW, T,
                            eps, s_acc, s1, s2 if r2 is not None else 0.0, ln_eps, 0, 0, 0, 0)
    t0 = TIMER.start() if TIMER is not None else None
    rc = lib.mofa_ff320_f16(args, L.stream_ptr())
    if rc != 0:
        L.check(rc, "mofa_ff320_f16")
    if t0 is not None:                                        # counted with the implicit GEMMs it replaces (roofline leg of bench.py)
        TIMER.stop("igemm_f16_kernel", t0, flops=2.0 * M * (2560 * 320 + 320 * 1280), tag=(9, 0, 0, M, 320, 320, 0))
    return out if ln_out is None else (out, yl)


_LIN320_ARGS = struct.Struct("@6P10i3f3i")
assert _LIN320_ARGS.size == C.sizeof(L.Lin320Args)
LIN320 = True            # A/B switch: False = mofa_layernorm_f16 + mofa_igemm_f16 for every 320-channel projection (blocks.Linear320)


def lin320_fits(M, ldo):
    """mofa_lin320_f16 addresses its output through a 32-bit buffer descriptor: (M + 256) rows of ldo halves must stay below 4 GB
    (include/mofa_hip.h); longer token lists take the LayerNorm + implicit-GEMM launches"""
    return (M + 256) * ldo * 2 < 0xFFFFFFFF


def lin320(x, wp, bias=None, norm=False, eps=1e-5, rowvec=None, rv=(1, 1, 1, 1 << 30), r1=None, s1=1.0, s_acc=1.0, out=None):
    """out = f16(f16(s_acc * (W . xhat + bias + rowvec[idx(m)])) + s1 * r1), xhat = LayerNorm(x) without affine part if ``norm`` (the
    norm's gain / bias folded into the packed operands, weights.pack_lin320) else x; N = 64 * wp.shape[0].  mofa_lin320_f16."""
    lib = L.load()
    _chk(x, F16)
    M, N = x.shape[0], wp.shape[0] * 64
    assert x.shape[1] == 320 and wp.dtype is F16 and wp.numel() == N * 320
    if out is None:
        out = torch.empty((M, N), dtype=F16, device=x.device)
    else:
        assert out.dtype is F16 and out.shape[0] == M and out.shape[1] >= N
        _written(out)
    if bias is not None:
        assert bias.dtype is F32 and bias.numel() == N
    if rowvec is not None:
        assert rowvec.dtype is F32 and rowvec.dim() == 2 and rowvec.shape[1] == N and rowvec.stride(1) == 1
        assert rowvec.is_contiguous() or (rowvec.stride(0) % N == 0 and rv[1] == rowvec.stride(0) // N), (rowvec.stride(), N, rv)
    if r1 is not None:
        assert r1.dtype is F16 and r1.shape[0] == M and r1.shape[1] >= N
    args = _LIN320_ARGS.pack(x.data_ptr(), wp.data_ptr(), L.ptr(bias) or 0, L.ptr(rowvec) or 0, L.ptr(r1) or 0, out.data_ptr(),
                             M, N, _ld(x), _ld(out), _ld(r1) if r1 is not None else 0, 1 if norm else 0,
                             rv[0], rv[1], rv[2], rv[3], eps, s_acc, s1, 0, 0, 0)
    t0 = TIMER.start() if TIMER is not None else None
    rc = lib.mofa_lin320_f16(args, L.stream_ptr())
    if rc != 0:
        L.check(rc, "mofa_lin320_f16")
    if t0 is not None:                                        # counted with the implicit GEMMs it replaces (roofline leg of bench.py)
        TIMER.stop("igemm_f16_kernel", t0, flops=2.0 * M * N * 320, tag=(10, 0, 0, M, N, 320, 0))
    return out


# ---- attention -----------------------------------------------------------------------------------------
Q_FOLD_LOG2E = 1.4426950408889634


def attn_spatial(q, k, v, nframes, heads, S, head_dim=64, scale=None, out=None, prescaled=False, query_blocks=0):
    """q/k/v: [nframes*S, heads*head_dim] column blocks (views allowed); scale defaults to head_dim**-0.5.
    prescaled: q already holds Q * head_dim**-0.5 * log2(e) (the constant folded into the Q projection weights).
    query_blocks: 0 = the launcher's rule, 1 / 2 = force 128- / 256-row workgroups (parity tests)."""
    lib = L.load()
    Cc = heads * head_dim
    scale = -1.0 if prescaled else (head_dim ** -0.5 if scale is None else scale)
    st = L.stream_ptr()
    if out is None:
        out = torch.empty((nframes * S, Cc), dtype=F16, device=q.device)
    else:
        _written(out)
    t0 = TIMER.start() if TIMER is not None else None
    L.check(lib.mofa_attn_spatial_qb_f16(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), nframes, heads, head_dim, S, _ld(q),
                                         _ld(k), _ld(v), _ld(out), scale, int(query_blocks), st), "mofa_attn_spatial_qb_f16")
    if t0 is not None:
        TIMER.stop("attn_spatial_kernel", t0, flops=4.0 * S * S * Cc * nframes)
    return out


def attn_temporal(q, k, v, nclips, T, HW, heads, head_dim=64, scale=None, out=None, Tq=None, key_mask=None):
    """k/v hold T frames per clip; q holds Tq <= T (Tq < T: frame-sharded clip with all-gathered K/V).  key_mask: bit j =
    key frame j exists (padding frames of uneven shards in the gathered buffer are masked, never read)."""
    lib = L.load()
    assert _ld(k) == _ld(v)
    Tq = T if Tq is None else Tq
    scale = head_dim ** -0.5 if scale is None else scale
    if out is None:
        out = torch.empty((nclips * Tq * HW, heads * head_dim), dtype=F16, device=q.device)
    else:
        _written(out)
    if key_mask is None:
        L.check(lib.mofa_attn_temporal_f16(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), nclips, Tq, T, HW, heads, head_dim,
                                           _ld(q), _ld(k), _ld(out), scale, L.stream_ptr()), "mofa_attn_temporal_f16")
    else:
        L.check(lib.mofa_attn_temporal_masked_f16(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), nclips, Tq, T, HW, heads, head_dim,
                                                  _ld(q), _ld(k), _ld(out), scale, int(key_mask), L.stream_ptr()),
                "mofa_attn_temporal_masked_f16")
    return out


def softmax_rows_(x):
    lib = L.load()
    L.check(lib.mofa_softmax_rows_f16(L.ptr(x), x.shape[0], x.shape[1], _ld(x), L.stream_ptr()), "mofa_softmax_rows_f16")
    return x


def transpose_v(v, nframes, ncb, S):
    """v [nframes*S, ncb*64] -> V^T [nframes*ncb*64, S]"""
    lib = L.load()
    vt = torch.empty((nframes * ncb * 64, S), dtype=F16, device=v.device)
    L.check(lib.mofa_transpose_v_f16(L.ptr(v), L.ptr(vt), nframes, ncb, S, _ld(v), L.stream_ptr()),
            "mofa_transpose_v_f16")
    return vt


# ---- element-wise ---------------------------------------------------------------------------------------
def axpby_(x, y, a=1.0, b=1.0):
    """y = a*x + b*y (in place on y)."""
    lib = L.load()
    assert x.shape == y.shape
    _written(y)
    L.check(lib.mofa_axpby_f16(L.ptr(x), L.ptr(y), x.shape[0], x.shape[1], _ld(x), _ld(y), a, b, L.stream_ptr()),
            "mofa_axpby_f16")
    return y


def axpby_out(x, y, a, b, out):
    """out = a*x + b*y (x, y untouched; out may be a column slice of a wider buffer)"""
    lib = L.load()
    assert x.shape == y.shape == out.shape
    _written(out)
    L.check(lib.mofa_axpby_out_f16(L.ptr(x), L.ptr(y), L.ptr(out), x.shape[0], x.shape[1], _ld(x), _ld(y), _ld(out), a, b,
                                   L.stream_ptr()), "mofa_axpby_out_f16")
    return out


def axpby_f32_(x, y, a=1.0, b=1.0):
    """y = a*x + b*y on contiguous fp32 tensors of equal size (in place on y)"""
    lib = L.load()
    _chk(x, F32); _chk(y, F32)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    L.check(lib.mofa_axpby_f32(L.ptr(x), L.ptr(y), x.numel(), a, b, L.stream_ptr()), "mofa_axpby_f32")
    return y


def resize_nearest_f32(x, h, w):
    """fp32 [n,H,W] -> [n,h,w], F.interpolate(mode='nearest') index rule"""
    lib = L.load()
    _chk(x, F32)
    n, H, W = x.shape
    y = torch.empty((n, h, w), dtype=F32, device=x.device)
    L.check(lib.mofa_resize_nearest_f32(L.ptr(x.contiguous()), L.ptr(y), n, H, W, h, w, L.stream_ptr()),
            "mofa_resize_nearest_f32")
    return y


def mask_blend(a, b, w, HW, out=None):
    """out = a*w[m % HW] + b*(1-w[m % HW])"""
    lib = L.load()
    assert a.shape == b.shape
    if out is None:
        out = torch.empty(a.shape, dtype=F16, device=a.device)
    else:
        _written(out)
    L.check(lib.mofa_mask_blend_f16(L.ptr(a), L.ptr(b), L.ptr(w), L.ptr(out), a.shape[0], a.shape[1], HW, _ld(a), _ld(b),
                                    _ld(out), L.stream_ptr()), "mofa_mask_blend_f16")
    return out


def matting_blend(warped, matting, logit, want_mask=True):
    lib = L.load()
    M, Cc = warped.shape
    out = torch.empty((M, Cc), dtype=F16, device=warped.device)
    mask = torch.empty((M,), dtype=F32, device=warped.device) if want_mask else None
    L.check(lib.mofa_matting_blend_f16(L.ptr(warped), L.ptr(matting), L.ptr(logit), L.ptr(out), L.ptr(mask), M, Cc,
                                       _ld(warped), _ld(matting), _ld(logit), _ld(out), L.stream_ptr()),
            "mofa_matting_blend_f16")
    return out, mask


def geglu(x, out=None):
    lib = L.load()
    M, C2 = x.shape
    Ch = C2 // 2
    if out is None:
        out = torch.empty((M, Ch), dtype=F16, device=x.device)
    else:
        _written(out)
    L.check(lib.mofa_geglu_f16(L.ptr(x), L.ptr(out), M, Ch, _ld(x), _ld(out), L.stream_ptr()), "mofa_geglu_f16")
    return out


def copy2d(src, dst):
    lib = L.load()
    assert src.shape == dst.shape
    _written(dst)
    L.check(lib.mofa_copy2d_f16(L.ptr(src), L.ptr(dst), src.shape[0], src.shape[1], _ld(src), _ld(dst), L.stream_ptr()),
            "mofa_copy2d_f16")
    return dst


def concat_channels(a, b):
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=F16, device=a.device)
    copy2d(a, out[:, :a.shape[1]])
    copy2d(b, out[:, a.shape[1]:])
    return out


def silu_f32(x):
    lib = L.load()
    y = torch.empty_like(x)
    L.check(lib.mofa_silu_f32(L.ptr(x), L.ptr(y), x.numel(), L.stream_ptr()), "mofa_silu_f32")
    return y


def cast_f32_to_f16(x):
    lib = L.load()
    _chk(x, F32)
    y = torch.empty(x.shape, dtype=F16, device=x.device)
    L.check(lib.mofa_cast_f32_to_f16(L.ptr(x.contiguous()), L.ptr(y), x.numel(), L.stream_ptr()), "mofa_cast_f32_to_f16")
    return y


def cast_f16_to_f32(x, out=None):
    lib = L.load()
    _chk(x, F16)
    assert x.is_contiguous()
    y = torch.empty(x.shape, dtype=F32, device=x.device) if out is None else out
    assert y.is_contiguous() and y.numel() == x.numel() and y.dtype == F32
    L.check(lib.mofa_cast_f16_to_f32(L.ptr(x), L.ptr(y), x.numel(), L.stream_ptr()), "mofa_cast_f16_to_f32")
    return y


def subsample_tokens(x, n, H, W, s):
    """token-major [n*H*W, C] -> [n*(H/s)*(W/s), C], nearest (F.interpolate(scale_factor=1/s))"""
    lib = L.load()
    _chk(x, F16)
    Cc = x.shape[1]
    y = torch.empty((n * (H // s) * (W // s), Cc), dtype=F16, device=x.device)
    L.check(lib.mofa_subsample_tokens_f16(L.ptr(x), L.ptr(y), n, H, W, s, Cc, _ld(x), _ld(y), L.stream_ptr()),
            "mofa_subsample_tokens_f16")
    return y


def nchw_to_tokens(x, ld=None, scale=1.0, out=None):
    """fp32 [n,C,H,W] -> fp16 token-major [n*H*W, ld] (zero-padded channels when ld > C).  out: an existing
    token-major view (e.g. a column block of a wider buffer) to write the C channels into."""
    lib = L.load()
    _chk(x, F32)
    n, Cc, H, W = x.shape
    if out is not None:
        assert out.shape[0] == n * H * W and out.shape[1] >= Cc
        _written(out)
        L.check(lib.mofa_nchw_f32_to_nhwc_f16(L.ptr(x.contiguous()), L.ptr(out), n, Cc, H * W, _ld(out), float(scale),
                                              L.stream_ptr()), "mofa_nchw_f32_to_nhwc_f16")
        return out
    ld = ld or Cc
    y = (torch.zeros if ld > Cc else torch.empty)((n * H * W, ld), dtype=F16, device=x.device)
    L.check(lib.mofa_nchw_f32_to_nhwc_f16(L.ptr(x.contiguous()), L.ptr(y), n, Cc, H * W, ld, float(scale), L.stream_ptr()),
            "mofa_nchw_f32_to_nhwc_f16")
    return y


def tokens_to_nchw(x, n, Cc, H, W):
    lib = L.load()
    _chk(x, F16)
    y = torch.empty((n, Cc, H, W), dtype=F32, device=x.device)
    L.check(lib.mofa_nhwc_f16_to_nchw_f32(L.ptr(x), L.ptr(y), n, Cc, H * W, _ld(x), L.stream_ptr()),
            "mofa_nhwc_f16_to_nchw_f32")
    return y


def timestep_embedding(t, dim):
    lib = L.load()
    _chk(t, F32)
    out = torch.empty((t.numel(), dim), dtype=F32, device=t.device)
    L.check(lib.mofa_timestep_embedding(L.ptr(t), L.ptr(out), t.numel(), dim, L.stream_ptr()), "mofa_timestep_embedding")
    return out


# ---- adapter warp ------------------------------------------------------------------------------------------
def softsplat_avg_tokens(feat, flow, H, W):
    """feat fp16 [H*W, C] token-major (one image); flow fp32 [nflows,2,H,W] -> fp16 [nflows*H*W, C]."""
    lib = L.load()
    _chk(feat, F16); _chk(flow, F32)
    nflows = flow.shape[0]
    Cc = feat.shape[1]
    ws = torch.empty((lib.mofa_softsplat_ws_bytes(nflows, H, W),), dtype=torch.uint8, device=feat.device)
    out = torch.empty((nflows * H * W, Cc), dtype=F16, device=feat.device)
    t0 = TIMER.start() if TIMER is not None else None
    L.check(lib.mofa_softsplat_avg_f16(L.ptr(feat), L.ptr(flow.contiguous()), L.ptr(out), L.ptr(ws), nflows, H, W, Cc,
                                       _ld(feat), _ld(out), L.stream_ptr()), "mofa_softsplat_avg_f16")
    if t0 is not None:       # real bytes per flow frame and pixel: C fp16 gathered + C fp16 written, the flow, <= 4 CSR entries
        TIMER.stop("softsplat_avg", t0, nbytes=float(nflows) * H * W * (4.0 * Cc + 8.0 + 64.0))
    return out


def softsplat_scatter_f32(tenIn, tenFlow):
    lib = L.load()
    _chk(tenIn, F32); _chk(tenFlow, F32)
    N, Cc, H, W = tenIn.shape
    out = torch.zeros_like(tenIn)
    L.check(lib.mofa_softsplat_scatter_f32(L.ptr(tenIn.contiguous()), L.ptr(tenFlow.contiguous()), L.ptr(out), N, Cc, H,
                                           W, L.stream_ptr()), "mofa_softsplat_scatter_f32")
    return out


def softsplat_weight_f32(tenIn, tenMetric, mode):
    """[in * w | w], w = metric (mode 1, 'linear') or exp(metric) (mode 2, 'soft'): fp32 [N, C + 1, H, W]"""
    lib = L.load()
    _chk(tenIn, F32); _chk(tenMetric, F32)
    N, Cc, H, W = tenIn.shape
    assert tuple(tenMetric.shape) == (N, 1, H, W)
    out = torch.empty((N, Cc + 1, H, W), dtype=F32, device=tenIn.device)
    L.check(lib.mofa_softsplat_weight_f32(L.ptr(tenIn.contiguous()), L.ptr(tenMetric.contiguous()), L.ptr(out), N, Cc, H, W, mode,
                                          L.stream_ptr()), "mofa_softsplat_weight_f32")
    return out


def softsplat_normalize_f32(summed, eps_mode):
    """summed fp32 [N, C + 1, H, W] -> [N, C, H, W] = summed[:, :C] / norm(summed[:, C:])"""
    lib = L.load()
    _chk(summed, F32)
    N, C1, H, W = summed.shape
    out = torch.empty((N, C1 - 1, H, W), dtype=F32, device=summed.device)
    L.check(lib.mofa_softsplat_normalize_f32(L.ptr(summed.contiguous()), L.ptr(out), N, C1 - 1, H, W, eps_mode, L.stream_ptr()),
            "mofa_softsplat_normalize_f32")
    return out


def flow_downscale(flow, s):
    lib = L.load()
    _chk(flow, F32)
    n, two, H, W = flow.shape
    assert two == 2
    out = torch.empty((n, 2, H // s, W // s), dtype=F32, device=flow.device)
    L.check(lib.mofa_flow_downscale_f32(L.ptr(flow.contiguous()), L.ptr(out), n, H, W, s, L.stream_ptr()),
            "mofa_flow_downscale_f32")
    return out


# ---- scheduler math ----------------------------------------------------------------------------------------
def prepare_model_input(latents, image_latents, out, sigma):
    lib = L.load()
    T, four, h, w = latents.shape
    L.check(lib.mofa_prepare_model_input(L.ptr(latents), L.ptr(image_latents), L.ptr(out), T, h * w, _ld(out),
                                         float(sigma), L.stream_ptr()), "mofa_prepare_model_input")
    return out


def cfg_euler_step_(latents, noise_pred, sigma, sigma_next, gmin, gmax):
    lib = L.load()
    T, four, h, w = latents.shape
    L.check(lib.mofa_cfg_euler_step(L.ptr(latents), L.ptr(noise_pred), T, h * w, _ld(noise_pred), float(sigma),
                                    float(sigma_next), float(gmin), float(gmax), L.stream_ptr()), "mofa_cfg_euler_step")
    return latents


STEP_SCALARS = 8


def prepare_model_input_dev(latents, image_latents, out, scal):
    """scal: fp32 [STEP_SCALARS] on the device (a row of the per-clip step table, or the graph's `cur` row)"""
    lib = L.load()
    T, four, h, w = latents.shape
    _chk(scal, F32)
    L.check(lib.mofa_prepare_model_input_dev(L.ptr(latents), L.ptr(image_latents), L.ptr(out), T, h * w, _ld(out), L.ptr(scal),
                                             L.stream_ptr()), "mofa_prepare_model_input_dev")
    return out


def cfg_euler_step_dev_(latents, noise_pred, scal, gmin, gmax):
    lib = L.load()
    T, four, h, w = latents.shape
    _chk(scal, F32)
    L.check(lib.mofa_cfg_euler_step_dev(L.ptr(latents), L.ptr(noise_pred), T, h * w, _ld(noise_pred), L.ptr(scal), float(gmin),
                                        float(gmax), L.stream_ptr()), "mofa_cfg_euler_step_dev")
    return latents


def step_select(table, counter, cur):
    lib = L.load()
    _chk(table, F32); _chk(cur, F32)
    assert counter.dtype == torch.int32 and table.is_contiguous() and table.shape[1] == STEP_SCALARS
    L.check(lib.mofa_step_select(L.ptr(table), L.ptr(counter), L.ptr(cur), table.shape[0], L.stream_ptr()), "mofa_step_select")


# ---- output stage ---------------------------------------------------------------------------
def frames_postprocess(frames, mode):
    """decoded fp32 frames [n,3,H,W] -> mode 0: fp32 [n,3,H,W] in [0,1]; 1: fp32 [n,H,W,3]; 2: uint8 [n,H,W,3]."""
    lib = L.load()
    _chk(frames, F32)
    n, c, H, W = frames.shape
    assert c == 3
    frames = frames.contiguous()
    if mode == 0:
        out = torch.empty_like(frames)
    else:
        out = torch.empty((n, H, W, 3), dtype=F32 if mode == 1 else torch.uint8, device=frames.device)
    L.check(lib.mofa_frames_postprocess_f32(L.ptr(frames), L.ptr(out), n, H, W, mode, L.stream_ptr()),
            "mofa_frames_postprocess_f32")
    return out


def flow_to_image(flow_hw2):
    """fp32 flow [H,W,2] -> uint8 Middlebury colour image [H,W,3]."""
    lib = L.load()
    _chk(flow_hw2, F32)
    H, W, two = flow_hw2.shape
    assert two == 2
    flow_hw2 = flow_hw2.contiguous()
    ws = torch.empty((lib.mofa_flow_to_image_ws_bytes(H, W),), dtype=torch.uint8, device=flow_hw2.device)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=flow_hw2.device)
    L.check(lib.mofa_flow_to_image_u8(L.ptr(flow_hw2), L.ptr(out), H, W, L.ptr(ws), L.stream_ptr()), "mofa_flow_to_image_u8")
    return out


# ---- CMP sparse-to-dense motion encoder pieces ----------------------------------------------
def pool2d(x, nimg, H, W, C, k, stride, pad=0, mode="max", out=None):
    """token-major fp16 [nimg*H*W, ld>=C] -> [nimg*Ho*Wo, ld_out]; nn.MaxPool2d / nn.AvgPool2d semantics."""
    lib = L.load()
    _chk(x, F16)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((nimg * Ho * Wo, C), dtype=F16, device=x.device)
    assert x.shape[0] == nimg * H * W and out.shape[0] == nimg * Ho * Wo
    _written(out)
    L.check(lib.mofa_pool2d_f16(L.ptr(x), L.ptr(out), nimg, H, W, C, _ld(x), _ld(out), k, stride, pad, 0 if mode == "max" else 1,
                                L.stream_ptr()), "mofa_pool2d_f16")
    return out, Ho, Wo


def resize_bilinear_ac(x, nimg, H, W, C, Ho, Wo, out=None):
    """token-major fp16 maps, F.interpolate(mode='bilinear', align_corners=True)"""
    lib = L.load()
    _chk(x, F16)
    if out is None:
        out = torch.empty((nimg * Ho * Wo, C), dtype=F16, device=x.device)
    assert x.shape[0] == nimg * H * W and out.shape[0] == nimg * Ho * Wo
    _written(out)
    L.check(lib.mofa_resize_bilinear_ac_f16(L.ptr(x), L.ptr(out), nimg, H, W, Ho, Wo, C, _ld(x), _ld(out), L.stream_ptr()),
            "mofa_resize_bilinear_ac_f16")
    return out


def resize_bilinear_ac_f32(x, Ho, Wo):
    """fp32 [..., H, W] planes -> [..., Ho, Wo], align_corners=True"""
    lib = L.load()
    _chk(x, F32)
    H, W = x.shape[-2:]
    x = x.contiguous()
    y = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), dtype=F32, device=x.device)
    L.check(lib.mofa_resize_bilinear_ac_f32(L.ptr(x), L.ptr(y), x.numel() // (H * W), H, W, Ho, Wo, L.stream_ptr()),
            "mofa_resize_bilinear_ac_f32")
    return y


def flow_expectation(logits, nimg, H, W, nbins, fmax):
    """fp16 logits [nimg*H*W, ld >= 2*nbins] -> fp32 flow [nimg, 2, H, W] (Fuser.convert_flow)"""
    lib = L.load()
    _chk(logits, F16)
    out = torch.empty((nimg, 2, H, W), dtype=F32, device=logits.device)
    L.check(lib.mofa_flow_expectation_f16(L.ptr(logits), L.ptr(out), nimg, H * W, _ld(logits), nbins, float(fmax), L.stream_ptr()),
            "mofa_flow_expectation_f16")
    return out


# ---- image conditioning front end -------------------------------------------------------------
def filter1d_reflect(x, taps, axis):
    """x fp32 [..., H, W]; taps fp32 [k] on the device; axis 1 = along W, 0 = along H (pipeline.py:587-610 _filter2d)"""
    lib = L.load()
    _chk(x, F32); _chk(taps, F32)
    assert x.is_contiguous() and taps.is_contiguous() and x.dim() >= 2
    H, W = x.shape[-2:]
    out = torch.empty_like(x)
    L.check(lib.mofa_filter1d_reflect_f32(L.ptr(x), L.ptr(out), L.ptr(taps), x.numel() // (H * W), H, W, taps.numel(), axis,
                                          L.stream_ptr()), "mofa_filter1d_reflect_f32")
    return out


def resize_bicubic_ac(x, Ho, Wo):
    """F.interpolate(x, (Ho, Wo), mode="bicubic", align_corners=True) on fp32 [..., H, W]"""
    lib = L.load()
    _chk(x, F32)
    assert x.is_contiguous() and x.dim() >= 2
    H, W = x.shape[-2:]
    out = torch.empty((*x.shape[:-2], Ho, Wo), dtype=F32, device=x.device)
    L.check(lib.mofa_resize_bicubic_ac_f32(L.ptr(x), L.ptr(out), x.numel() // (H * W), H, W, Ho, Wo, L.stream_ptr()),
            "mofa_resize_bicubic_ac_f32")
    return out


def patchify(x, p, ld):
    """fp32 [n, C, H, W] -> fp16 [n*(H/p)*(W/p), ld] rows of C*p*p patch values (+ zero columns up to ld)"""
    lib = L.load()
    _chk(x, F32)
    assert x.is_contiguous() and x.dim() == 4
    n, Cc, H, W = x.shape
    out = torch.empty((n * (H // p) * (W // p), ld), dtype=F16, device=x.device)
    L.check(lib.mofa_patchify_f16(L.ptr(x), L.ptr(out), n, Cc, H, W, p, ld, L.stream_ptr()), "mofa_patchify_f16")
    return out
