"""Every implicit-GEMM output tile (128x128, 192x128, 256x256 and 256x320 phase-pipelined) x every epilogue kind
x every addressing mode, ELEMENT-WISE against an fp32 PyTorch reference computed on the GPU, forced through
``mofa_igemm_args.tile`` so that the kernels the bench runs are the kernels compared here (the launcher's cost model
picks 128x128 for everything small).  Shapes are ragged in M and N, span several rounds of persistent workgroups
(> 512 tiles of 128x128, > 256 tiles of 256x256) and include the bench's own problem shapes.

Tolerance (stated): fp16 storage, fp32 accumulate -> |err| <= 2e-3 * max|ref| + 2e-3 * |ref| per element; kinds with
GEGLU / GELU add the 6e-5 of the erf polynomial (inside the bound).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TILES = {"128x128": 2, "192x128": 4, "256x256": 5, "256x320": 6}
PIPE = ("256x256", "256x320")          # the 8-wave phase-pipelined kernels (igemm8.hip, igemm320.hip)
# "rvu": a row vector that is constant over blocks of 1000 rows (idx = ((m / 1000) * 3) % 5) -- the time-embedding / per-clip
# pattern: waves whose 128 rows lie inside one block take the load-once path of the 256x256 kernel, waves that straddle a
# block boundary the per-row path, both in the same launch; plain "rv" (idx = ((m / 7) * 3 + m % 4) % 5) changes every row
KINDS = ["bias", "r1", "r1r2", "rv", "r1rv", "r1r2rv", "silu", "gelu", "rvu", "r1rvu", "r2rvu", "r1r2rvu", "rvusilu"]


def _close(out, ref, tol=2e-3, what=""):
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    scale = ref.abs().max().item() + 1e-12
    err = (out - ref).abs()
    bad = err > tol * scale + tol * ref.abs()
    if bad.any():
        i = int(err.argmax().item())
        raise AssertionError(f"{what}: {int(bad.sum().item())} / {bad.numel()} elements out of tolerance; max err "
                             f"{err.max().item():.4e} (scale {scale:.4e}) at row {i // ref.shape[1]} col {i % ref.shape[1]}")


def _h(*shape, seed=0, scale=1.0):
    """seeded fp16 test data, drawn on the GPU (the bench-shape cases hold half a billion elements)"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).half()


def _f(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator(device=DEV).manual_seed(seed), device=DEV)


@pytest.fixture(scope="module")
def ops():
    from mofa_video_amd import lib
    from mofa_video_amd import ops as o
    lib.load()
    return o


def _epilogue(kind, M, N, seed=100):
    """kwargs for ops.igemm + a function applying the same epilogue to the fp32 accumulator"""
    bias = _f(N, seed=seed)
    kw, s_acc = dict(), 0.75
    r1 = r2 = rowvec = None
    uni = "rvu" in kind
    rv = (1000, 3, 1, 5) if uni else (7, 3, 4, 5)       # idx = ((m / 1000) * 3) % 5  /  ((m / 7) * 3 + m % 4) % 5
    if "r1" in kind:
        r1 = _h(M, N, seed=seed + 1)
        kw.update(r1=r1, s1=0.5)
    if "r2" in kind:
        r2 = _h(M, N, seed=seed + 2)
        kw.update(r2=r2, s2=-1.25)
    if "rv" in kind:
        rowvec = _f(5, N, seed=seed + 3)
        kw.update(rowvec=rowvec, rv=rv)
    act = 1 if kind.endswith("silu") else (4 if kind == "gelu" else 0)
    kw.update(act=act, s_acc=s_acc)

    def apply(acc):
        y = acc + bias
        if rowvec is not None:
            m = torch.arange(M, device=DEV)
            y = y + rowvec[((m // 1000) * 3) % 5 if uni else ((m // 7) * 3 + m % 4) % 5]
        y = s_acc * y
        if r1 is not None:
            y = y + 0.5 * r1.float()
        if r2 is not None:
            y = y - 1.25 * r2.float()
        if act == 1:
            y = F.silu(y)
        elif act == 4:
            y = F.gelu(y)
        return y
    return bias, kw, apply


# M = 10317 (41 row tiles of 256, ragged), N = 2056 (9 column tiles, ragged, N % 8 == 0): 369 tiles of 256x256 > 256 CUs,
# 1377 tiles of 128x128 > 512 slots; K = 192 = 3 K tiles
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("tile", list(TILES))
def test_plain_gemm_every_tile_every_kind(ops, tile, kind):
    M, N, K = 10317, 2056, 192
    x, w = _h(M, K, seed=1), _h(N, K, seed=2, scale=0.1)
    bias, kw, apply = _epilogue(kind, M, N)
    refused = (kind == "rvusilu" and tile in PIPE) or (kind in ("r1r2rv", "r1r2rvu") and tile == "256x320")
    if refused:       # row vector + activation: 4-wave tiles only; row vector + two residuals: not on the 256x320 tile
        from mofa_video_amd.lib import MofaHipError
        with pytest.raises(MofaHipError):
            ops.igemm(x, w, bias, tile=TILES[tile], **kw)
        return
    out = ops.igemm(x, w, bias, tile=TILES[tile], **kw)
    _close(out, apply(x.float() @ w.float().t()), what=f"plain {tile} {kind}")


@pytest.mark.parametrize("tile", list(TILES))
def test_geglu_pair_every_tile(ops, tile):
    from mofa_video_amd.weights import interleave_geglu
    M, Cc = 33000, 128                                  # N = 8 C = 1024: 129 x 4 tiles of 256x256
    x = _h(M, Cc, seed=9)
    w = (torch.randn(8 * Cc, Cc, generator=torch.Generator().manual_seed(10)) * 0.2).half()
    b = torch.randn(8 * Cc, generator=torch.Generator().manual_seed(11))
    wi, bi = interleave_geglu(w, b)
    out = ops.igemm(x, wi.to(DEV).contiguous(), bi.to(DEV), act=2, tile=TILES[tile])
    h = x.float() @ w.float().t().to(DEV) + b.to(DEV)
    _close(out, h[:, :4 * Cc] * F.gelu(h[:, 4 * Cc:]), what=f"geglu {tile}")


@pytest.mark.parametrize("stride,up,pad", [(1, 1, 0), (2, 1, 0), (1, 2, 0), (2, 1, 1)])
@pytest.mark.parametrize("kind", ["bias", "r1rv"])
@pytest.mark.parametrize("tile", list(TILES))
def test_conv3x3_every_tile(ops, tile, kind, stride, up, pad):
    from mofa_video_amd.weights import pack_conv3x3
    n, Cin, Cout, H, W = 6, 64, 264, 37, 53            # ragged everywhere; 6 x 37 x 53 = 11766 rows at stride 1
    x = torch.randn(n, Cin, H, W, generator=torch.Generator().manual_seed(12)).half()
    w = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(13)) * 0.05).half()
    xt = x.permute(0, 2, 3, 1).reshape(n * H * W, Cin).contiguous().to(DEV)
    geom = ops.conv3x3_geom(H, W, stride=stride, up=up, pad=pad)
    M = n * geom.Hout * geom.Wout
    bias, kw, apply = _epilogue(kind, M, Cout)
    out = ops.igemm(xt, pack_conv3x3(w).to(DEV), bias, geom=geom, tile=TILES[tile], **kw)
    xi = x.float().to(DEV)
    if up == 2:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    if pad == 1:                                        # diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1))
        ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w.float().to(DEV), None, stride=stride, padding=0)
    else:
        ref = F.conv2d(xi, w.float().to(DEV), None, stride=stride, padding=1)
    assert (geom.Hout, geom.Wout) == tuple(ref.shape[2:])
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    _close(out, apply(ref), what=f"conv3x3 {tile} {kind} s{stride} up{up} pad{pad}")


@pytest.mark.parametrize("halo", [False, True])
@pytest.mark.parametrize("tile", list(TILES))
def test_convt3_every_tile(ops, tile, halo):
    from mofa_video_amd.weights import pack_conv3d_t3
    B, T, HW, Cc = 2, 7, 1153, 128                      # 16142 rows; clips of 7 frames
    x = torch.randn(B, Cc, T, HW, 1, generator=torch.Generator().manual_seed(15)).half()
    w = (torch.randn(Cc, Cc, 3, 1, 1, generator=torch.Generator().manual_seed(16)) * 0.1).half()
    xt = x[..., 0].permute(0, 2, 3, 1).reshape(B * T * HW, Cc).contiguous().to(DEV)
    if halo:
        # T = 0: no clipping, the caller supplies one halo frame before and after the rows (frame-sharded clips):
        # one clip of B * T frames between two zero frames == a (3,1,1) conv over all B * T frames
        ext = torch.zeros((B * T + 2) * HW, Cc, dtype=torch.float16, device=DEV)
        ext[HW:-HW] = xt
        bias, kw, apply = _epilogue("r1", B * T * HW, Cc)
        out = ops.igemm(ext[HW:], pack_conv3d_t3(w).to(DEV), bias, geom=ops.convt3_geom(0, HW), M=B * T * HW,
                        tile=TILES[tile], **kw)
        xr = x.float().permute(1, 0, 2, 3, 4).reshape(1, Cc, B * T, HW, 1).to(DEV)
        ref = F.conv3d(xr, w.float().to(DEV), None, padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0).reshape(-1, Cc)
    else:
        bias, kw, apply = _epilogue("rv", B * T * HW, Cc)
        out = ops.igemm(xt, pack_conv3d_t3(w).to(DEV), bias, geom=ops.convt3_geom(T, HW), tile=TILES[tile], **kw)
        ref = F.conv3d(x.float().to(DEV), w.float().to(DEV), None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, Cc)
    _close(out, apply(ref), what=f"conv(3,1,1) {tile} halo={halo}")


def test_identity_asymmetric_every_tile(ops):
    """A = I with an asymmetric B: a transposed fragment or output layout cannot pass"""
    K = 512
    x = torch.eye(K).half().to(DEV)
    w = (torch.arange(K * K, dtype=torch.float32).reshape(K, K) % 251 / 251.0 - 0.3).half().to(DEV)   # w[n][k]
    for name, t in TILES.items():
        _close(ops.igemm(x, w, tile=t), w.float().t(), tol=1e-3, what=f"identity {name}")


def test_closeness_check_sees_one_wrong_element():
    """the comparison itself: one element of 21 million off by 1 % of the scale must fail"""
    ref = torch.randn(10317, 2056, generator=torch.Generator().manual_seed(0))
    out = ref.clone()
    out[7777, 1234] += 0.01 * ref.abs().max()
    with pytest.raises(AssertionError):
        _close(out, ref, what="self-check")


def test_forced_phase_pipelined_tile_rejects_unaligned_rows(ops):
    from mofa_video_amd.lib import MofaHipError
    x, w = _h(300, 64, seed=1), _h(68, 64, seed=2)      # N = 68: N % 8 != 0 -> no 16-byte output rows
    ops.igemm(x, w)                                      # fine on the default path
    for t in PIPE:
        with pytest.raises(MofaHipError):
            ops.igemm(x, w, tile=TILES[t])


# ---- the bench's own problem shapes (BASELINE config 2), reference in row chunks on the GPU ------------------------------
def _chunked_ref(x, w, rows=32768):
    return torch.cat([x[i:i + rows].float() @ w.float().t() for i in range(0, x.shape[0], rows)])


@pytest.mark.parametrize("tile", ["192x128", "256x256", "256x320"])
def test_bench_shape_geglu_l0(ops, tile):
    from mofa_video_amd.weights import interleave_geglu
    M, Cc = 460800, 320
    x = _h(M, Cc, seed=21)
    w = (torch.randn(8 * Cc, Cc, generator=torch.Generator().manual_seed(22)) * 0.06).half()
    b = torch.randn(8 * Cc, generator=torch.Generator().manual_seed(23))
    wi, bi = interleave_geglu(w, b)
    out = ops.igemm(x, wi.to(DEV).contiguous(), bi.to(DEV), act=2, tile=TILES[tile])
    wd, bd = w.to(DEV), b.to(DEV)
    for i in range(0, M, 65536):
        h = x[i:i + 65536].float() @ wd.float().t() + bd
        _close(out[i:i + 65536], h[:, :4 * Cc] * F.gelu(h[:, 4 * Cc:]), what=f"bench GEGLU L0 {tile} rows {i}")


@pytest.mark.parametrize("tile", ["192x128", "256x256", "256x320"])
def test_bench_shape_geglu_l2(ops, tile):
    from mofa_video_amd.weights import interleave_geglu
    M, Cc = 28800, 1280
    x = _h(M, Cc, seed=24)
    w = (torch.randn(8 * Cc, Cc, generator=torch.Generator().manual_seed(25)) * 0.03).half()
    b = torch.randn(8 * Cc, generator=torch.Generator().manual_seed(26))
    wi, bi = interleave_geglu(w, b)
    out = ops.igemm(x, wi.to(DEV).contiguous(), bi.to(DEV), act=2, tile=TILES[tile])
    h = _chunked_ref(x, w.to(DEV)) + b.to(DEV)
    _close(out, h[:, :4 * Cc] * F.gelu(h[:, 4 * Cc:]), what=f"bench GEGLU L2 {tile}")


@pytest.mark.parametrize("tile", ["192x128", "256x256", "256x320"])
def test_bench_shape_conv3x3_l3_k11520(ops, tile):
    from mofa_video_amd.weights import pack_conv3x3
    n, Cc, H, W = 50, 1280, 9, 16                        # 7200 x 1280 x 11520 (K = 180 K tiles)
    x = torch.randn(n, Cc, H, W, generator=torch.Generator().manual_seed(27)).half().to(DEV)
    w = (torch.randn(Cc, Cc, 3, 3, generator=torch.Generator().manual_seed(28)) * 0.01).half().to(DEV)
    xt = x.permute(0, 2, 3, 1).reshape(n * H * W, Cc).contiguous()
    r1 = _h(n * H * W, Cc, seed=29)
    out = ops.igemm(xt, pack_conv3x3(w.cpu()).to(DEV), None, geom=ops.conv3x3_geom(H, W), r1=r1, s1=1.0, tile=TILES[tile])
    ref = F.conv2d(x.float(), w.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, Cc) + r1.float()
    _close(out, ref, what=f"bench conv3x3 L3 {tile}")


@pytest.mark.parametrize("tile", ["192x128", "256x256", "256x320"])
def test_bench_shape_ff_out_l0_two_residuals(ops, tile):
    """460800 x 320 x 1280 with the AlphaBlender epilogue (s_acc, r1, r2) of the temporal feed-forward"""
    M, N, K = 460800, 320, 1280
    x, w = _h(M, K, seed=30), _h(N, K, seed=31, scale=0.03)
    bias = _f(N, seed=32)
    r1, r2 = _h(M, N, seed=33), _h(M, N, seed=34)
    out = ops.igemm(x, w, bias, s_acc=0.4, r1=r1, s1=0.4, r2=r2, s2=0.6, tile=TILES[tile])
    for i in range(0, M, 65536):
        s = slice(i, i + 65536)
        ref = 0.4 * (x[s].float() @ w.float().t() + bias) + 0.4 * r1[s].float() + 0.6 * r2[s].float()
        _close(out[s], ref, what=f"bench ff-out L0 {tile} rows {i}")


@pytest.mark.parametrize("tile", ["256x256", "256x320"])
def test_bench_shape_conv3x3_l0_time_embedding(ops, tile):
    """460800 x 320 x 2880: ResnetBlock2D conv1 of level 0 with the per-frame time-embedding row vector (uniform over every
    wave's rows: 9216 rows per frame) -- the shape that reads X once on the 256x320 tile"""
    from mofa_video_amd.weights import pack_conv3x3
    n, Cc, H, W = 50, 320, 72, 128
    x = torch.randn(n, Cc, H, W, generator=torch.Generator(device=DEV).manual_seed(50), device=DEV).half()
    w = (torch.randn(Cc, Cc, 3, 3, generator=torch.Generator().manual_seed(51)) * 0.02).half().to(DEV)
    xt = x.permute(0, 2, 3, 1).reshape(n * H * W, Cc).contiguous()
    bias, rowvec = _f(Cc, seed=52), _f(2, Cc, seed=53)
    out = ops.igemm(xt, pack_conv3x3(w.cpu()).to(DEV), bias, geom=ops.conv3x3_geom(H, W), rowvec=rowvec,
                    rv=(25 * H * W, 1, 1, 1 << 30), tile=TILES[tile])
    for f0 in range(0, n, 5):                            # (5 divides 25: a chunk never straddles the two clips)
        ref = F.conv2d(x[f0:f0 + 5].float(), w.float(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, Cc)
        ref = ref + bias + rowvec[f0 // 25]
        _close(out[f0 * H * W:(f0 + 5) * H * W], ref, what=f"bench conv3x3 L0 + temb {tile} frames {f0}")


@pytest.mark.parametrize("kind", ["bias", "r1", "r1r2", "rv", "r1rvu", "silu"])
@pytest.mark.parametrize("M,N,K", [(10317, 640, 2560), (520 * 256 - 77, 320, 2880), (7200, 1280, 3840)])
def test_split_k_remainder_tiles_256x320(ops, M, N, K, kind):
    """The 256x320 tile cuts the tiles of a partial last round of workgroups into K slices when a workspace is supplied (fp32
    partial tiles + a fix-up launch): 82 tiles x 3 slices; 2 full rounds + 8 tiles x 5 slices; 116 tiles x 2 -- same result as the whole-tile path within the
    tolerance, element-wise against fp32, bit-identical run to run (the slices are added in a fixed order)."""
    x, w = _h(M, K, seed=61), _h(N, K, seed=62, scale=0.05)
    bias, kw, apply = _epilogue(kind, M, N)
    ref = apply(_chunked_ref(x, w))
    whole = ops.igemm(x, w, bias, tile=TILES["256x320"], split_k=False, **kw)
    split = ops.igemm(x, w, bias, tile=TILES["256x320"], split_k=True, **kw)
    _close(whole, ref, what=f"whole tiles {kind}")
    _close(split, ref, what=f"split-K {kind}")
    assert not torch.equal(whole, split)                     # (the split path really ran: fp32 summation order differs)
    assert torch.equal(split, ops.igemm(x, w, bias, tile=TILES["256x320"], split_k=True, **kw))


@pytest.mark.parametrize("kind", ["r1 s1=1", "r1 s1=0.5", "r1r2", "r1 + uniform row vector"])
def test_split_and_whole_tiles_round_alike(ops, kind):
    """round-4 advice: every tile of a launch must round the same way -- the whole 256x320 tiles round s_acc * acc to fp16 and then
    add the residual (the reference's fp16 module semantics), so the split-K remainder tiles (fix-up kernel) must too.  With
    operands whose products and sums are EXACT in fp32 (small integers x multiples of 2^-4) the summation order cannot matter,
    so a split and a whole-tile launch have to agree bit for bit."""
    M, N, K = 10317, 640, 2560                                  # 82 tiles: a partial round -> remainder tiles are split
    g = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randint(-3, 4, (M, K), generator=g, device=DEV).half()
    w = (torch.randint(-2, 3, (N, K), generator=g, device=DEV).float() / 16).half()
    bias = torch.randint(-64, 65, (N,), generator=g, device=DEV).float() / 16
    kw = dict(s_acc=0.75, r1=_h(M, N, seed=78), s1=1.0)
    if kind == "r1 s1=0.5":
        kw["s1"] = 0.5
    elif kind == "r1r2":
        kw.update(r2=_h(M, N, seed=79), s2=0.25)
    elif kind.endswith("row vector"):
        kw.update(rowvec=torch.randint(-64, 65, (3, N), generator=g, device=DEV).float() / 16, rv=(4096, 1, 1, 3))
    whole = ops.igemm(x, w, bias, tile=TILES["256x320"], split_k=False, **kw)
    split = ops.igemm(x, w, bias, tile=TILES["256x320"], split_k=True, **kw)
    acc = x.float() @ w.float().t() + bias
    if "rowvec" in kw:
        acc = acc + kw["rowvec"][(torch.arange(M, device=DEV) // 4096) % 3]
    ref = (0.75 * acc).half().float() + kw["s1"] * kw["r1"].float() + (kw["s2"] * kw["r2"].float() if "r2" in kw else 0.0)
    d = (whole.float() - split.float()).abs().max().item()
    assert torch.equal(whole, split), f"{kind}: split-K remainder tiles round differently from whole tiles (max diff {d:.3e})"
    assert (whole.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("kind", ["r1 s1=1", "r1 s1=0.5", "r1r2", "r1 + per-row row vector", "r1 + uniform row vector"])
def test_all_tiles_round_residual_adds_alike(ops, kind):
    """round-5 advice: the same layer must give the same bits whichever tile the cost model picks (the pick depends on shape, CU
    count, workspace, sharded vs unsharded M).  Every tile kernel rounds s_acc * (acc + bias + rowvec) to fp16 BEFORE the residual
    add (include/mofa_hip.h).  Operands exact in fp32 => summation order cannot matter => all four tiles agree bit for bit, and
    with the literal double-rounding formula."""
    M, N, K = 2309, 640, 512                                    # ragged M: tail rows on every tile height
    g = torch.Generator(device=DEV).manual_seed(91)
    x = torch.randint(-3, 4, (M, K), generator=g, device=DEV).half()
    w = (torch.randint(-2, 3, (N, K), generator=g, device=DEV).float() / 16).half()
    bias = torch.randint(-64, 65, (N,), generator=g, device=DEV).float() / 16
    # residuals on a 2^-6 grid, |r| <= 8: x0 + s1 r1 + s2 r2 is exact in fp32 in any association
    r1 = (torch.randint(-512, 513, (M, N), generator=g, device=DEV).float() / 64).half()
    kw = dict(s_acc=0.75, r1=r1, s1=1.0)
    if kind == "r1 s1=0.5":
        kw["s1"] = 0.5
    elif kind == "r1r2":
        kw.update(r2=(torch.randint(-512, 513, (M, N), generator=g, device=DEV).float() / 64).half(), s2=0.25)
    elif kind.endswith("row vector"):
        rv = (1000, 3, 1, 5) if "uniform" in kind else (7, 3, 4, 5)
        kw.update(rowvec=torch.randint(-64, 65, (5, N), generator=g, device=DEV).float() / 16, rv=rv)
    acc = x.float() @ w.float().t() + bias
    if "rowvec" in kw:
        m = torch.arange(M, device=DEV)
        d, mul, mi, mo = kw["rv"]
        acc = acc + kw["rowvec"][((m // d) * mul + (m % mi)) % mo]
    ref = ((0.75 * acc).half().float() + kw["s1"] * r1.float() + (kw["s2"] * kw["r2"].float() if "r2" in kw else 0.0)).half()
    outs = {name: ops.igemm(x, w, bias, tile=t, split_k=False, **kw) for name, t in TILES.items()}
    for name, o in outs.items():
        dmax = (o.float() - ref.float()).abs().max().item()
        assert torch.equal(o, ref), f"{kind}: tile {name} differs from round16(s_acc*acc) + residuals (max diff {dmax:.3e})"


@pytest.mark.parametrize("mode", ["conv3x3", "conv3x3_s2", "convt3", "convt3_halo"])
def test_split_k_convolutions_slices_start_inside_a_tap(ops, mode):
    """split-K on the implicit-GEMM convolutions: the K slices of a remainder tile start in the MIDDLE of a tap (9 or 3 taps of
    Cin / 64 K tiles each, cut into 4-6 slices), so the producer cursors set up (tap, K tile within the tap) from the slice start"""
    from mofa_video_amd.weights import pack_conv3d_t3, pack_conv3x3
    g = torch.Generator().manual_seed(70)
    if mode.startswith("conv3x3"):
        n, Cin, Cout, H, W = 6, 256, 320, 37, 53              # K = 2304 = 36 K tiles, 4 per tap
        stride = 2 if mode.endswith("s2") else 1
        x = torch.randn(n, Cin, H, W, generator=g).half()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.03).half()
        xt = x.permute(0, 2, 3, 1).reshape(n * H * W, Cin).contiguous().to(DEV)
        geom = ops.conv3x3_geom(H, W, stride=stride)
        M = n * geom.Hout * geom.Wout
        bias, kw, apply = _epilogue("r1rvu", M, Cout)
        wk = pack_conv3x3(w).to(DEV)
        ref = F.conv2d(x.float().to(DEV), w.float().to(DEV), None, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
        call = lambda sk: ops.igemm(xt, wk, bias, geom=geom, tile=TILES["256x320"], split_k=sk, **kw)   # noqa: E731
    else:
        B, T, HW, Cc = 2, 5, 1153, 1024                        # K = 3072 = 48 K tiles, 16 per tap
        x = torch.randn(B, Cc, T, HW, 1, generator=g).half()
        w = (torch.randn(320, Cc, 3, 1, 1, generator=g) * 0.03).half()
        xt = x[..., 0].permute(0, 2, 3, 1).reshape(B * T * HW, Cc).contiguous().to(DEV)
        wk = pack_conv3d_t3(w).to(DEV)
        M = B * T * HW
        bias, kw, apply = _epilogue("r1", M, 320)
        if mode.endswith("halo"):
            ext = torch.zeros((B * T + 2) * HW, Cc, dtype=torch.float16, device=DEV)
            ext[HW:-HW] = xt
            xr = x.float().permute(1, 0, 2, 3, 4).reshape(1, Cc, B * T, HW, 1).to(DEV)
            ref = F.conv3d(xr, w.float().to(DEV), None, padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0).reshape(-1, 320)
            call = lambda sk: ops.igemm(ext[HW:], wk, bias, geom=ops.convt3_geom(0, HW), M=M, tile=TILES["256x320"], split_k=sk, **kw)   # noqa: E731
        else:
            ref = F.conv3d(x.float().to(DEV), w.float().to(DEV), None, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(-1, 320)
            call = lambda sk: ops.igemm(xt, wk, bias, geom=ops.convt3_geom(T, HW), tile=TILES["256x320"], split_k=sk, **kw)   # noqa: E731
    whole, split = call(False), call(True)
    _close(whole, apply(ref), what=f"{mode} whole tiles")
    _close(split, apply(ref), what=f"{mode} split-K")
    assert not torch.equal(whole, split)                     # the split path ran
    assert torch.equal(split, call(True))


@pytest.mark.parametrize("tile", list(TILES))
def test_repeat_launches_bit_identical_every_tile(ops, tile):
    """a race in the staged K loop (LDS-DMA landing under a fragment read) or a dropped epilogue term shows up as a
    run-to-run difference at full occupancy"""
    M, N, K = 115200, 640, 2560
    x, w = _h(M, K, seed=40), _h(N, K, seed=41, scale=0.03)
    r1 = _h(M, N, seed=42)
    first = ops.igemm(x, w, None, r1=r1, s1=1.0, tile=TILES[tile]).clone()
    for _ in range(6):
        assert torch.equal(ops.igemm(x, w, None, r1=r1, s1=1.0, tile=TILES[tile]), first)
    _close(first[:32768], x[:32768].float() @ w.float().t() + r1[:32768].float(), what=f"repeat {tile}")
