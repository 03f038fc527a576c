"""mofa_lin320_f16 (csrc/lin320.hip: 320-input-channel linear layers, transposed form, optional LayerNorm in front) through the C ABI
against an fp32 PyTorch reference of the same op and against the launches it replaces (mofa_layernorm_f16 + mofa_igemm_f16).

Reference ops: diffusers 0.24.0 ``Attention.to_q / to_k / to_v`` behind ``norm1``, ``to_out[0]`` + residual + the single-key
cross-attention vector, ``proj_in`` of TransformerSpatioTemporalModel, as the reference builds them at 320 channels
(MOFA-Video-Traj/models/unet_spatio_temporal_condition_controlnet.py:169-232; restated in oracle/blocks.py).
Stated tolerance: |err| <= 2e-3 * (max|ref| + |ref|) element-wise (the implicit-GEMM tests' bar)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from mofa_video_amd import lib
    from mofa_video_amd import ops as o
    lib.load()
    return o


def _close(got, ref, what, tol=2e-3):
    err = (got.float() - ref).abs()
    bad = err > tol * (ref.abs().max() + ref.abs())
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} outside, max err {err.max().item():.3e} (scale {ref.abs().max().item():.3e})"


@pytest.mark.parametrize("N", [320, 960, 64])
@pytest.mark.parametrize("kind", ["plain", "norm", "bias+r1", "norm+bias", "rv+r1", "rv quirk+r1", "all"])
@pytest.mark.parametrize("M", [256 * 3, 256 * 260 + 77, 33, 256 * 355 + 5, 256 * 456])   # (last three: left-over round split 1 chunk / 8 chunks / whole tiles per item)
def test_lin320_vs_fp32_reference(ops, N, kind, M):
    from mofa_video_amd.weights import pack_lin320
    if M > 60000 and (N != 320 or kind not in ("all", "plain")):
        pytest.skip("the large-M case runs the widest and the plainest kind only")
    g = torch.Generator().manual_seed(N + len(kind))
    w = (torch.randn(N, 320, generator=g) * 320 ** -0.5).half()
    b = torch.randn(N, generator=g) * 0.3 if ("bias" in kind or kind == "all") else None
    norm = "norm" in kind or kind == "all"
    gamma, beta = (1 + 0.2 * torch.randn(320, generator=g), 0.2 * torch.randn(320, generator=g)) if norm else (None, None)
    wp, bp = pack_lin320(w, b, gamma, beta)
    wp, bp = wp.to(DEV), (bp.to(DEV) if bp is not None else None)
    gg = torch.Generator(device=DEV).manual_seed(M)
    x = (torch.randn(M, 320, generator=gg, device=DEV) * 1.2 + 0.3).half()
    kw = {}
    xf = x.float()
    ref = (F.layer_norm(xf, (320,), gamma.to(DEV), beta.to(DEV), 1e-5) if norm else xf) @ w.to(DEV).float().T
    if b is not None:
        ref = ref + b.to(DEV)
    if "rv" in kind or kind == "all":
        rv = (7, 3, 4, 5) if "quirk" in kind or kind == "all" else (100, 1, 1, 1 << 30)
        nrow = 5 if rv[3] == 5 else (M - 1) // 100 + 1
        rowvec = torch.randn(nrow, N, generator=gg, device=DEV) * 0.5
        m = torch.arange(M, device=DEV)
        ref = ref + rowvec[((m // rv[0]) * rv[1] + (m % rv[2])) % rv[3]]
        kw.update(rowvec=rowvec, rv=rv)
    s_acc = 0.75 if kind == "all" else 1.0
    ref = s_acc * ref
    if "r1" in kind or kind == "all":
        r1 = torch.randn(M, N, generator=gg, device=DEV).half()
        s1 = 0.5 if kind == "all" else 1.0
        ref = ref + s1 * r1.float()
        kw.update(r1=r1, s1=s1)
    got = ops.lin320(x, wp, bp, norm=norm, s_acc=s_acc, **kw)
    assert tuple(got.shape) == (M, N)
    _close(got, ref, f"lin320 N={N} {kind} M={M}")
    assert torch.equal(got, ops.lin320(x, wp, bp, norm=norm, s_acc=s_acc, **kw)), "not deterministic"


def test_lin320_strided_views_and_rows_beyond_m_untouched(ops):
    from mofa_video_amd.weights import pack_lin320
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(320, 320, generator=g) * 320 ** -0.5).half()
    wp, _ = pack_lin320(w)
    M = 256 * 2 + 100
    gg = torch.Generator(device=DEV).manual_seed(4)
    xb = torch.randn(M, 640, generator=gg, device=DEV).half()
    rb = torch.randn(M, 384, generator=gg, device=DEV).half()
    ob = torch.full((M + 64, 336), 5.0, device=DEV).half()
    x, r1 = xb[:, 128:448], rb[:, 8:328]
    got = ops.lin320(x, wp.to(DEV), r1=r1, out=ob[:M])
    ref = x.float() @ w.to(DEV).float().T + r1.float()
    _close(got[:, :320], ref, "strided")
    assert (ob[M:] == 5.0).all() and (ob[:, 320:] == 5.0).all()


def test_lin320_vs_layernorm_plus_igemm(ops):
    """norm1 + to_q|k|v (the one use the transformer blocks make of the kernel) through the two launches it replaces: rel-L2 <= 1e-3"""
    from mofa_video_amd import blocks
    g = torch.Generator().manual_seed(5)
    sd = {"a.to_q.weight": (torch.randn(320, 320, generator=g) * 0.05).half(), "a.to_k.weight": (torch.randn(320, 320, generator=g) * 0.05).half(),
          "a.to_v.weight": (torch.randn(320, 320, generator=g) * 0.05).half(), "a.to_out.0.weight": (torch.randn(320, 320, generator=g) * 0.05).half(),
          "a.to_out.0.bias": torch.randn(320, generator=g) * 0.1, "n.weight": 1 + 0.1 * torch.randn(320, generator=g), "n.bias": 0.1 * torch.randn(320, generator=g)}
    s = blocks.Sub(sd, "", DEV)
    att = blocks.SelfAttn(s.sub("a"), 5, fold_q_scale=True, norm=s.sub("n"))
    norm = blocks.LayerNorm(s.sub("n"))
    M = 9216 * 2
    gg = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(M, 320, generator=gg, device=DEV).half()
    a = torch.randn(M, 320, generator=gg, device=DEV).half()
    vec = torch.randn(2, 320, generator=gg, device=DEV)
    q1, k1, v1 = att.qkv_normed(x)
    q0, k0, v0 = att.qkv(norm(x))
    for name, u, v in (("q", q1, q0), ("k", k1, k0), ("v", v1, v0)):
        e = ((u.float() - v.float()).norm() / v.float().norm()).item()
        assert e < 1e-3, (name, e)
