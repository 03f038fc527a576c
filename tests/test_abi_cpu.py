"""CPU tests of the boundary and host logic: the C-ABI library loads and exports every symbol include/mofa_hip.h
declares (no compute calls without a GPU), the ctypes prototypes cover exactly that set, the product package
never reaches into oracle/, and the host-side weight repacking / geometry helpers behave."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mofa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(mofa_\w+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from mofa_video_amd import _build, lib
    _build.build()                                   # hipcc cross-compiles for gfx950 without a GPU
    syms = _declared_symbols()
    assert len(syms) >= 25
    dll = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"libmofa_hip.so does not export {s}"
    assert sorted(lib.PROTOTYPES) == syms, set(lib.PROTOTYPES) ^ set(syms)
    assert lib.load().mofa_version() >= 100


def test_igemm_args_struct_layout_matches_header():
    from mofa_video_amd.lib import IgemmArgs
    # 7 pointers + 22 int32 + 3 float + 3 int32 (dil, pad, tile) = 56 + 88 + 12 + 12 = 168, + workspace pointer + int64 size + stats pointer
    assert ctypes.sizeof(IgemmArgs) == 192 and IgemmArgs.workspace.offset == 168 and IgemmArgs.stats.offset == 184
    assert IgemmArgs.M.offset == 56 and IgemmArgs.ksize.offset == 112 and IgemmArgs.s_acc.offset == 144


def test_argument_validation_without_gpu():
    """entry points reject bad arguments before touching the device"""
    from mofa_video_amd import lib
    l = lib.load()
    a = lib.IgemmArgs()
    assert l.mofa_igemm_f16(ctypes.byref(a), None) == -22          # null pointers
    assert l.mofa_attn_spatial_f16(None, None, None, None, 1, 1, 64, 8, 8, 8, 8, 8, 0.125, None) == -22
    assert l.mofa_gn_nparts(9216, 320) == 16 and l.mofa_gn_nparts(576, 1280) == 4
    assert l.mofa_softsplat_ws_bytes(24, 72, 128) > 24 * 72 * 128 * 44


def test_every_entry_point_rejects_null_arguments_without_gpu():
    """each of the header's compute entry points, called with NULL pointers and zero sizes, returns MOFA_EINVAL before any
    device call (there is no GPU in this container: a launch attempt would return MOFA_ELAUNCH or crash instead)"""
    from mofa_video_amd import lib
    l = lib.load()
    query = {"mofa_version", "mofa_gn_nparts", "mofa_softsplat_ws_bytes", "mofa_flow_to_image_ws_bytes", "mofa_igemm_stats_ok"}
    checked = 0
    for name, argtypes in lib.PROTOTYPES.items():
        if name in query:
            continue
        args = [None if t is ctypes.c_void_p else (0.0 if t in (ctypes.c_float, ctypes.c_double) else 0) for t in argtypes]
        assert getattr(l, name)(*args) == -22, name
        checked += 1
    assert checked == len(lib.PROTOTYPES) - len(query) and checked >= 50


def test_fused_level0_entry_points_validate_layout_without_gpu():
    """mofa_ff320_f16 / mofa_lin320_f16 (include/mofa_hip.h): struct sizes as documented, and every layout rule of the header is
    checked before the launch -- one violated rule at a time on otherwise valid (never dereferenced) addresses"""
    from mofa_video_amd import lib
    l = lib.load()
    assert ctypes.sizeof(lib.Ff320Args) == 152 and ctypes.sizeof(lib.Lin320Args) == 112
    A = 0x10000                                                     # 16-byte aligned, never touched: validation fails first

    def ff(**kw):
        a = lib.Ff320Args(x=A, w1p=A, b1=A, w2p=A, b2=A, out=A, M=256, ldx=320, ldo=320, eps=1e-5, s_acc=1.0)
        for k, v in kw.items():
            setattr(a, k, v)
        return l.mofa_ff320_f16(ctypes.byref(a), None)
    for bad in (dict(M=0), dict(ldx=312), dict(ldx=324), dict(ldo=316), dict(x=A + 8), dict(out=A + 2), dict(w1p=A + 4), dict(b2=A + 4),
                dict(w2p=None), dict(pos=A), dict(pos=A, HW=64), dict(pos=A + 4, HW=64, T=4), dict(r2=A, ldr2=0), dict(r2=A + 8, ldr2=320),
                dict(out_ln=A, ldoln=320), dict(out_ln=A, ln_gamma=A, ln_beta=A, ldoln=300), dict(out_ln=A, ln_gamma=A + 4, ln_beta=A, ldoln=320)):
        assert ff(**bad) == -22, bad

    def lin(**kw):
        a = lib.Lin320Args(x=A, wp=A, out=A, M=256, N=320, ldx=320, ldo=320, s_acc=1.0)
        for k, v in kw.items():
            setattr(a, k, v)
        return l.mofa_lin320_f16(ctypes.byref(a), None)
    for bad in (dict(M=0), dict(N=0), dict(N=96), dict(N=330), dict(ldx=312), dict(ldo=256), dict(N=960), dict(x=A + 8), dict(wp=A + 2),
                dict(bias=A + 4), dict(rowvec=A), dict(rowvec=A + 4, rv_div=1, rv_mod_in=1, rv_mod_out=1), dict(r1=A, ldr1=0),
                dict(r1=A + 8, ldr1=320), dict(r1=A, ldr1=324),
                dict(M=1 << 22, N=960, ldo=960)):                      # (M + 256) * ldo * 2 bytes beyond the 32-bit output descriptor
        assert lin(**bad) == -22, bad
    from mofa_video_amd import ops
    assert ops.lin320_fits(460800, 960) and ops.lin320_fits(2 * 97 * 9216, 960) and not ops.lin320_fits(1 << 22, 960)


def test_product_does_not_import_oracle_or_reference():
    pkg = os.path.join(ROOT, "mofa_video_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f
    for f in ("bench.py",):
        src = open(os.path.join(ROOT, f)).read()
        assert "/root/reference" not in src


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mofa_video_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.MofaHipError):
        lib.load()


def test_weight_packing():
    from mofa_video_amd.weights import interleave_geglu, pack_conv3d_t3, pack_conv3x3, pack_linear, pad_rows
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = pack_conv3x3(w)                                # Cin 3 -> 64
    assert p.shape == (2, 9 * 64) and p.dtype == torch.float16
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        assert torch.equal(p[:, tap * 64:tap * 64 + 3].float(), w[:, :, ky, kx])
        assert p[:, tap * 64 + 3:(tap + 1) * 64].abs().max() == 0
    w3 = torch.randn(4, 64, 3, 1, 1)
    p3 = pack_conv3d_t3(w3)
    assert torch.equal(p3[:, 64:128], w3[:, :, 1, 0, 0].half())
    assert pack_linear(torch.randn(8, 100)).shape == (8, 128)
    assert pad_rows(torch.ones(3, 5)).shape == (4, 5)
    Ch = 64
    wg = torch.arange(2 * Ch, dtype=torch.float32).reshape(2 * Ch, 1)
    wi, bi = interleave_geglu(wg, wg[:, 0])
    assert wi[:16, 0].tolist() == list(range(16)) and wi[16:32, 0].tolist() == list(range(Ch, Ch + 16))
    assert wi[32:48, 0].tolist() == list(range(16, 32))
    assert torch.equal(wi[:, 0], bi)


def test_conv_geometry():
    from mofa_video_amd import ops
    g = ops.conv3x3_geom(72, 128, stride=2)
    assert (g.Hout, g.Wout) == (36, 64)
    g = ops.conv3x3_geom(9, 16, stride=2)
    assert (g.Hout, g.Wout) == (5, 8)
    g = ops.conv3x3_geom(36, 64, up=2)
    assert (g.Hout, g.Wout) == (72, 128)
