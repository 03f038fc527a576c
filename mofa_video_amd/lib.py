"""ctypes binding of libmofa_hip.so (C ABI declared in include/mofa_hip.h).

Same calling discipline as the reference's CuPy launch of its softsplat kernel
(MOFA-Video-Traj/models/softsplat.py:341-345): raw device pointers from
``tensor.data_ptr()`` and the caller's torch stream.  There is NO fallback: a
missing library raises at import of this module's ``load()``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MOFA_HIP_LIB: another BUILD of the same library for same-box A/B runs of the test suite and the tools -- never a fallback)
LIB_PATH = os.environ.get("MOFA_HIP_LIB") or os.path.join(_HERE, "libmofa_hip.so")

MODE_PLAIN, MODE_CONV3X3, MODE_CONVT3 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GEGLU_PAIR, ACT_RELU, ACT_GELU = 0, 1, 2, 3, 4
PAD_SAME, PAD_TRAILING = 0, 1
TILE_AUTO, TILE_128X128, TILE_192X128, TILE_256X256, TILE_256X320 = 0, 2, 4, 5, 6


class IgemmArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p),
        ("r1", C.c_void_p), ("r2", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("Cin", C.c_int32),
        ("ldx", C.c_int32), ("ldo", C.c_int32), ("ldr1", C.c_int32), ("ldr2", C.c_int32),
        ("mode", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("stride", C.c_int32), ("up", C.c_int32), ("ksize", C.c_int32),
        ("T", C.c_int32), ("HW", C.c_int32),
        ("rv_div", C.c_int32), ("rv_mul", C.c_int32), ("rv_mod_in", C.c_int32), ("rv_mod_out", C.c_int32),
        ("act", C.c_int32),
        ("s_acc", C.c_float), ("s1", C.c_float), ("s2", C.c_float),
        ("dil", C.c_int32), ("pad", C.c_int32), ("tile", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("stats", C.c_void_p),
    ]


class Ff320Args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("pos", C.c_void_p), ("w1p", C.c_void_p), ("b1", C.c_void_p), ("w2p", C.c_void_p), ("b2", C.c_void_p),
        ("r2", C.c_void_p), ("out", C.c_void_p), ("out_ln", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
        ("M", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("ldr2", C.c_int32), ("ldoln", C.c_int32),
        ("HW", C.c_int32), ("T", C.c_int32),
        ("eps", C.c_float), ("s_acc", C.c_float), ("s1", C.c_float), ("s2", C.c_float), ("ln_eps", C.c_float),
        ("reserved", C.c_int32 * 4),
    ]


class Lin320Args(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wp", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("r1", C.c_void_p), ("out", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("ldr1", C.c_int32), ("norm", C.c_int32),
        ("rv_div", C.c_int32), ("rv_mul", C.c_int32), ("rv_mod_in", C.c_int32), ("rv_mod_out", C.c_int32),
        ("eps", C.c_float), ("s_acc", C.c_float), ("s1", C.c_float),
        ("reserved", C.c_int32 * 3),
    ]


_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes (every symbol include/mofa_hip.h declares; tests check they all resolve)
PROTOTYPES = {
    "mofa_version": [],
    "mofa_igemm_f16": [_P, _P],                     # (const mofa_igemm_args*: a byref(IgemmArgs) or the packed 192 bytes)
    "mofa_igemm_stats_ok": [_P],
    "mofa_ff320_f16": [_P, _P],
    "mofa_lin320_f16": [_P, _P],                    # (const mofa_lin320_args*: the packed 112 bytes)                     # (const mofa_ff320_args*: the packed 152 bytes)
    "mofa_attn_spatial_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "mofa_attn_spatial_qb_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "mofa_transpose_v_f16": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_attn_temporal_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "mofa_attn_temporal_masked_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, C.c_uint32, _P],
    "mofa_softmax_rows_f16": [_P, _I, _I, _I, _P],
    "mofa_gn_nparts": [_I, _I],
    "mofa_gn_partial_f16": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_gn_partial_from_stats": [_P, _P, _I, _I, _I, _P],
    "mofa_gn_finalize": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "mofa_gn_reduce": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_gn_finalize_sums": [_P, _P, _P, _P, _P, _I, _I, _I, C.c_double, _F, _P],
    "mofa_gn_apply_f16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "mofa_gn_apply_gathered_f16": [_P, _P, _I, C.c_double, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "mofa_affine_act_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mofa_layernorm_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _I, _I, _P],
    "mofa_axpby_f32": [_P, _P, _L, _F, _F, _P],
    "mofa_resize_nearest_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_mask_blend_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mofa_matting_blend_f16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mofa_subsample_tokens_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "mofa_axpby_f16": [_P, _P, _I, _I, _I, _I, _F, _F, _P],
    "mofa_axpby_out_f16": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P],
    "mofa_geglu_f16": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_copy2d_f16": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_silu_f32": [_P, _P, _I, _P],
    "mofa_cast_f32_to_f16": [_P, _P, _L, _P],
    "mofa_cast_f16_to_f32": [_P, _P, _L, _P],
    "mofa_nchw_f32_to_nhwc_f16": [_P, _P, _I, _I, _I, _I, _F, _P],
    "mofa_nhwc_f16_to_nchw_f32": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_timestep_embedding": [_P, _P, _I, _I, _P],
    "mofa_softsplat_ws_bytes": [_I, _I, _I],
    "mofa_softsplat_avg_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mofa_softsplat_scatter_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "mofa_softsplat_weight_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_softsplat_normalize_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_flow_downscale_f32": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_prepare_model_input": [_P, _P, _P, _I, _I, _I, _F, _P],
    "mofa_cfg_euler_step": [_P, _P, _I, _I, _I, _F, _F, _F, _F, _P],
    "mofa_prepare_model_input_dev": [_P, _P, _P, _I, _I, _I, _P, _P],
    "mofa_cfg_euler_step_dev": [_P, _P, _I, _I, _I, _P, _F, _F, _P],
    "mofa_step_select": [_P, _P, _P, _I, _P],
    "mofa_pool2d_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mofa_resize_bilinear_ac_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "mofa_resize_bilinear_ac_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_flow_expectation_f16": [_P, _P, _I, _I, _I, _I, _F, _P],
    "mofa_filter1d_reflect_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_resize_bicubic_ac_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "mofa_patchify_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "mofa_frames_postprocess_f32": [_P, _P, _I, _I, _I, _I, _P],
    "mofa_flow_to_image_ws_bytes": [_I, _I],
    "mofa_flow_to_image_u8": [_P, _P, _I, _I, _P, _P],
}
_RESTYPE = {"mofa_softsplat_ws_bytes": C.c_int64, "mofa_flow_to_image_ws_bytes": C.c_int64}

_lib = None


class MofaHipError(RuntimeError):
    pass


def load():
    """dlopen libmofa_hip.so; raises if it has not been built (no CPU/torch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MofaHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). mofa_video_amd has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


def stream_ptr():
    """the caller's current torch stream as a plain integer (ctypes converts it for a void* parameter; building c_void_p
    objects per argument was a measurable part of the host time per launch)"""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def check(rc, what):
    if rc != 0:
        raise MofaHipError(f"{what} failed with code {rc}")
