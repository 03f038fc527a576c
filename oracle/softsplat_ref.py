"""Oracle (test infrastructure): ctypes binding of oracle/_ref/libsoftsplat_ref.so -- the reference's own
``softsplat_out`` kernel body (MOFA-Video-Traj/models/softsplat.py:284-345) compiled for the host by
``oracle/build_ref.py`` -- plus the reference's 'avg' wrapper arithmetic (softsplat.py:240-270)."""
import ctypes as C
import os

import torch

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libsoftsplat_ref.so")


def available():
    return os.path.exists(LIB)


def softsplat_out_ref(tenIn, tenFlow):
    lib = C.CDLL(LIB)
    fn = lib.softsplat_out_host
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    fn.restype = C.c_int
    tenIn = tenIn.float().contiguous()
    tenFlow = tenFlow.float().contiguous()
    N, Cc, H, W = tenIn.shape
    out = torch.zeros_like(tenIn)
    rc = fn(tenIn.data_ptr(), tenFlow.data_ptr(), out.data_ptr(), N, Cc, H, W)
    assert rc == 0
    return out


def softsplat_avg_ref(tenIn, tenFlow):
    """reference ``softsplat(tenIn, tenFlow, None, 'avg')`` with the kernel replaced by its host build"""
    tenIn = torch.cat([tenIn.float(), tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]]).float()], 1)
    tenOut = softsplat_out_ref(tenIn, tenFlow)
    return tenOut[:, :-1, :, :] / (tenOut[:, -1:, :, :] + 0.0000001)
