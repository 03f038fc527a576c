"""debug: run every igemm launch of a tiny VAE decode twice and report launches whose two results differ bitwise"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import TINY_VAE
from mofa_video_amd import ops, schema
from mofa_video_amd.vae import AutoencoderKLTemporalDecoder
orig = ops.igemm
bad = {}
ncalls = [0]
def dbg(x, w, *args, **kw):
    out = kw.get("out")
    snap = out.clone() if out is not None else None
    r1 = orig(x, w, *args, **kw).clone()
    if out is not None:
        out.copy_(snap)
    r2 = orig(x, w, *args, **kw)
    ncalls[0] += 1
    d = (r1.float() - r2.float()).abs().max().item()
    if d != 0:
        g = kw.get("geom", ops.PLAIN)
        key = (g.mode, g.stride, g.up, g.ksize, g.T, x.shape[0], tuple(w.shape), kw.get("bias") is not None, kw.get("r1") is not None,
               kw.get("r2") is not None, kw.get("rowvec") is not None, kw.get("act", 0), out is not None, r2.shape[1], r2.stride(0))
        bad[key] = max(bad.get(key, 0), d)
    return r2
ops.igemm = dbg
import mofa_video_amd.blocks as B, mofa_video_amd.vae as V
for mod in (B, V):
    if hasattr(mod, "ops"):
        mod.ops.igemm = dbg
sdv = schema.synthetic_state_dict(schema.vae_decoder_schema(**TINY_VAE), seed=2)
hv = AutoencoderKLTemporalDecoder(sdv, TINY_VAE, "cuda")
z = torch.randn(2, 4, 32, 32, device="cuda")
y = hv.decode(z, num_frames=2)
torch.cuda.synchronize()
print("igemm launches", ncalls[0], "nondeterministic kinds", len(bad))
for k, v in bad.items():
    print("mode,stride,up,ksize,T,xrows,wshape,bias,r1,r2,rv,act,out_given,ncols,ldo =", k, "max diff", v)
