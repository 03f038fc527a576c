"""Drop-in for the reference warp op ``softsplat(tenIn, tenFlow, tenMetric, strMode)``
(MOFA-Video-Traj/models/softsplat.py:232-274; kernel :284-345) on MI355X.

Only the modes the reference inference path uses are provided: ``'avg'`` with ``tenMetric=None`` (every
call site: svdxt_..._norefine.py:231, Hybrid/models/ldmk_ctrlnet.py:300, Hybrid/models/traj_ctrlnet.py:240)
and ``'sum'``.  ``'avg'`` runs the deterministic gather kernel on fp16 features (the adapter's features are
fp16 and the reference rounds the result back to fp16, so the rounding points coincide); ``'sum'`` runs the
literal fp32 atomicAdd scatter.  Same assertions as the reference; no CPU path (the reference asserts on
non-CUDA tensors too, softsplat.py:347-348).
"""
import torch

from . import ops


def softsplat(tenIn: torch.Tensor, tenFlow: torch.Tensor, tenMetric: torch.Tensor, strMode: str):
    assert strMode.split('-')[0] in ['sum', 'avg', 'linear', 'soft']
    if strMode == 'sum':
        assert tenMetric is None
    if strMode == 'avg':
        assert tenMetric is None
    if strMode.split('-')[0] == 'linear':
        assert tenMetric is not None
    if strMode.split('-')[0] == 'soft':
        assert tenMetric is not None
    assert tenIn.is_cuda and tenFlow.is_cuda, "softsplat: CUDA/HIP tensors required (as in the reference)"
    N, C, H, W = tenIn.shape
    assert tenFlow.shape == (N, 2, H, W)
    if strMode == 'sum':
        return ops.softsplat_scatter_f32(tenIn.float().contiguous(), tenFlow.float().contiguous())
    if strMode == 'avg':
        Cp = (C + 7) // 8 * 8
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=tenIn.device)
        for n in range(N):
            tok = ops.nchw_to_tokens(tenIn[n:n + 1].float().contiguous(), ld=Cp)
            w = ops.softsplat_avg_tokens(tok, tenFlow[n:n + 1].float().contiguous(), H, W)
            out[n:n + 1] = ops.tokens_to_nchw(w, 1, C, H, W)
        return out
    raise NotImplementedError(f"softsplat mode {strMode!r} is not on the MOFA inference path")
