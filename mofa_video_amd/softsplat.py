"""Drop-in for the reference warp op ``softsplat(tenIn, tenFlow, tenMetric, strMode)``
(MOFA-Video-Traj/models/softsplat.py:232-274; kernel :284-345) on MI355X.

The inference path uses ``'avg'`` with ``tenMetric=None`` only (every call site: svdxt_..._norefine.py:231,
Hybrid/models/ldmk_ctrlnet.py:300, Hybrid/models/traj_ctrlnet.py:240): it runs the deterministic gather kernel on fp16
features (the adapter's features are fp16 and the reference rounds the result back to fp16, so the rounding points
coincide).  ``'sum'`` runs the literal fp32 atomicAdd scatter; the metric-weighted modes ``'linear'`` / ``'soft'`` and the
``-addeps`` / ``-zeroeps`` / ``-clipeps`` suffixes (softsplat.py:243-270) are that scatter between a weighting and a
normalising kernel, all in fp32 as the reference's ``custom_fwd(cast_inputs=torch.float32)``.  Same assertions as the
reference; no CPU path (the reference asserts on non-CUDA tensors too, softsplat.py:347-348).
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
    if strMode.split('-')[0] == 'sum':       # 'sum' and 'sum-<suffix>': the raw splat, nothing is normalised (softsplat.py:252)
        return ops.softsplat_scatter_f32(tenIn.float().contiguous(), tenFlow.float().contiguous())
    if strMode == 'avg':
        Cp = (C + 7) // 8 * 8
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=tenIn.device)
        for n in range(N):
            tok = ops.nchw_to_tokens(tenIn[n:n + 1].float().contiguous(), ld=Cp)
            w = ops.softsplat_avg_tokens(tok, tenFlow[n:n + 1].float().contiguous(), H, W)
            out[n:n + 1] = ops.tokens_to_nchw(w, 1, C, H, W)
        return out
    # 'linear' / 'soft' (and 'avg-<suffix>', for which the reference concatenates nothing: softsplat.py:243 tests strMode == 'avg')
    parts = strMode.split('-')
    t = tenIn.float().contiguous()
    if parts[0] == 'linear':
        t = ops.softsplat_weight_f32(t, tenMetric.float().contiguous(), 1)
    elif parts[0] == 'soft':
        t = ops.softsplat_weight_f32(t, tenMetric.float().contiguous(), 2)
    summed = ops.softsplat_scatter_f32(t, tenFlow.float().contiguous())
    eps_mode = 0 if len(parts) == 1 or parts[1] == 'addeps' else {'zeroeps': 1, 'clipeps': 2}.get(parts[1], 3)
    return ops.softsplat_normalize_f32(summed, eps_mode)
