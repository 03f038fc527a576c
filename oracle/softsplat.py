"""Oracle (test infrastructure): deterministic CPU forward splatting.

Follows MOFA-Video-Traj/models/softsplat.py:
  * ``softsplat(...)``      :232-274 (mode handling, 'avg' ones-channel, +1e-7 normalise)
  * ``softsplat_out`` kernel :284-345 (bilinear scatter; non-finite flow skipped; per-corner bounds)
The reference kernel uses fp32 atomicAdd (summation order undefined); here the four
corners are accumulated with ``index_add_`` in the fixed order NW, NE, SW, SE over
source pixels in raster order, in fp32 -- same operation set, deterministic order.
"""
import torch


def softsplat_sum(tenIn: torch.Tensor, tenFlow: torch.Tensor) -> torch.Tensor:
    """tenIn [N,C,H,W] fp32, tenFlow [N,2,H,W] fp32 -> forward-splatted sum [N,C,H,W]."""
    tenIn = tenIn.float().contiguous()
    tenFlow = tenFlow.float().contiguous()
    N, C, H, W = tenIn.shape
    assert tenFlow.shape == (N, 2, H, W)
    dev = tenIn.device
    out = torch.zeros(N, C, H * W, dtype=torch.float32, device=dev)
    gx = torch.arange(W, dtype=torch.float32, device=dev).view(1, 1, W).expand(N, H, W)
    gy = torch.arange(H, dtype=torch.float32, device=dev).view(1, H, 1).expand(N, H, W)
    fx = gx + tenFlow[:, 0]                       # :298
    fy = gy + tenFlow[:, 1]                       # :299
    finite = torch.isfinite(fx) & torch.isfinite(fy)   # :301-302
    fx = torch.where(finite, fx, torch.zeros_like(fx))
    fy = torch.where(finite, fy, torch.zeros_like(fy))
    x0 = torch.floor(fx)
    y0 = torch.floor(fy)
    x0i, y0i = x0.long(), y0.long()
    # weights :315-318 (SE corner coords = x0+1, y0+1)
    wnw = ((x0 + 1.0) - fx) * ((y0 + 1.0) - fy)
    wne = (fx - x0) * ((y0 + 1.0) - fy)
    wsw = ((x0 + 1.0) - fx) * (fy - y0)
    wse = (fx - x0) * (fy - y0)
    src = tenIn.view(N, C, H * W)
    for (cx, cy, w) in ((x0i, y0i, wnw), (x0i + 1, y0i, wne), (x0i, y0i + 1, wsw), (x0i + 1, y0i + 1, wse)):
        ok = finite & (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)      # :320-334
        idx = (cy.clamp(0, H - 1) * W + cx.clamp(0, W - 1)).view(N, H * W)
        wv = torch.where(ok, w, torch.zeros_like(w)).view(N, 1, H * W)
        for n in range(N):
            out[n].index_add_(1, idx[n], src[n] * wv[n])
    return out.view(N, C, H, W)


def softsplat(tenIn, tenFlow, tenMetric, strMode):
    """Same signature / assertions as the reference ``softsplat`` (softsplat.py:232-274)."""
    mode = strMode.split('-')[0]
    assert mode in ['sum', 'avg', 'linear', 'soft']
    if strMode == 'sum':
        assert tenMetric is None
    if strMode == 'avg':
        assert tenMetric is None
    if mode == 'linear':
        assert tenMetric is not None
    if mode == 'soft':
        assert tenMetric is not None
    tenIn = tenIn.float()
    if strMode == 'avg':
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif mode == 'linear':
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif mode == 'soft':
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)
    tenOut = softsplat_sum(tenIn, tenFlow)
    if mode in ['avg', 'linear', 'soft']:
        tenNormalize = tenOut[:, -1:, :, :]
        parts = strMode.split('-')
        if len(parts) == 1 or parts[1] == 'addeps':
            tenNormalize = tenNormalize + 0.0000001
        elif parts[1] == 'zeroeps':
            tenNormalize = tenNormalize.clone()
            tenNormalize[tenNormalize == 0.0] = 1.0
        elif parts[1] == 'clipeps':
            tenNormalize = tenNormalize.clip(0.0000001, None)
        tenOut = tenOut[:, :-1, :, :] / tenNormalize
    return tenOut
