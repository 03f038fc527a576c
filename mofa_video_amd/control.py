"""Control-signal rasterisers: host-side conversion of user trajectories / facial landmarks into the sparse
flow + mask the CMP encoder consumes, with the reference's function names and array conventions.  Plain numpy / torch on
the host, as in the reference (these are format conversions of a few dozen points, not kernels).

  interpolate_trajectory, divide_points_afterinterpolate, get_sparseflow_and_mask_forward   Traj/run_gradio.py:41-86, :162-177
  tracking_points_to_drags                                                                   Traj/run_gradio.py:487-535 (glue of `run`)
  merge_inmask_outmask, controlnet_flow_from_drags                                           Traj/run_gradio.py:290-330 (forward_sample)
  sample_optical_flow, get_sparse_flow                                                       Keypoint/utils/utils.py:81-119"""
import numpy as np
import torch
from scipy.interpolate import PchipInterpolator


def interpolate_trajectory(points, n_points):
    """PCHIP through the user's control points (parameter = index / (len-1)), resampled at n_points."""
    pts = np.asarray(points, dtype=np.float64)
    t = np.linspace(0, 1, len(pts))
    s = np.linspace(0, 1, n_points)
    return list(zip(PchipInterpolator(t, pts[:, 0])(s), PchipInterpolator(t, pts[:, 1])(s)))


def divide_points_afterinterpolate(resized_all_points, motion_brush_mask):
    """tracks [K,T,2] (x,y) split by whether their START pixel lies inside the motion brush (mask == 255)."""
    pts = np.asarray(resized_all_points)
    inside = [motion_brush_mask[int(p[0][1])][int(p[0][0])] == 255 for p in pts]
    return (np.array([p for p, i in zip(pts, inside) if i]), np.array([p for p, i in zip(pts, inside) if not i]))


def get_sparseflow_and_mask_forward(resized_all_points, n_steps, H, W, is_backward_flow=False):
    """[K, n_steps+1, 2] tracks -> flow [n_steps,H,W,2] / mask [n_steps,H,W]: at every track's start pixel, step i holds
    the INTEGER displacement to its (i+1)-th point; tracks starting on the same pixel add up."""
    pts = np.asarray(resized_all_points)
    flow = np.zeros((n_steps, H, W, 2))
    mask = np.zeros((n_steps, H, W))
    sign = -1 if is_backward_flow is True else 1
    for track in pts:
        x0, y0 = int(track[0][0]), int(track[0][1])
        for i in range(n_steps):
            flow[i, y0, x0] += np.int64(track[i + 1] - track[0]) * sign
            mask[i, y0, x0] += 1
    return flow, mask


def tracking_points_to_drags(tracking_points, width, height, model_length, motion_brush_mask, work=384,
                             original_size=None):
    """The glue of DragNUWA-style `run`: user tracks (pixel coordinates at original_size) -> in-brush / out-of-brush sparse
    drags at the CMP working size.  Returns dict(drag_in, mask_in, drag_out, mask_out [1,T-1,work,work(,2)] tensors,
    in_flag, out_flag)."""
    ow, oh = original_size if original_size is not None else (width, height)
    tracks_work = [[(int(x * work / ow), int(y * work / oh)) for x, y in tr] for tr in tracking_points]
    pts = np.array([interpolate_trajectory(tr, model_length) for tr in tracks_work])
    brush = np.asarray(motion_brush_mask)
    if brush.shape != (work, work):
        # cv2.resize(mask, (work, work), cv2.INTER_NEAREST) passes the flag as dst, i.e. it is a BILINEAR resize in the
        # reference (run_gradio.py:394); brushes already at the working size avoid the ambiguity
        raise ValueError("pass the motion brush mask at the CMP working size")
    inm, outm = divide_points_afterinterpolate(pts, brush)
    n = model_length - 1
    out = {}
    for name, group in (("in", inm), ("out", outm)):
        if group.shape[0] != 0:
            f, m = get_sparseflow_and_mask_forward(group, n, work, work)
        else:
            f, m = np.zeros((n, work, work, 2)), np.zeros((n, work, work))
        out["drag_" + name], out["mask_" + name] = torch.from_numpy(f).unsqueeze(0), torch.from_numpy(m).unsqueeze(0)
        out[name + "_flag"] = group.shape[0] != 0
    return out


def merge_inmask_outmask(flow_inmask, flow_outmask):
    """forward_sample: where BOTH components of the in-brush flow are non-zero it wins, elsewhere the out-of-brush flow."""
    keep = (flow_inmask != 0).all(dim=2).unsqueeze(2).expand_as(flow_inmask)
    return torch.where(keep, flow_inmask, flow_outmask)


def controlnet_flow_from_drags(cmp, first_frame, drags, height, width, motion_brush_mask=None, work=384):
    """first_frame [1,3,H,W] in (0,1); drags = tracking_points_to_drags(...).  Returns controlnet_flow [1,T-1,2,H,W]."""
    from .cmp import get_flow
    n = drags["drag_in"].shape[1]
    ff = torch.nn.functional.interpolate(first_frame.float(), (work, work)).repeat(n, 1, 1, 1).unsqueeze(0)
    flows = {}
    for name in ("in", "out"):
        if drags[name + "_flag"]:
            d = drags["drag_" + name].permute(0, 1, 4, 2, 3).float()
            m = drags["mask_" + name].unsqueeze(2).repeat(1, 1, 2, 1, 1).float()
            flows[name] = get_flow(cmp, ff, d, m, height, width, motion_brush_mask if name == "in" else None)
        else:
            flows[name] = torch.zeros(1, n, 2, height, width, device=cmp.device)
    return merge_inmask_outmask(flows["in"], flows["out"])


def sample_optical_flow(A, B, h, w):
    """A [b,l,k,2] integer-valued (row, col) positions, B [b,l,k,2] values -> dense [b,l,h,w,2] + uint8 mask [b,l,h,w,2].
    Positions are clipped as the reference clips them (rows to h-1, cols to w-1); later points overwrite earlier ones."""
    b, l, k, _ = A.shape
    flow = torch.zeros((b, l, h, w, 2), dtype=B.dtype, device=B.device)
    mask = torch.zeros((b, l, h, w), dtype=torch.uint8, device=B.device)
    rows = torch.clip(A[..., 0].long(), 0, h - 1)
    cols = torch.clip(A[..., 1].long(), 0, w - 1)
    bi = torch.arange(b)[:, None, None].expand(b, l, k)
    li = torch.arange(l)[None, :, None].expand(b, l, k)
    flow[bi, li, rows, cols] = B
    mask[bi, li, rows, cols] = 1
    return flow, mask.unsqueeze(-1).repeat(1, 1, 1, 1, 2)


@torch.no_grad()
def get_sparse_flow(landmarks, h, w, t):
    """landmarks [b,t,68,2] (x,y) pixels -> forward sparse flow of frames 1..t-1 w.r.t. frame 0, sampled at frame 0's
    landmark pixels: ([b,t-1,2,h,w] flow (dx,dy), [b,t-1,2,h,w] mask)."""
    yx = torch.flip(landmarks, dims=[3])
    disp = torch.flip((yx - yx[:, 0:1])[:, 1:], dims=[3])            # back to (dx, dy)
    pos = yx[:, 0:1].repeat(1, t - 1, 1, 1)
    flow, mask = sample_optical_flow(pos, disp, h, w)
    return flow.permute(0, 1, 4, 2, 3), mask.permute(0, 1, 4, 2, 3)
