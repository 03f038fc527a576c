"""TEST INFRASTRUCTURE (see oracle/__init__.py).  CPU restatement of the output stage.

* ``postprocess`` / ``tensor2vid``: Traj/pipeline/pipeline.py:57-69 calls diffusers 0.24.0
  ``VaeImageProcessor.postprocess`` (absent from /root/reference: PARITY UNPINNED for this function beyond its
  published definition) = denormalize ``(x / 2 + 0.5).clamp(0, 1)``; "np": ``.cpu().permute(0,2,3,1).float().numpy()``;
  "pil": ``(images * 255).round().astype("uint8")`` -> PIL.
* ``flow_to_image``: Traj/utils/flow_viz.py:146-277 (make_color_wheel, compute_color, flow_to_image), pinned by
  tests/golden/reference_golden_output.pt (the reference's own function run on seeded flows,
  tests/golden/make_golden_output.py)."""
import numpy as np
import torch

UNKNOWN_FLOW_THRESH = 1e7


def postprocess(image: torch.Tensor, output_type: str = "pil"):
    image = (image / 2 + 0.5).clamp(0, 1)                      # denormalize
    if output_type == "pt":
        return image
    image = image.cpu().permute(0, 2, 3, 1).float().numpy()    # pt_to_numpy
    if output_type == "np":
        return image
    from PIL import Image
    return [Image.fromarray(im) for im in (image * 255).round().astype("uint8")]   # numpy_to_pil


def tensor2vid(video: torch.Tensor, output_type="np"):         # pipeline.py:57-69
    return [postprocess(video[b].permute(1, 0, 2, 3), output_type) for b in range(video.shape[0])]


def make_color_wheel():                                        # flow_viz.py:146-193
    RY, YG, GC, CB, BM, MR = 15, 6, 4, 11, 13, 6
    cw = np.zeros([RY + YG + GC + CB + BM + MR, 3])
    col = 0
    cw[0:RY, 0] = 255
    cw[0:RY, 1] = np.floor(255 * np.arange(0, RY) / RY)
    col += RY
    cw[col:col + YG, 0] = 255 - np.floor(255 * np.arange(0, YG) / YG)
    cw[col:col + YG, 1] = 255
    col += YG
    cw[col:col + GC, 1] = 255
    cw[col:col + GC, 2] = np.floor(255 * np.arange(0, GC) / GC)
    col += GC
    cw[col:col + CB, 1] = 255 - np.floor(255 * np.arange(0, CB) / CB)
    cw[col:col + CB, 2] = 255
    col += CB
    cw[col:col + BM, 2] = 255
    cw[col:col + BM, 0] = np.floor(255 * np.arange(0, BM) / BM)
    col += BM
    cw[col:col + MR, 2] = 255 - np.floor(255 * np.arange(0, MR) / MR)
    cw[col:col + MR, 0] = 255
    return cw


def compute_color(u, v):                                       # flow_viz.py:196-238
    h, w = u.shape
    img = np.zeros([h, w, 3])
    nan = np.isnan(u) | np.isnan(v)
    u[nan] = 0
    v[nan] = 0
    cw = make_color_wheel()
    ncols = cw.shape[0]
    rad = np.sqrt(u ** 2 + v ** 2)
    a = np.arctan2(-v, -u) / np.pi
    fk = (a + 1) / 2 * (ncols - 1) + 1
    k0 = np.floor(fk).astype(int)
    k1 = k0 + 1
    k1[k1 == ncols + 1] = 1
    f = fk - k0
    for i in range(3):
        tmp = cw[:, i]
        col0, col1 = tmp[k0 - 1] / 255, tmp[k1 - 1] / 255
        col = (1 - f) * col0 + f * col1
        idx = rad <= 1
        col[idx] = 1 - rad[idx] * (1 - col[idx])
        col[~idx] *= 0.75
        img[:, :, i] = np.uint8(np.floor(255 * col * (1 - nan)))
    return img


def flow_to_image(flow: torch.Tensor):                         # flow_viz.py:241-277; flow [H,W,2] torch tensor
    flow = flow.clone()
    u, v = flow[:, :, 0], flow[:, :, 1]
    unknown = (abs(u) > UNKNOWN_FLOW_THRESH) | (abs(v) > UNKNOWN_FLOW_THRESH)
    u[unknown] = 0
    v[unknown] = 0
    rad = torch.sqrt(u ** 2 + v ** 2)
    maxrad = max(-1, torch.max(rad).cpu().numpy())
    u = u / (maxrad + np.finfo(float).eps)
    v = v / (maxrad + np.finfo(float).eps)
    img = compute_color(u.cpu().numpy(), v.cpu().numpy())
    img[np.repeat(unknown[:, :, None].cpu().numpy(), 3, axis=2)] = 0
    return np.uint8(img)
