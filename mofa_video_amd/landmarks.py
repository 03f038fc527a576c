"""Landmark pose images: ``draw_landmarks`` of MOFA-Video-Keypoint/utils/utils.py:7-46 -- the 15 polyline
``PARTS`` of a 68-point face drawn with ``cv2.line(image, p1, p2, color, thickness=2)`` on a float64 canvas -- and the
call site's resize (MOFA-Video-Keypoint/mofa_keypoint.py:304-314), host-side like the reference.

OpenCV is a third-party dependency that is absent here (requirements: ``opencv-python``, unpinned), so its rasteriser is
RESTATED from the published algorithm of OpenCV 4.x ``modules/imgproc/src/drawing.cpp``: ``ThickLine`` for thickness > 1 =
``FillConvexPoly`` of the quad p0 +- d, p1 +- d in 16.16 fixed point (d = the unit normal times thickness / 2, components
rounded half-to-even), its four edges drawn with the fixed-point ``Line2``, then a filled ``Circle`` of radius
(thickness + 1) / 2 at both end points.  All integer arithmetic, reproduced operation by operation.  PARITY UNPINNED
against cv2 itself (it cannot be imported here): tests/test_landmarks_cpu.py pins the restatement with hand-derived
vectors (axis-aligned, diagonal and clipped lines, the radius-1 circle) and structural properties.
"""
import numpy as np
import torch

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT

PARTS = [
    ('FACE', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17], (10, 200, 10)),
    ('LEFT_EYE', [43, 44, 45, 46, 47, 48, 43], (180, 200, 10)),
    ('LEFT_EYEBROW', [23, 24, 25, 26, 27], (180, 220, 10)),
    ('RIGHT_EYE', [37, 38, 39, 40, 41, 42, 37], (10, 200, 180)),
    ('RIGHT_EYEBROW', [18, 19, 20, 21, 22], (10, 220, 180)),
    ('NOSE_UP', [28, 29, 30, 31], (10, 200, 250)),
    ('NOSE_DOWN', [32, 33, 34, 35, 36], (250, 200, 10)),
    ('LIPS_OUTER_BOTTOM_LEFT', [55, 56, 57, 58], (10, 180, 20)),
    ('LIPS_OUTER_BOTTOM_RIGHT', [49, 60, 59, 58], (20, 10, 180)),
    ('LIPS_INNER_BOTTOM_LEFT', [65, 66, 67], (100, 100, 30)),
    ('LIPS_INNER_BOTTOM_RIGHT', [61, 68, 67], (100, 150, 50)),
    ('LIPS_OUTER_TOP_LEFT', [52, 53, 54, 55], (20, 80, 100)),
    ('LIPS_OUTER_TOP_RIGHT', [52, 51, 50, 49], (80, 100, 20)),
    ('LIPS_INNER_TOP_LEFT', [63, 64, 65], (120, 100, 200)),
    ('LIPS_INNER_TOP_RIGHT', [63, 62, 61], (150, 120, 100)),
]


def _cv_round(v):
    """cvRound: round half to even (lrint in the default rounding mode)"""
    return int(np.rint(v))


def _tdiv(a, b):
    """C integer division (truncation toward zero)"""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _hline(img, y, x1, x2, color):
    if x2 >= x1:
        img[y, x1:x2 + 1] = color


def _clip_line(w_fx, h_fx, p1, p2):
    """cv::clipLine on fixed-point coordinates (Cohen-Sutherland, integer; the double-precision products truncate)"""
    right, bottom = w_fx - 1, h_fx - 1
    x1, y1, x2, y2 = p1[0], p1[1], p2[0], p2[1]
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def _line2(img, p1, p2, color):
    """cv::Line2: a line between 16.16 fixed-point end points, one pixel per step along the major axis"""
    h, w = img.shape[:2]
    ok, p1, p2 = _clip_line(w << XY_SHIFT, h << XY_SHIFT, p1, p2)
    if not ok:
        return
    x1, y1, x2, y2 = p1[0], p1[1], p2[0], p2[1]
    dx, dy = x2 - x1, y2 - y1
    ax, ay = abs(dx), abs(dy)

    def put(x, y):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = color
    if ax > ay:
        if dx < 0:                                   # walk left to right
            dy = -dy
            x1, x2, y1, y2 = x2, x1, y2, y1
        x_step, y_step = XY_ONE, _tdiv(dy << XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> XY_SHIFT
    else:
        if dy < 0:                                   # walk top to bottom
            dx = -dx
            x1, x2, y1, y2 = x2, x1, y2, y1
        x_step, y_step = _tdiv(dx << XY_SHIFT, ay | 1), XY_ONE
        ecount = (y2 - y1) >> XY_SHIFT
    x1 += XY_ONE >> 1
    y1 += XY_ONE >> 1
    put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)
    if ax > ay:
        x1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1, y1 >> XY_SHIFT)
            x1 += 1
            y1 += y_step
            ecount -= 1
    else:
        y1 >>= XY_SHIFT
        while ecount >= 0:
            put(x1 >> XY_SHIFT, y1)
            x1 += x_step
            y1 += 1
            ecount -= 1


def _fill_convex_poly(img, v, color):
    """cv::FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT): outline by Line2, then one horizontal span per
    scanline between the two active edges (edge x advances by a rounded per-row increment)"""
    h, w = img.shape[:2]
    npts, shift = len(v), XY_SHIFT
    delta = 1 << shift >> 1
    delta1 = delta2 = XY_ONE >> 1
    xmin = xmax = v[0][0]
    ymin = ymax = v[0][1]
    imin = 0
    p0 = v[npts - 1]
    for i, p in enumerate(v):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax, xmax, xmin = max(ymax, p[1]), max(xmax, p[0]), min(xmin, p[0])
        _line2(img, p0, p, color)
        p0 = p
    xmin, xmax = (xmin + delta) >> shift, (xmax + delta) >> shift
    ymin, ymax = (ymin + delta) >> shift, (ymax + delta) >> shift
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= w or ymin >= h:
        return
    ymax = min(ymax, h - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=npts - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = npts
    y = ymin
    while True:
        for e in edge:
            if y >= e["ye"]:
                idx0, di = e["idx"], e["di"]
                idx = idx0 + di
                if idx >= npts:
                    idx -= npts
                while True:
                    edges -= 1
                    if edges < 0:                    # `for (; edges-- > 0; )` ran out
                        break
                    ty = (v[idx][1] + delta) >> shift
                    if ty > y:
                        xs, xe = v[idx0][0], v[idx][0]
                        e["ye"] = ty
                        e["dx"] = _tdiv((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e["x"] = xs
                        e["idx"] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= npts:
                        idx -= npts
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = (edge[left]["x"] + delta1) >> XY_SHIFT
            xx2 = (edge[right]["x"] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < w:
                _hline(img, y, max(xx1, 0), min(xx2, w - 1), color)
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break


def _circle_filled(img, cx, cy, radius, color):
    """cv::Circle(img, center, radius, color, fill = 1): midpoint circle, horizontal spans"""
    h, w = img.shape[:2]
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        y11, y12, y21, y22 = cy - dy, cy + dy, cy - dx, cy + dx
        x11, x12, x21, x22 = cx - dx, cx + dx, cx - dy, cx + dy
        if x11 < w and x12 >= 0 and y21 < h and y22 >= 0:
            a, b = max(x11, 0), min(x12, w - 1)
            if 0 <= y11 < h:
                _hline(img, y11, a, b, color)
            if 0 <= y12 < h:
                _hline(img, y12, a, b, color)
            if x21 < w and x22 >= 0:
                a, b = max(x21, 0), min(x22, w - 1)
                if 0 <= y21 < h:
                    _hline(img, y21, a, b, color)
                if 0 <= y22 < h:
                    _hline(img, y22, a, b, color)
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0                  # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2


def line(img, pt1, pt2, color, thickness=2):
    """``cv2.line(img, pt1, pt2, color, thickness)`` for thickness >= 2, LINE_8, shift 0 (cv::ThickLine), in place"""
    if thickness < 2:
        raise NotImplementedError("only the thick-line path the reference uses (thickness = 2) is restated")
    p0 = (int(pt1[0]) << XY_SHIFT, int(pt1[1]) << XY_SHIFT)
    p1 = (int(pt2[0]) << XY_SHIFT, int(pt2[1]) << XY_SHIFT)
    inv = 1.0 / XY_ONE
    dx, dy = (p0[0] - p1[0]) * inv, (p1[1] - p0[1]) * inv
    r = dx * dx + dy * dy
    odd = thickness & 1
    th = thickness << (XY_SHIFT - 1)
    if abs(r) > np.finfo(np.float64).eps:
        r = (th + odd * XY_ONE * 0.5) / np.sqrt(r)
        dpx, dpy = _cv_round(dy * r), _cv_round(dx * r)
        quad = [(p0[0] + dpx, p0[1] + dpy), (p0[0] - dpx, p0[1] - dpy), (p1[0] - dpx, p1[1] - dpy), (p1[0] + dpx, p1[1] + dpy)]
        _fill_convex_poly(img, quad, color)
    for p in (p0, p1):
        _circle_filled(img, (p[0] + (XY_ONE >> 1)) >> XY_SHIFT, (p[1] + (XY_ONE >> 1)) >> XY_SHIFT,
                       (th + (XY_ONE >> 1)) >> XY_SHIFT, color)
    return img


def draw_landmarks(keypoints, h, w):
    """MOFA-Video-Keypoint/utils/utils.py:26-46: keypoints [68, 2] (x, y) -> float64 image [h, w, 3], colours 0..250"""
    image = np.zeros((h, w, 3))
    kp = np.asarray(keypoints)
    for _name, indices, color in PARTS:
        pts = kp[np.array(indices) - 1]
        for i in range(len(indices) - 1):
            line(image, (int(pts[i][0]), int(pts[i][1])), (int(pts[i + 1][0]), int(pts[i + 1][1])), color, thickness=2)
    return image


def resize_linear(img, width, height):
    """``cv2.resize(img, (width, height))`` with the default INTER_LINEAR on a float64 image -- what the call site's
    ``cv2.resize(pose_img, (pw, ph), cv2.INTER_NEAREST)`` executes (the third positional argument is ``dst``, so the
    interpolation stays bilinear; mofa_keypoint.py:312).  Half-pixel centres, edge replication, float32 weights applied in
    double precision, rows then columns (OpenCV resizeGeneric_ / HResizeLinear / VResizeLinear for CV_64F)."""
    sh, sw = img.shape[:2]

    def taps(dsize, ssize):
        scale = ssize / dsize
        d = np.arange(dsize, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= ssize - 1
        f[hi], s[hi] = 0.0, ssize - 1
        return s, np.minimum(s + 1, ssize - 1), (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)
    x0, x1, ax0, ax1 = taps(width, sw)
    y0, y1, by0, by1 = taps(height, sh)
    rows = img[:, x0] * ax0[None, :, None] + img[:, x1] * ax1[None, :, None]             # horizontal pass, all source rows
    return rows[y0] * by0[:, None, None] + rows[y1] * by1[:, None, None]


def pose_images(landmarks, height, width, draw_size=320):
    """MOFA-Video-Keypoint/mofa_keypoint.py:299-316: landmarks [N, 68, 2] in pixel coordinates of the height x width
    frame -> pose images fp32 [1, N, 3, height, width] in [0, 1] (drawn at 320 x 320 "because training uses 320 x 320",
    then resized), the ``landmarks`` argument of the Keypoint / Hybrid pipelines"""
    lm = np.array(landmarks, dtype=np.float64).copy()
    lm[:, :, 0] = lm[:, :, 0] / width * draw_size
    lm[:, :, 1] = lm[:, :, 1] / height * draw_size
    imgs = np.stack([resize_linear(draw_landmarks(lm[i], draw_size, draw_size), width, height) for i in range(lm.shape[0])])
    return (torch.from_numpy(imgs).permute(0, 3, 1, 2).float() / 255.0).unsqueeze(0)
