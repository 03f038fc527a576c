"""Closes the ``draw_landmarks`` pin on a machine that HAS OpenCV (this build image does not: `import cv2` fails, so
mofa_video_amd/landmarks.py restates cv2.line(thickness=2) / cv2.resize from the published algorithm and is pinned only to
hand-derived vectors -- DESIGN.md "PARITY UNPINNED against cv2 itself").

    python tests/golden/make_golden_cv2.py            # needs cv2 + /root/reference; writes reference_golden_cv2.npz

Runs the REFERENCE's own ``draw_landmarks`` (MOFA-Video-Keypoint/utils/utils.py:26-46, taken from the source file in place
through ast: its module imports torch-side packages at import time) and its call site's resize
(mofa_keypoint.py:310-311: ``cv2.resize(pose_img, (pw, ph), cv2.INTER_NEAREST)`` -- the flag lands in the ``dst`` position,
so the default bilinear interpolation runs) on seeded landmark sets, plus single ``cv2.line`` segments in every octant and
clipped at the borders.  tests/test_landmarks_cpu.py::test_against_cv2_fixture compares mofa_video_amd.landmarks with the
file BIT FOR BIT when it exists and is skipped otherwise."""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/MOFA-Video-Keypoint/utils/utils.py"


def main():
    import cv2                                                   # fails here by design on a box without OpenCV
    tree = ast.parse(open(REF).read())
    ns = {"np": np, "cv2": cv2}
    for node in tree.body:                                       # PARTS (module-level assignment) + draw_landmarks
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "PARTS" for t in node.targets):
            exec(compile(ast.Module([node], []), REF, "exec"), ns)
        if isinstance(node, ast.FunctionDef) and node.name == "draw_landmarks":
            exec(compile(ast.Module([node], []), REF, "exec"), ns)
    rng = np.random.RandomState(11)
    out = {}
    # (a) single segments: every octant, degenerate, clipped on each border, long and short
    segs = [((2, 2), (6, 2)), ((3, 1), (3, 8)), ((5, 5), (5, 5)), ((1, 1), (9, 6)), ((9, 1), (1, 6)), ((2, 9), (7, 1)),
            ((-4, 3), (5, 8)), ((10, 12), (20, 3)), ((0, 0), (15, 15)), ((14, 2), (30, 9))]
    segs += [tuple(map(tuple, rng.randint(-5, 37, size=(2, 2)))) for _ in range(40)]
    imgs = []
    for p1, p2 in segs:
        img = np.zeros((32, 32, 3))
        cv2.line(img, (int(p1[0]), int(p1[1])), (int(p2[0]), int(p2[1])), (7, 8, 9), thickness=2)
        imgs.append(img)
    out["seg_pts"] = np.array(segs, dtype=np.int64)
    out["seg_imgs"] = np.array(imgs)
    # (b) the reference's draw_landmarks at its 320 x 320 drawing size + the call site's resize to 576 x 1024 and 256 x 256
    th = np.linspace(0, 2 * np.pi, 69)[:68]
    sets = []
    for k in range(4):
        base = np.stack([160 + 70 * np.cos(th), 160 + 110 * np.sin(th)], 1) + rng.uniform(-12, 12, size=(68, 2))
        sets.append(base)
    sets = np.array(sets)
    out["ldmk"] = sets
    out["drawn"] = np.array([ns["draw_landmarks"](s, 320, 320) for s in sets])
    out["resized_576x1024"] = np.array([cv2.resize(d, (1024, 576), cv2.INTER_NEAREST) for d in out["drawn"]])
    out["resized_256x256"] = np.array([cv2.resize(d, (256, 256), cv2.INTER_NEAREST) for d in out["drawn"]])
    path = os.path.join(HERE, "reference_golden_cv2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()}, "cv2", cv2.__version__)


if __name__ == "__main__":
    main()
