"""The only numeric pins the reference offers: its own renderings of test_data/man.jpg (assets/*.png, produced by
/root/reference/src/lib.rs:18-84 with BackCamera).  The oracle must reproduce them to +-2 px.
The PNG-derived numbers below were recovered from the pure-colour pixels of the committed PNG copies."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, model_path


def _colour_extent(png, rgb):
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(GOLDEN, png)).convert("RGB")).astype(int)
    m = (a[..., 0] == rgb[0]) & (a[..., 1] == rgb[1]) & (a[..., 2] == rgb[2])
    ys, xs = np.nonzero(m)
    return xs.min(), xs.max(), ys.min(), ys.max()


def test_bbox_pin(oracle, man_image):
    x0, x1, y0, y1 = _colour_extent("man_bbox.png", (0, 255, 0))
    H, W = man_image.shape[:2]
    t, pad = oracle.image_to_tensor(man_image, None, (256, 256), True, (-1., 1.), False)
    rb, rs = oracle.Model(model_path("back")).run(t[None])
    d = oracle.fd_postprocess(rb[0], rs[0], oracle.ssd_anchors(oracle.FD_BACK), 256.0, pad)
    assert len(d) == 1
    left, top = d[0, 0] * W, d[0, 1] * H
    w, h = (d[0, 2] - d[0, 0]) * W, (d[0, 3] - d[0, 1]) * H
    # renderer truncates (render.rs:447-448): rectangle at (left, top) with (w, h)
    assert abs(int(left) - x0) <= 1 and abs(int(top) - y0) <= 1
    assert abs(int(w) - (x1 - x0)) <= 2 and abs(int(h) - (y1 - y0)) <= 2


def test_landmark_and_iris_pins(oracle, man_image):
    gold = np.load(os.path.join(GOLDEN, "golden.npz"))
    H, W = man_image.shape[:2]
    lx0, lx1, ly0, ly1 = _colour_extent("man_landmark.png", (255, 0, 0))
    lm = gold["man_face_landmarks"]
    assert abs(lm[:, 0].min() * W - lx0) <= 2.5 and abs(lm[:, 0].max() * W - lx1) <= 2.5
    assert abs(lm[:, 1].min() * H - ly0) <= 2.5 and abs(lm[:, 1].max() * H - ly1) <= 2.5
    ix0, ix1, iy0, iy1 = _colour_extent("man_iris.png", (255, 0, 0))
    p = np.concatenate([gold["man_eye_right_contour"][:15], gold["man_eye_left_contour"][:15]])
    assert abs(p[:, 0].min() * W - ix0) <= 2.5 and abs(p[:, 0].max() * W - ix1) <= 2.5
    assert abs(p[:, 1].min() * H - iy0) <= 2.5 and abs(p[:, 1].max() * H - iy1) <= 2.5
