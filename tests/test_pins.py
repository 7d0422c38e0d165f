"""The numeric pins the reference offers: its own renderings of test_data/man.jpg (assets/*.png, produced by
/root/reference/src/lib.rs:18-84 with BackCamera; copies under tests/golden/).

The oracle's results on man.jpg (BackCamera detection -> face ROI -> 468-point mesh -> both eye ROIs -> eye contours) are
drawn by the restated renderer (oracle/render.py = render.rs:262-479 + the imageproc primitives it calls) and compared
with the reference's PNGs PIXEL FOR PIXEL on the annotation colour: every one of the 552 green bbox pixels, the 2414 red
mesh pixels (468 2x2 dots + 124 Bresenham segments) and the 150 red eye-contour pixels must coincide — pixel budget 0.
A landmark that moved by one pixel would move its dot, so this pins 4 + 936 + 60 truncated pixel coordinates of the
reference's own run (TFLite + OpenCV) — the tightest pin the reference's fixtures allow (they hold no float vectors).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, model_path


def _ref_mask(png, colour):
    from PIL import Image
    from oracle import render
    return render.colour_mask(np.asarray(Image.open(os.path.join(GOLDEN, png)).convert("RGBA")), colour)


@pytest.fixture(scope="module")
def man_pipeline(oracle, man_image):
    """lib.rs:24-40 through the oracle, from the decoded picture."""
    H, W = man_image.shape[:2]
    t, pad = oracle.image_to_tensor(man_image, None, (256, 256), True, (-1., 1.), False)
    rb, rs = oracle.Model(model_path("back")).run(t[None])
    dets = oracle.fd_postprocess(rb[0], rs[0], oracle.ssd_anchors(oracle.FD_BACK), 256.0, pad)
    assert len(dets) == 1
    roi = oracle.face_detection_to_roi(dets[0], (W, H))
    t2, pad2 = oracle.image_to_tensor(man_image, roi, (192, 192), False, (0., 1.), False)
    raw, flag = oracle.Model(model_path("landmark")).run(t2[None])
    assert oracle.lib().orc_face_flag_passes(float(flag.reshape(-1)[-1]))
    lms = oracle.project_landmarks(raw[0], (192, 192), (W, H), pad2, roi, False)
    left, right = oracle.iris_rois_from_face_landmarks(lms, (W, H))
    ir = oracle.Model(model_path("iris"))
    eyes = {}
    for tag, r, is_right in (("right", right, True), ("left", left, False)):
        t3, pad3 = oracle.image_to_tensor(man_image, r, (64, 64), True, (0., 1.), is_right)
        c, _i5 = ir.run(t3[None])
        eyes[tag] = oracle.project_landmarks(c[0], (64, 64), (W, H), pad3, r, is_right)
    return dict(dets=dets, landmarks=lms, eyes=eyes)


def test_bbox_rendering_matches_reference_png(man_pipeline, man_image):
    from oracle import render
    ann = render.detections_to_render_data(man_pipeline["dets"], render.GREEN, None, 4, 2)      # lib.rs:43-51
    ours = render.colour_mask(render.render_to_image(ann, man_image), render.GREEN)
    ref = _ref_mask("man_bbox.png", render.GREEN)
    assert ref.sum() == 552
    assert int((ours ^ ref).sum()) == 0
    # the four integers the rectangle encodes (render.rs:447-448): left 195, top 74, 139 x 139
    d = man_pipeline["dets"][0].astype(np.float64)
    H, W = man_image.shape[:2]
    assert (int(d[0] * W), int(d[1] * H), int(d[2] * W - d[0] * W), int(d[3] * H - d[1] * H)) == (195, 74, 139, 139)


def test_mesh_rendering_matches_reference_png(man_pipeline, man_image):
    from oracle import render
    ann = render.face_landmarks_to_render_data(man_pipeline["landmarks"], render.RED, render.RED, 2.0)   # lib.rs:61-63
    ours = render.colour_mask(render.render_to_image(ann, man_image), render.RED)
    ref = _ref_mask("man_landmark.png", render.RED)
    assert ref.sum() == 2414
    assert int((ours ^ ref).sum()) == 0, "mismatching pixels: %d" % int((ours ^ ref).sum())
    # sensitivity of the pin: moving the mesh by one pixel in x changes hundreds of pixels
    moved = man_pipeline["landmarks"].copy()
    moved[:, 0] += 1.0 / man_image.shape[1]
    other = render.colour_mask(render.render_to_image(render.face_landmarks_to_render_data(moved), man_image), render.RED)
    assert int((other ^ ref).sum()) > 1000


def test_eye_contour_rendering_matches_reference_png(man_pipeline, man_image):
    from oracle import render
    ann = render.eye_landmarks_to_render_data(man_pipeline["eyes"]["right"][:15]) + \
        render.eye_landmarks_to_render_data(man_pipeline["eyes"]["left"][:15])                  # lib.rs:66-83, eyeball_contour()
    ours = render.colour_mask(render.render_to_image(ann, man_image), render.RED)
    ref = _ref_mask("man_iris.png", render.RED)
    assert ref.sum() == 150
    assert int((ours ^ ref).sum()) == 0


def test_committed_goldens_are_the_pinned_pipeline(man_pipeline):
    """tests/golden/golden.npz (what the GPU parity tests compare against) holds exactly the values the renderings pin."""
    gold = np.load(os.path.join(GOLDEN, "golden.npz"))
    np.testing.assert_array_equal(gold["man_back_dets"], man_pipeline["dets"])
    np.testing.assert_array_equal(gold["man_face_landmarks"], man_pipeline["landmarks"])
    np.testing.assert_array_equal(gold["man_eye_right_contour"], man_pipeline["eyes"]["right"])
    np.testing.assert_array_equal(gold["man_eye_left_contour"], man_pipeline["eyes"]["left"])


def test_renderer_primitives():
    """imageproc semantics the restatement depends on: inclusive Bresenham end points, steep lines, clipping, 2x2 dots."""
    from oracle import render
    img = np.zeros((8, 8, 3), np.uint8)
    out = render.render_to_image([("lines", [(0.0, 0.0, 7 / 8, 3 / 8)], 2.0, render.RED)], img)
    m = render.colour_mask(out, render.RED)
    assert m.sum() == 8 and m[0, 0] and m[3, 7]                       # one pixel per major step, both ends drawn
    out = render.render_to_image([("lines", [(1 / 8, 0.0, 2 / 8, 7 / 8)], 2.0, render.RED)], img)
    m = render.colour_mask(out, render.RED)
    assert m.sum() == 8 and m[0, 1] and m[7, 2]                       # steep: walks y
    out = render.render_to_image([("points", [(3.5 / 8, 4.9 / 8)], 2.0, render.RED)], img)
    ys, xs = np.nonzero(render.colour_mask(out, render.RED))
    assert sorted(set(xs)) == [2, 3] and sorted(set(ys)) == [3, 4]     # truncation, then [x-1, x] x [y-1, y]
    out = render.render_to_image([("rects", [(1 / 8, 1 / 8, 9 / 8, 5 / 8)], 4.0, render.GREEN)], img)
    m = render.colour_mask(out, render.GREEN)
    assert m[1, 1:].all() and m[4, 1:].all() and m[1:5, 1].all() and not m[2, 2] and not m[5].any()   # right edge clipped away
