"""Generates the committed golden vectors under tests/golden/ from the C oracle (oracle/c), after cross-checking the
oracle's network outputs against the independent torch-CPU evaluation (oracle/np/evaluate.py) — SURVEY.md §8c.

The reference itself (Rust + TensorFlow-Lite + OpenCV) cannot be built or imported in this image, so these vectors
pin the *oracle*, not the reference: PARITY UNPINNED apart from the rendered-PNG pins checked in tests/test_pins.py.

Run from the repo root:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from oracle.np import evaluate, tfl3  # noqa: E402
from tests.conftest import MODEL_FILES, model_path, seeded_input  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def crosscheck(name, x, outs):
    g = tfl3.load(model_path(name))
    ref = evaluate.run(g, x)
    for a, b in zip(ref, outs):
        err = np.abs(a.reshape(b.shape) - b) / np.maximum(1.0, np.abs(b))
        assert err.max() <= 1e-4, (name, float(err.max()))
    return float(max(np.abs(a.reshape(b.shape) - b).max() for a, b in zip(ref, outs)))


def main():
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(OUT, "man.jpg")).convert("RGB"))
    H, W = img.shape[:2]
    gold = {}

    # ---- seeded synthetic frames through every network (2 frames each)
    for name in MODEL_FILES:
        m = po.Model(model_path(name))
        x = seeded_input(name, 2, 1234, m.input_dims[1:3])
        outs = m.run(x, nthreads=2)
        d = crosscheck(name, x, outs)
        print("%-9s torch-vs-C max|diff| = %.3e" % (name, d))
        for k, o in enumerate(outs):
            gold["noise_%s_out%d" % (name, k)] = o.reshape(2, -1)

    # ---- man.jpg through the whole pipeline (BackCamera -> mesh -> both irises), every stage recorded
    fd = po.Model(model_path("back"))
    t, pad = po.image_to_tensor(img, None, (256, 256), True, (-1., 1.), False)
    u8 = np.round((t + 1.0) * 255.0 / 2.0).astype(np.uint8)
    assert np.array_equal((u8.astype(np.float64) * 2.0 / 255.0 - 1.0).astype(np.float32), t)
    rb, rs = fd.run(t[None])
    crosscheck("back", t[None], [rb, rs])
    anchors = po.ssd_anchors(po.FD_BACK)
    dets = po.fd_postprocess(rb[0], rs[0], anchors, 256.0, pad)
    gold.update(man_back_u8=u8, man_back_pad=np.asarray(pad), man_back_boxes=rb[0], man_back_scores=rs[0].reshape(-1),
                man_back_dets=dets)
    roi = po.face_detection_to_roi(dets[0], (W, H))
    gold["man_face_roi"] = np.array([roi.x_center, roi.y_center, roi.width, roi.height, roi.rotation, roi.normalized])
    fl = po.Model(model_path("landmark"))
    t2, pad2 = po.image_to_tensor(img, roi, (192, 192), False, (0., 1.), False)
    u8b = np.round(t2 * 255.0).astype(np.uint8)
    assert np.array_equal((u8b.astype(np.float64) * 1.0 / 255.0 + 0.0).astype(np.float32), t2)
    lm_raw, flag = fl.run(t2[None])
    crosscheck("landmark", t2[None], [lm_raw, flag])
    lms = po.project_landmarks(lm_raw[0], (192, 192), (W, H), pad2, roi, False)
    gold.update(man_face_u8=u8b, man_face_raw=lm_raw.reshape(-1), man_face_flag=flag.reshape(-1), man_face_landmarks=lms)
    left, right = po.iris_rois_from_face_landmarks(lms, (W, H))
    ir = po.Model(model_path("iris"))
    for tag, r, is_right in (("right", right, True), ("left", left, False)):
        t3, pad3 = po.image_to_tensor(img, r, (64, 64), True, (0., 1.), is_right)
        u8c = np.round(t3 * 255.0).astype(np.uint8)
        assert np.array_equal((u8c.astype(np.float64) / 255.0).astype(np.float32), t3)
        c, i5 = ir.run(t3[None])
        crosscheck("iris", t3[None], [c, i5])
        gold["man_eye_%s_roi" % tag] = np.array([r.x_center, r.y_center, r.width, r.height, r.rotation, r.normalized])
        gold["man_eye_%s_u8" % tag] = u8c
        gold["man_eye_%s_pad" % tag] = np.asarray(pad3)
        gold["man_eye_%s_raw_contour" % tag] = c.reshape(-1)
        gold["man_eye_%s_raw_iris" % tag] = i5.reshape(-1)
        gold["man_eye_%s_contour" % tag] = po.project_landmarks(c[0], (64, 64), (W, H), pad3, r, is_right)
        gold["man_eye_%s_iris" % tag] = po.project_landmarks(i5[0], (64, 64), (W, H), pad3, r, is_right)

    # ---- anchors
    for kind, tag in ((po.FD_FRONT, "front"), (po.FD_BACK, "back"), (po.FD_FULL, "full")):
        gold["anchors_" + tag] = po.ssd_anchors(kind)

    # ---- NMS stress vectors: many overlapping candidates, ties, degenerate boxes
    rsn = np.random.RandomState(7)
    n = 896
    rb = rsn.normal(0, 6, (4, n, 16)).astype(np.float32)
    rb[:, :, 2:4] = np.abs(rb[:, :, 2:4]) * 8 + 20          # plausible sizes
    sc = rsn.normal(-2, 3, (4, n)).astype(np.float32)
    sc[1, ::7] = 3.0                                          # exact score ties
    rb[2, ::5, 2] = -5.0                                      # invalid (negative width) boxes
    sc[3] = 100.0                                             # clamp + everything passes
    pads = np.array([[0, 0, 0, 0], [0.1, 0, 0.1, 0], [0, 0.125, 0, 0.125], [0, 0, 0, 0]], np.float64)
    nd = []
    for f in range(4):
        d = po.fd_postprocess(rb[f], sc[f], gold["anchors_back"], 256.0, pads[f])
        nd.append(d)
        gold["nms_dets_%d" % f] = d
    gold.update(nms_raw_boxes=rb, nms_raw_scores=sc, nms_pads=pads)
    print("nms stress counts:", [len(d) for d in nd])

    np.savez_compressed(os.path.join(OUT, "golden.npz"), **gold)
    print("wrote", os.path.join(OUT, "golden.npz"), os.path.getsize(os.path.join(OUT, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
