"""CPU tests of the oracle (test infrastructure): golden regression, independent cross-check, glue properties.
PARITY UNPINNED — see oracle/c/oracle.h."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, MODEL_FILES, model_path, seeded_input


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


@pytest.mark.parametrize("name", list(MODEL_FILES))
def test_oracle_matches_golden_noise(oracle, gold, name):
    m = oracle.Model(model_path(name))
    x = seeded_input(name, 2, 1234, m.input_dims[1:3])
    outs = m.run(x, nthreads=2)
    for k, o in enumerate(outs):
        g = gold["noise_%s_out%d" % (name, k)]
        np.testing.assert_allclose(o.reshape(2, -1), g, rtol=0, atol=1e-4 * max(1.0, float(np.abs(g).max())))


@pytest.mark.parametrize("name", ["front", "landmark", "iris", "full", "sparse"])
def test_oracle_vs_independent_torch_eval(oracle, name):
    torch = pytest.importorskip("torch")
    from oracle.np import evaluate, tfl3
    m = oracle.Model(model_path(name))
    x = seeded_input(name, 1, 99, m.input_dims[1:3])
    ref = evaluate.run(tfl3.load(model_path(name)), x)
    for a, b in zip(ref, m.run(x)):
        err = np.abs(a.reshape(b.shape) - b) / np.maximum(1.0, np.abs(b))
        assert err.max() <= 1e-4


def test_anchor_counts_and_layout(oracle, gold):
    a = oracle.ssd_anchors(oracle.FD_BACK)
    assert a.shape == (896, 2)
    np.testing.assert_array_equal(a, gold["anchors_back"])
    # 16x16 cells x 2 then 8x8 cells x 6, y-major, centres (x+0.5)/fm
    assert a[0, 0] == np.float32(0.5) / np.float32(16) and a[0, 1] == np.float32(0.5) / np.float32(16)
    assert np.array_equal(a[0], a[1]) and a[2, 0] == np.float32(1.5) / np.float32(16)
    assert a[512, 0] == np.float32(0.5) / np.float32(8)
    assert oracle.ssd_anchors(oracle.FD_FRONT).shape == (896, 2)
    assert oracle.ssd_anchors(oracle.FD_SHORT).shape == (896, 2)
    f = oracle.ssd_anchors(oracle.FD_FULL)
    assert f.shape == (2304, 2) and np.array_equal(f, oracle.ssd_anchors(oracle.FD_FULL_SPARSE))


def test_man_pipeline_matches_golden(oracle, gold, man_image):
    H, W = man_image.shape[:2]
    t, pad = oracle.image_to_tensor(man_image, None, (256, 256), True, (-1., 1.), False)
    u8 = np.round((t + 1.0) * 255.0 / 2.0).astype(np.uint8)
    np.testing.assert_array_equal(u8, gold["man_back_u8"])
    np.testing.assert_allclose(pad, gold["man_back_pad"])
    fd = oracle.Model(model_path("back"))
    rb, rs = fd.run(t[None])
    dets = oracle.fd_postprocess(rb[0], rs[0], oracle.ssd_anchors(oracle.FD_BACK), 256.0, pad)
    np.testing.assert_allclose(dets, gold["man_back_dets"], atol=1e-5)


def test_nms_golden_and_properties(oracle, gold):
    anchors = gold["anchors_back"]
    for f in range(4):
        d = oracle.fd_postprocess(gold["nms_raw_boxes"][f], gold["nms_raw_scores"][f], anchors, 256.0, gold["nms_pads"][f])
        np.testing.assert_array_equal(d, gold["nms_dets_%d" % f])
        # outputs come out in non-increasing head-score order and every score passed the 0.5 threshold
        assert np.all(np.diff(d[:, 16]) <= 0) and np.all(d[:, 16] > 0.5)


def test_nms_empty_and_single(oracle):
    anchors = oracle.ssd_anchors(oracle.FD_BACK)
    rb = np.zeros((896, 16), np.float32)
    assert len(oracle.fd_postprocess(rb, np.full(896, -10.0, np.float32), anchors, 256.0)) == 0
    rb[:, 2:4] = 30.0
    sc = np.full(896, -10.0, np.float32)
    sc[100] = 4.0
    d = oracle.fd_postprocess(rb, sc, anchors, 256.0)
    assert len(d) == 1
    ax, ay = anchors[100]
    np.testing.assert_allclose(d[0, :4], [ax - 30 / 512, ay - 30 / 512, ax + 30 / 512, ay + 30 / 512], atol=1e-7)
    # sigmoid(x) > 0.5 is evaluated on the f32 sigmoid: a tiny positive logit rounds to exactly 0.5 and is dropped
    sc[100] = 1e-9
    assert len(oracle.fd_postprocess(rb, sc, anchors, 256.0)) == 0


def test_letterbox_guard(oracle):
    anchors = oracle.ssd_anchors(oracle.FD_BACK)
    rb = np.zeros((896, 16), np.float32)
    rb[:, 2:4] = 30.0
    sc = np.full(896, 5.0, np.float32)
    with pytest.raises(RuntimeError):
        oracle.fd_postprocess(rb, sc, anchors, 256.0, (0.5, 0.0, 0.5, 0.0))


def test_project_landmarks_identity_and_flip(oracle):
    raw = np.array([96, 48, 19.2, 0, 192, -9.6], np.float32)
    out = oracle.project_landmarks(raw, (192, 192), (640, 480))
    np.testing.assert_allclose(out, [[0.5, 0.25, 0.1], [0.0, 1.0, -0.05]], atol=1e-7)
    out = oracle.project_landmarks(raw, (192, 192), (640, 480), flip=True)
    np.testing.assert_allclose(out[:, 0], [0.5, 1.0], atol=1e-7)
    roi = oracle.Rect(0.5, 0.5, 0.5, 0.25, np.pi / 2, 1)
    out = oracle.project_landmarks(np.array([192, 96, 0], np.float32), (192, 192), (640, 480), roi=roi)
    # (1.0,0.5) -> centred (0.5,0) -> rotated by +90deg (0,0.5) -> scaled (0,0.125) -> + centre
    np.testing.assert_allclose(out[0], [0.5, 0.625, 0.0], atol=1e-6)


def test_roi_helpers(oracle, gold):
    roi = oracle.face_detection_to_roi(gold["man_back_dets"][0], (540, 360))
    np.testing.assert_allclose([roi.x_center, roi.y_center, roi.width, roi.height, roi.rotation, roi.normalized],
                               gold["man_face_roi"], rtol=1e-12)
    # SquareLong: the ROI is square in pixels
    assert abs(roi.width * 540 - roi.height * 360) < 1e-9
    l, r = oracle.iris_rois_from_face_landmarks(gold["man_face_landmarks"], (540, 360))
    np.testing.assert_allclose([l.x_center, l.y_center, l.width, l.height, l.rotation], gold["man_eye_left_roi"][:5], rtol=1e-12)
    np.testing.assert_allclose([r.x_center, r.y_center, r.width, r.height, r.rotation], gold["man_eye_right_roi"][:5], rtol=1e-12)


def test_update_face_landmarks_with_iris_results(mi, oracle):
    """iris_landmark.rs:380-398 through the C ABI (host-only entry point: runs without a GPU) against the oracle's
    restatement, plus the properties the index maps (iris_landmark.rs:64-95) guarantee."""
    rs = np.random.RandomState(3)
    face, left, right = rs.rand(468, 3), rs.rand(71, 3) + 2.0, rs.rand(71, 3) + 4.0
    lm = lambda a: [mi.Landmark(*map(float, p)) for p in a]
    got = mi.update_face_landmarks_with_iris_results(lm(face), mi.IrisResults(lm(left), lm(np.zeros((5, 3)))), mi.IrisResults(lm(right), lm(np.zeros((5, 3)))))
    got = np.array([[p.x, p.y, p.z] for p in got])
    want = oracle.update_face_landmarks_with_iris_results(face, left, right)
    np.testing.assert_array_equal(got, want)
    changed = np.where((got != face).any(axis=1))[0]
    assert len(changed) == 142                                   # 71 + 71 distinct mesh points, nothing else touched
    assert set(changed) == set(oracle.EYE_TO_FACE[0]) | set(oracle.EYE_TO_FACE[1])
    assert (got[list(oracle.EYE_TO_FACE[0])] >= 2.0).all() and (got[list(oracle.EYE_TO_FACE[0])] < 3.0).all()
    assert (got[33] == left[0]).all() and (got[263] == right[0]).all()   # eye corners: first entries of the maps
    with pytest.raises(mi.MiError):                              # the reference returns Err for a wrong landmark count
        mi.update_face_landmarks_with_iris_results(lm(face[:100]), mi.IrisResults(lm(left), []), mi.IrisResults(lm(right), []))


def test_bbox_helpers_through_the_c_abi(mi, oracle):
    """transform.rs:44-109 (bbox_to_roi / select_roi_size, all three SizeModes, rotation wrap) and 146-165 (bbox_from_landmarks)
    through the C ABI (host-only: runs without a GPU) against the oracle's restatement, dtype for dtype (f64): bit-exact."""
    import ctypes as C
    rs = np.random.RandomState(7)
    L = oracle.lib()
    for _ in range(200):
        x0, y0 = rs.rand(2) * 0.6
        bbox = (x0, y0, x0 + 0.05 + rs.rand() * 0.3, y0 + 0.05 + rs.rand() * 0.3)
        size = (int(rs.randint(64, 2000)), int(rs.randint(64, 2000)))
        kp = [(rs.rand() * size[0], rs.rand() * size[1]), (rs.rand() * size[0], rs.rand() * size[1])] if rs.rand() < 0.8 else None
        scale = (1.0 + rs.rand() * 2, 1.0 + rs.rand() * 2)
        mode = int(rs.randint(0, 3))
        got = mi.bbox_to_roi(bbox, size, kp, scale, mode)
        want = oracle.Rect()
        b = (C.c_double * 4)(*bbox)
        k = (C.c_double * 4)(*[v for p in kp for v in p]) if kp else None
        assert L.orc_bbox_to_roi(b, size[0], size[1], k, scale[0], scale[1], mode, C.byref(want)) == 0
        for f in ("x_center", "y_center", "width", "height", "rotation"):
            assert getattr(got, f) == getattr(want, f), f
        assert -np.pi <= got.rotation < np.pi and got.normalized
    with pytest.raises(mi.MiError):
        mi.bbox_to_roi((0.1, 0.1, 2.5, 0.5), (100, 100))        # BBox::normalized is xmin >= -1 && xmax < 2 && ymin >= -1 (types.rs:133-135)
    assert mi.bbox_to_roi((0.1, 0.1, 1.5, 0.5), (100, 100)).normalized   # ... so this box passes, as in the reference
    pts = rs.rand(50, 3)
    box = mi.bbox_from_landmarks([mi.Landmark(*map(float, p)) for p in pts])
    assert box == (pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max())
    with pytest.raises(mi.MiError):
        mi.bbox_from_landmarks([mi.Landmark(0.1, 0.2, 0.0)])


def test_synthetic_graphs_oracles_agree_and_lower(tmp_path):
    """tests/synth_tflite.py: the flatbuffers it writes are read alike by the C oracle, the numpy evaluation and the product's reader,
    the two oracles agree on them bit for bit, and the planner lowers them at every level (host side, no GPU)."""
    import rs_face_detection_tflite_amd as mi
    import synth_tflite
    from oracle import pyoracle
    from oracle.np import evaluate, tfl3
    for name, (make, h, w) in synth_tflite.CASES.items():
        blob = make()
        p = tmp_path / (name + ".tflite")
        p.write_bytes(blob)
        x = np.random.RandomState(3).uniform(-1, 1, (2, h, w, 3)).astype(np.float32)
        refs = pyoracle.Model(str(p)).run(x, nthreads=2)
        ref2 = evaluate.run(tfl3.load(str(p)), x)
        assert len(refs) == len(ref2) >= 1
        for a, b in zip(refs, ref2):
            np.testing.assert_allclose(np.asarray(a).reshape(np.asarray(b).shape), np.asarray(b), rtol=0, atol=2e-5 * max(1.0, float(np.abs(b).max())))
        sizes = [int(mi.plan_describe(blob, lvl).splitlines()[0].split()[2].split("=")[1]) for lvl in range(6)]
        assert sizes == sorted(sizes, reverse=True) and sizes[5] < sizes[0], (name, sizes)
