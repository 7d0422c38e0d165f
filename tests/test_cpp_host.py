"""The C++ host-side mirror of the reference API (include/mi_face.hpp) over the C ABI: compiles with g++ everywhere;
on a GPU box examples/pipeline.cpp reproduces the reference's rendered bbox pin on man.jpg."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, MODELS, ROOT

LIBDIR = os.path.join(ROOT, "rs-face-detection-tflite_amd")


def _build(tmp_path):
    exe = str(tmp_path / "pipeline")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pipeline.cpp"), "-o", exe,
           "-L", LIBDIR, "-lmiface", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_fails_loudly_without_gpu(tmp_path, mi, man_image):
    exe = _build(tmp_path)
    raw = tmp_path / "man.rgb"
    raw.write_bytes(man_image.tobytes())
    if mi.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, str(raw), "540", "360", MODELS], capture_output=True, text=True)
    assert r.returncode == 1 and "mi_face error -4" in r.stderr     # MI_EDEVICE, no silent CPU fallback


@pytest.mark.gpu
def test_cpp_pipeline_on_man_jpg(tmp_path, man_image):
    exe = _build(tmp_path)
    raw = tmp_path / "man.rgb"
    raw.write_bytes(man_image.tobytes())
    r = subprocess.run([exe, str(raw), "540", "360", MODELS], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = {l.split()[0]: l.split()[1:] for l in r.stdout.strip().splitlines()}
    assert out["faces"] == ["1"] and out["landmarks"] == ["468"]
    xmin, ymin, xmax, ymax = (float(v) for v in out["bbox"][:4])
    # the reference's own rendering of this image: left 195, top 74, 139 x 139 px
    assert int(xmin * 540) == 195 and int(ymin * 360) == 74 and int((xmax - xmin) * 540) == 139 and int((ymax - ymin) * 360) == 139
    gold = np.load(os.path.join(GOLDEN, "golden.npz"))
    np.testing.assert_allclose([float(v) for v in out["face_roi"]], gold["man_face_roi"][:5], atol=3e-3)
    np.testing.assert_allclose([float(v) for v in out["lm0"][:3]], gold["man_face_landmarks"][0], atol=3e-3)
    assert out["right_iris_center"][-4:] == ["contour", "15", "iris", "5"]
    np.testing.assert_allclose([float(out["right_iris_center"][0]), float(out["right_iris_center"][1])], gold["man_eye_right_iris"][0, :2], atol=5e-3)
    # update_face_landmarks_with_iris_results: mesh points 33 / 263 are replaced by the first contour landmark of each eye
    r = out["refined"]
    assert r[0] == "468" and r[2:4] == r[8:10] and r[5:7] == r[10:12]
    assert out["batch3"] == ["same"]        # FaceDetection::infer_batch (mi_fd_infer_images) == infer, frame by frame
    # the same flow from the encoded bytes (convert_image_to_mat on the GPU path), like the reference's test (lib.rs:23-24)
    r2 = subprocess.run([exe, os.path.join(GOLDEN, "man.jpg"), "0", "0", MODELS], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    out2 = {l.split()[0]: l.split()[1:] for l in r2.stdout.strip().splitlines()}
    assert out2["decoded"] == ["540", "360"]
    for key in ("faces", "bbox", "face_roi", "landmarks", "lm0", "refined"):
        assert out2[key] == out[key], key      # the raw file was written from the same decode: identical pixels, identical results
