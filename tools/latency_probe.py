"""Per-call latency of the reference's own operating point: one image per call through the three `infer`s
(face_detection.rs:205-267, face_landmark.rs:232-306, iris_landmark.rs:158-248) and JPEG bytes -> detections (lib.rs:24-40).
Prints p50 / p99 / mean in microseconds.  bench.py reports the same figures under `single_image_latency_us`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rs_face_detection_tflite_amd as mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pct(ts):
    a = np.sort(np.asarray(ts)) * 1e6
    return {"p50": round(float(a[len(a) // 2]), 1), "p99": round(float(a[min(len(a) - 1, int(len(a) * 0.99))]), 1), "mean": round(float(a.mean()), 1), "calls": len(a)}


def timed(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return pct(ts)


def measure(n=300):
    from PIL import Image
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    jpg = open(os.path.join(ROOT, "tests", "golden", "man.jpg"), "rb").read()
    img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
    out = {}
    for kind in ("BackCamera", "Short", "Full", "FullSparse"):
        fd = mi.FaceDetection(getattr(mi.FaceDetectionModel, kind))
        out["FaceDetection::infer %s (man.jpg 540x360, Mat in host memory)" % kind] = timed(lambda: fd.infer(img, None), n)
        if kind == "BackCamera":
            out["convert_image_to_mat + FaceDetection::infer BackCamera (JPEG bytes -> detections)"] = timed(lambda: fd.infer(mi.convert_image_to_mat(jpg), None), n)
            # the same for a STREAM of pictures (mi_fd_submit_jpeg / mi_fd_collect_jpeg, two slots): Huffman decoding of picture n + 1 on this thread
            # while the device runs picture n; one "call" = submit n + 1, collect n — the sustained time per picture
            fd.submit_jpeg(0, jpg)
            state = {"k": 0}

            def stream_step():
                k = state["k"]
                fd.submit_jpeg(1 - k, jpg)
                dets = fd.collect_jpeg(k)
                state["k"] = 1 - k
                return dets
            out["streamed JPEG bytes -> detections, BackCamera (mi_fd_submit_jpeg / collect_jpeg, two slots; sustained per picture)"] = timed(stream_step, n)
            fd.collect_jpeg(state["k"])
            out["one JPEG at a time through the same entries (submit + collect of one slot)"] = timed(lambda: (fd.submit_jpeg(0, jpg), fd.collect_jpeg(0))[1], n)
        fd.close()
    fl = mi.FaceLandmark()
    roi = mi.Rect(*[float(v) for v in gold["man_face_roi"][:5]], int(gold["man_face_roi"][5]))
    out["FaceLandmark::infer (man.jpg, face ROI)"] = timed(lambda: fl.infer(img, roi), n)
    ir = mi.IrisLandmark()
    eye = mi.Rect(*[float(v) for v in gold["man_eye_left_roi"][:5]], int(gold["man_eye_left_roi"][5]))
    out["IrisLandmark::infer (man.jpg, left-eye ROI)"] = timed(lambda: ir.infer(img, eye, False), n)
    # the whole flow of lib.rs:24-40 on one picture: detector -> mesh of the first face -> both eyes, one call, every stage on the device
    pipe = mi.Pipeline(mi.FaceDetectionModel.BackCamera)
    one = np.ascontiguousarray(img[None])
    out["lib.rs flow on one picture: mi_pipeline_run, batch 1 (detector -> mesh -> 2 x iris)"] = timed(lambda: pipe.run(one), n)
    pipe_full = mi.Pipeline(mi.FaceDetectionModel.Full)   # (the same flow with the full-range detector: its trunk is one launch since round 6)
    out["lib.rs flow on one picture with FaceDetectionModel::Full (mi_pipeline_run, batch 1)"] = timed(lambda: pipe_full.run(one), n)
    pipe_full.close()
    # ... and for a stream of ENCODED pictures (lib.rs:18-40 from the bytes; mi_pipeline_submit_jpeg / collect_jpeg, two slots): sustained per picture
    pipe.submit_jpeg(0, jpg)
    state = {"k": 0}

    def flow_step():
        k = state["k"]
        pipe.submit_jpeg(1 - k, jpg)
        res = pipe.collect_jpeg(k)
        state["k"] = 1 - k
        return res
    out["lib.rs flow from JPEG bytes, streamed (mi_pipeline_submit_jpeg / collect_jpeg, two slots; sustained per picture)"] = timed(flow_step, n)
    pipe.collect_jpeg(state["k"])
    out["lib.rs flow from JPEG bytes, one picture at a time (convert_image_to_mat + mi_pipeline_run)"] = timed(lambda: pipe.run(np.ascontiguousarray(mi.convert_image_to_mat(jpg)[None])), n)
    pipe.close()
    return out


def measure_c_abi(n=300):
    """The same three calls timed at the C ABI itself (arguments marshalled once, outside the timed region): what a Rust / C++ caller of
    include/mi_face.h sees, without the Python wrapper's per-call conversion of the results into Detection / Landmark objects."""
    import ctypes as C
    from PIL import Image
    from rs_face_detection_tflite_amd import api
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    img = np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB")))
    h, w = img.shape[:2]
    p = C.c_void_p(img.ctypes.data)
    out = {}
    for kind in ("BackCamera", "Short", "Full", "FullSparse"):
        fd = mi.FaceDetection(getattr(mi.FaceDetectionModel, kind))
        dets, cnt = (api.CDetection * 256)(), C.c_int()
        def call():
            rc = fd.L.mi_fd_infer_image(fd.h, p, w, h, 3 * w, None, dets, 256, C.byref(cnt))
            assert rc == 0
        out["mi_fd_infer_image %s" % kind] = timed(call, n)
        fd.close()
    fl = mi.FaceLandmark()
    roi = mi.Rect(*[float(v) for v in gold["man_face_roi"][:5]], int(gold["man_face_roi"][5]))
    lms, cnt = (api.CLandmark * api.NUM_FACE_LANDMARKS)(), C.c_int()
    def call_fl():
        rc = fl.L.mi_fl_infer_image(fl.h, p, w, h, 3 * w, C.byref(roi), lms, api.NUM_FACE_LANDMARKS, C.byref(cnt))
        assert rc == 0
    out["mi_fl_infer_image"] = timed(call_fl, n)
    ir = mi.IrisLandmark()
    eye = mi.Rect(*[float(v) for v in gold["man_eye_left_roi"][:5]], int(gold["man_eye_left_roi"][5]))
    c71, i5 = (api.CLandmark * 71)(), (api.CLandmark * 5)()
    def call_ir():
        rc = ir.L.mi_iris_infer_image(ir.h, p, w, h, 3 * w, C.byref(eye), 0, c71, i5)
        assert rc == 0
    out["mi_iris_infer_image"] = timed(call_ir, n)
    return out


if __name__ == "__main__":
    for k, v in measure().items():
        print(json.dumps({"call": k, **v}))
    for k, v in measure_c_abi().items():
        print(json.dumps({"call": k, "timed_at": "C ABI", **v}))
