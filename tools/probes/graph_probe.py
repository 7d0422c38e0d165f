import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from PIL import Image
import rs_face_detection_tflite_amd as mi
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
def timed(fn, n=300, warm=30):
    for _ in range(warm): fn()
    ts=[]
    for _ in range(n):
        t=time.perf_counter(); fn(); ts.append(time.perf_counter()-t)
    ts.sort(); return round(ts[len(ts)//2]*1e6,1)
for g in (1, 0):
    fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera); fd.model.set_option("graph", g)
    fs = mi.FaceDetection(mi.FaceDetectionModel.Short); fs.model.set_option("graph", g)
    fl = mi.FaceLandmark(); fl.model.set_option("graph", g)
    ir = mi.IrisLandmark(); ir.model.set_option("graph", g)
    roi = mi.Rect(*[float(v) for v in gold["man_face_roi"][:5]], int(gold["man_face_roi"][5]))
    eye = mi.Rect(*[float(v) for v in gold["man_eye_left_roi"][:5]], int(gold["man_eye_left_roi"][5]))
    print("graph", g, "back", timed(lambda: fd.infer(img, None)), "short", timed(lambda: fs.infer(img, None)), "mesh", timed(lambda: fl.infer(img, roi)), "iris", timed(lambda: ir.infer(img, eye, False)))
