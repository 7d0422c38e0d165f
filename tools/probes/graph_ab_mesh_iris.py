import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import rs_face_detection_tflite_amd as mi
from latency_probe import timed
from PIL import Image
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
fl = mi.FaceLandmark(); ir = mi.IrisLandmark()
roi = mi.Rect(*[float(v) for v in gold["man_face_roi"][:5]], int(gold["man_face_roi"][5]))
eye = mi.Rect(*[float(v) for v in gold["man_eye_left_roi"][:5]], int(gold["man_eye_left_roi"][5]))
for rep in range(2):
    for g in (1, 0):
        fl.model.set_option("graph", g); ir.model.set_option("graph", g)
        print("graph", g, "mesh", timed(lambda: fl.infer(img, roi), 300)["p50"], "iris", timed(lambda: ir.infer(img, eye, False), 300)["p50"])
