"""The single-launch plan (bandnet_kernels.hip, option band=2) against the batched plan (band=0) on the same random frames:
python tools/band_check.py back 1 2 4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rs_face_detection_tflite_amd as mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"back": "face_detection_back.tflite", "front": "face_detection_front.tflite", "short": "face_detection_short_range.tflite", "full": "face_detection_full_range.tflite",
         "landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite"}
name = sys.argv[1]
rng = np.random.default_rng(5)
bad = 0
for B in [int(v) for v in sys.argv[2:]] or [1]:
    m = mi.Model(os.path.join(ROOT, "models", FILES[name]))
    h, w = m.input_dims[1], m.input_dims[2]
    x = rng.uniform(-1, 1, (B, h, w, 3)).astype(np.float32)
    m.set_option("band", 0)
    ref = m.run(x)
    m.set_option("band", 2)
    for rep in range(3):
        out = m.run(x)
        for k, (a, b) in enumerate(zip(ref, out)):
            d = float(np.abs(a - b).max())
            s = float(np.abs(a).max())
            print("%s B=%d run %d output %d: max |diff| %.3g of max |ref| %.3g" % (name, B, rep, k, d, s))
            if not d <= 1e-4 * max(1.0, s):
                bad += 1
    m.close()
print("MISMATCHES" if bad else "ok")
sys.exit(1 if bad else 0)
