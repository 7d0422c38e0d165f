"""Per-launch event times of one model's plan (mi_model_profile): python tools/profile_model.py iris 1024 [option=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rs_face_detection_tflite_amd as mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"back": "face_detection_back.tflite", "front": "face_detection_front.tflite", "short": "face_detection_short_range.tflite", "full": "face_detection_full_range.tflite",
         "sparse": "face_detection_full_range_sparse.tflite", "landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite"}
name, B = sys.argv[1], int(sys.argv[2])
m = mi.Model(os.path.join(ROOT, "models", FILES[name]))
for kv in sys.argv[3:]:  # engine options, e.g. fuse=4 reuse=0
    k, v = kv.split("=")
    m.set_option(k, int(v))
h, w = m.input_dims[1], m.input_dims[2]
x = torch.rand((B, h, w, 3), device="cuda")
recs = m.profile(x, reps=5)
tot = 0.0
for r in recs:
    tot += r["ms"]
    print("%8.4f ms  %-40s %s" % (r["ms"], r["kernel"], r.get("shape", "")))
print("total %.3f ms for %d frames, %d launches" % (tot, B, len(recs)))
