"""Host-side step times of one mi_pipeline_run call on one picture (MI_PIPE_TRACE=1 prints them to stderr): run a few calls, the last one is representative."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
import rs_face_detection_tflite_amd as mi
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
p = mi.Pipeline(mi.FaceDetectionModel.BackCamera)
one = np.ascontiguousarray(img[None])
for _ in range(30):
    p.run(one)
print("---- last call", file=sys.stderr)
p.run(one)
p.close()
