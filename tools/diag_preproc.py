"""Where do device image_to_tensor and the oracle differ?  Prints, per case, the number of differing values and a few of them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rs_face_detection_tflite_amd as mi
from oracle import pyoracle as po
from PIL import Image
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
cases = [
    (None, (256, 256), True, (-1., 1.), False),
    (None, (128, 128), True, (-1., 1.), False),
    (None, (192, 192), True, (-1., 1.), False),
    (gold["man_face_roi"], (192, 192), False, (0., 1.), False),
    (gold["man_eye_right_roi"], (64, 64), True, (0., 1.), True),
    (gold["man_eye_left_roi"], (64, 64), True, (0., 1.), False),
    (np.array([300.0, 150.0, 333.0, 217.0, -0.7, 0]), (192, 192), True, (0., 1.), False),
    (np.array([0.5, 0.5, 0.4, 0.6, 0.2, 1]), (192, 192), False, (0., 1.), False),
]
for roi, size, keep, rng, flip in cases:
    r = mi.Rect(*[float(v) for v in roi[:5]], int(roi[5])) if roi is not None else None
    o = po.Rect(*[float(v) for v in roi[:5]], int(roi[5])) if roi is not None else None
    got, pad = mi.image_to_tensor(img, r, size, keep, rng, flip)
    ref, rpad = po.image_to_tensor(img, o, size, keep, rng, flip)
    lsb = (rng[1] - rng[0]) / 255.0
    d = (got - ref) / lsb
    idx = np.argwhere(np.abs(d) > 1e-3)
    print("case", None if roi is None else list(np.round(roi, 4)), size, keep, flip, "ndiff", len(idx), "of", d.size, "max", np.abs(d).max())
    for k in idx[:8]:
        print("   at", tuple(k), "got", got[tuple(k)] , "ref", ref[tuple(k)], "d/lsb", d[tuple(k)])
    ys = sorted(set(int(k[0]) for k in idx)); xs = sorted(set(int(k[1]) for k in idx))
    print("   rows", ys[:20], "cols", xs[:20])
