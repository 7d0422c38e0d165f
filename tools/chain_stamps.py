"""Per-stage time of the frame-resident chain launch of a model (development aid; needs the stamps build of the library:
MI_VARIANT=cstamps MI_EXTRA_FLAGS=-DMI_CHAIN_STAMPS bash rs-face-detection-tflite_amd/build.sh).
usage: python tools/chain_stamps.py back 256"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-cstamps.so")
import rs_face_detection_tflite_amd as mi
name, B = sys.argv[1], int(sys.argv[2])
files = {"landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite", "back": "face_detection_back.tflite", "front": "face_detection_front.tflite"}
m = mi.Model(os.path.join(ROOT, "models", files[name]))
m.set_option("graph", 0)
print("\n".join(l[:200] for l in m.describe().splitlines() if "frame resident in LDS" in l))
L = mi.lib()
buf = torch.zeros(B * 24, dtype=torch.int64, device="cuda")
L.mi_debug_set_chain_stamps(C.c_void_p(buf.data_ptr()))
x = torch.rand([B] + m.input_dims[1:], device="cuda")
for _ in range(3):
    m.run(x)
torch.cuda.synchronize()
a = buf.cpu().numpy().astype(np.int64).reshape(B, 24)  # the last chain launch's stamps
names = ["zero+pre/load"] + ["block %d" % k for k in range(8)] + ["write-out", "post", "gap", "head 0", "head 1"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
tot = np.median(a[:, 14] - a[:, 0])
print("per-workgroup total: median %.0f ticks (s_memtime 100 MHz -> %.1f us); launch span %.1f us" % (tot, tot / 100.0, (a[:, 14].max() - a[:, 0].min()) / 100.0))
prev = 0
for k in range(1, 15):
    if (a[:, k] == 0).all():
        continue
    d = np.median(a[:, k] - a[:, prev])
    print("  -> stamp %2d  %6.0f ticks  %5.1f us  %5.1f%%" % (k, d, d / 100.0, 100 * d / tot))
    prev = k
if (a[:, 16] > 0).all():
    print("  pre: pass 0 staged at +%.0f ticks, pass 1 staged at +%.0f ticks, stage ends at +%.0f" % (
        np.median(a[:, 16] - a[:, 0]), np.median(a[:, 17] - a[:, 0]) if (a[:, 17] > 0).all() else -1, np.median(a[:, 1] - a[:, 0])))
