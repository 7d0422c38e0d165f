"""Per-stage time of the frame-resident stage programs of a model (development aid; needs the stamps build of the library:
MI_VARIANT=stamps MI_EXTRA_FLAGS=-DMI_RES_STAMPS bash rs-face-detection-tflite_amd/build.sh).
usage: python tools/res_stamps.py landmark 512"""
import ctypes as C, os, re, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-stamps.so")
import rs_face_detection_tflite_amd as mi
name, B = sys.argv[1], int(sys.argv[2])
files = {"landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite", "back": "face_detection_back.tflite", "full": "face_detection_full_range.tflite"}
m = mi.Model(os.path.join(ROOT, "models", files[name]))
m.set_option("graph", 0)
desc = m.describe()
res = [l for l in desc.splitlines() if l.startswith("resident")]
print("\n".join(l[:160] for l in res))
L = mi.lib()
buf = torch.zeros(64 << 20 >> 3, dtype=torch.int64, device="cuda")
L.mi_debug_set_res_stamps(C.c_void_p(buf.data_ptr()))
x = torch.rand([B] + m.input_dims[1:], device="cuda")
for _ in range(2):
    m.run(x)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.uint64)
# the LAST resident launch's stamps are intact at the front: [wg][stage][8]
last = res[-1]
nst = int(re.search(r"in (\d+) stages", last).group(1))
nwg = B
a = s[: nwg * nst * 8].reshape(nwg, nst, 8).astype(np.int64)
ok = a[:, :, 0] > 0
print("stages", nst, "workgroups with stamps", int(ok.all(axis=1).sum()))
a = a[ok.all(axis=1)]
dur = (a[:, :, 3] - a[:, :, 0])            # stage start -> stage end
tot = (a[:, -1, 3] - a[:, 0, 0])
print("per-workgroup total ticks: median %.0f (s_memtime, 100 MHz -> %.1f us)" % (np.median(tot), np.median(tot) / 100.0))
for k in range(nst):
    print("  stage %2d  %7.0f ticks  %5.1f%%   const-issue %5.0f  work %6.0f  tail %5.0f" % (
        k, np.median(dur[:, k]), 100 * np.median(dur[:, k]) / np.median(tot), np.median(a[:, k, 1] - a[:, k, 0]),
        np.median(a[:, k, 2] - a[:, k, 1]), np.median(a[:, k, 3] - a[:, k, 2])))
