"""Per-stage time inside the single-launch plan (bandnet_kernels.hip) of a model (development aid; needs the stamps build of the
library: MI_VARIANT=stamps MI_EXTRA_FLAGS=-DMI_BAND_STAMPS bash rs-face-detection-tflite_amd/build.sh).
usage: python tools/band_stamps.py back [option=value ...]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-stamps.so")
import rs_face_detection_tflite_amd as mi
name = sys.argv[1]
files = {"landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite", "back": "face_detection_back.tflite", "front": "face_detection_front.tflite", "full": "face_detection_full_range.tflite"}
m = mi.Model(os.path.join(ROOT, "models", files[name]))
m.set_option("graph", 0)
m.set_option("band", 2)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    m.set_option(k, int(v))
L = mi.lib()
x = torch.rand([1] + m.input_dims[1:], device="cuda")
NWMAX, NST = 256, 64
buf = torch.zeros(NWMAX * NST * 10, dtype=torch.int64, device="cuda")
L.mi_debug_set_band_stamps(C.c_void_p(buf.data_ptr()))
for _ in range(3):
    buf.zero_()
    m.run(x)
    torch.cuda.synchronize()
a = buf.cpu().numpy().astype(np.int64).reshape(NWMAX, NST, 10)
TICK = 2.4e3   # s_memtime ticks per microsecond on these boxes (about)
print("stage: workgroups | recv  barrier  dw  pw  next-stage prep + barrier | stage total | gap to the next stage of the same workgroup   (microseconds, medians)")
tot = 0.0
first = a[:, :, 0]
start = first[first > 0].min()
end = a[:, :, 5].max()
for s in range(NST):
    on = a[:, s, 0] > 0
    if not on.any():
        continue
    d = a[on, s, :]
    ph = np.median(d[:, 1:6] - d[:, :5], axis=0) / TICK
    sub = [np.median(d[:, 6] - d[:, 0]) / TICK, np.median(d[:, 7] - d[:, 6]) / TICK, np.median(d[:, 1] - d[:, 7]) / TICK, np.median(d[:, 8] - d[:, 4]) / TICK]
    t = np.median(d[:, 5] - d[:, 0]) / TICK
    print("  %2d: %3d | %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f | recv = requests + next_active %5.2f + descriptor %5.2f + wait, check, LDS %5.2f; fetch %5.2f" % (
        s, int(on.sum()), ph[0], ph[1], ph[2], ph[3], ph[4], t, sub[0], sub[1], sub[2], sub[3]))
    tot += t
print("sum of stage medians %.1f us; first stamp to last stamp %.1f us" % (tot, (end - start) / TICK))
