"""Where a wave of the register-resident bottleneck kernel spends its time (development aid; needs the stamps build:
MI_VARIANT=bstamps MI_EXTRA_FLAGS=-DMI_BNECK_STAMPS bash rs-face-detection-tflite_amd/build.sh).
usage: python tools/bneck_stamps.py 1024 [first|last]   (first: the 32x32 row-band launches' last one; last: the 16x16 run)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-bstamps.so")
import rs_face_detection_tflite_amd as mi
B = int(sys.argv[1])
m = mi.Model(os.path.join(ROOT, "models", "iris_landmark.tflite"))
m.set_option("graph", 0)
L = mi.lib()
buf = torch.zeros((1 << 20) + B * 64, dtype=torch.int64, device="cuda")
L.mi_debug_set_bneck_stamps(C.c_void_p(buf.data_ptr()))
x = torch.rand([B] + m.input_dims[1:], device="cuda")
for _ in range(2):
    m.run(x)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.int64)
band = raw[: 256 * 64].reshape(256, 8, 8)      # one persistent workgroup per CU: the stamps of its LAST band
run = raw[1 << 20:].reshape(B, 8, 8)
names = ["x -> registers (+ barrier)", "pw1 own pixels", "pw1 halo rows", "barrier", "dw + pw2", "epilogue", "store"]
def show(title, a):
    a = a[(a[:, 0, 7] > 0)]
    print(title, "(%d workgroups)" % len(a))
    for w in (0, 3, 7):
        d = np.diff(a[:, w, :], axis=1)
        print(" wave %d: " % w + "  ".join("%s %.0f" % (names[k], np.median(d[:, k])) for k in range(7)), " total %.0f" % np.median(a[:, w, 7] - a[:, w, 0]))
show("32x32 row bands: last band of each persistent workgroup; stamp 0 = kernel start; ticks = shader clock", band)
show("16x16 run (stamps 2..5 are those of its last block)", run)
