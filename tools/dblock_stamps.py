"""Where a wave of the double-block kernel spends a band (development aid; stamps build:
MI_VARIANT=dstamps MI_EXTRA_FLAGS=-DMI_DBLOCK_STAMPS bash rs-face-detection-tflite_amd/build.sh).  usage: python tools/dblock_stamps.py 128
The stamps of the LAST stamped dblock launch of full_range survive; MI_DB_H=96 (48, 24) stamps only the launches on frames of that height."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-dstamps.so")
import rs_face_detection_tflite_amd as mi
B = int(sys.argv[1])
m = mi.Model(os.path.join(ROOT, "models", "face_detection_full_range.tflite"))
m.set_option("graph", 0)
L = mi.lib()
buf = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
x = torch.rand([B] + m.input_dims[1:], device="cuda")
m.run(x)
L.mi_debug_set_dblock_stamps(C.c_void_p(buf.data_ptr()))
m.run(x)
torch.cuda.synchronize()
a = buf.cpu().numpy().astype(np.int64).reshape(256, 8, 8)
names = ["x -> LDS", "barrier", "stage 1 (a)", "barrier", "stage 2 (y)"]
print("last band of each workgroup, last dblock launch; ticks = shader clock; kernel start -> end of last band: %.0f" % np.median(a[:, 0, 6] - a[:, 0, 0]))
for w in (0, 3, 7):
    d = np.diff(a[:, w, 1:7], axis=1)
    print(" wave %d: " % w + "  ".join("%s %.0f" % (names[k], np.median(d[:, k])) for k in range(5)), " band %.0f" % np.median(a[:, w, 6] - a[:, w, 1]))
