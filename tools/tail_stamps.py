"""Per-stage time of the several-frames-per-workgroup stage programs (tail_kernels.hip) of a model (development aid; needs the stamps
build of the library: MI_VARIANT=stamps MI_EXTRA_FLAGS=-DMI_TAIL_STAMPS bash rs-face-detection-tflite_amd/build.sh).
usage: python tools/tail_stamps.py landmark 512 [option=value ...]   (stamps of the LAST tail launch of the plan unless which=k)"""
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
m.set_option("fork", 0)
which = -1
G = 0
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "which":
        which = int(v)
    else:
        m.set_option(k, int(v))
        if k == "tail_g":
            G = int(v)
desc = m.describe()
res = [l for l in desc.splitlines() if "several frames per workgroup" in l]
print("\n".join(l[:170] for l in res))
L = mi.lib()
x = torch.rand([B] + m.input_dims[1:], device="cuda")
m.run(x)
torch.cuda.synchronize()
# every tail launch writes its stamps from the start of the buffer: to keep launch `which`, run the plan once per candidate and
# remember the buffer after each... simpler: the launches run in plan order, so give each its own run with the pointer moved past
# the previous ones (the kernel indexes from the pointer it is given)
bufs = []
for k in range(len(res)):
    bufs.append(torch.zeros(8 << 20 >> 3, dtype=torch.int64, device="cuda"))
# one shared buffer: the last launch wins at the front; for `which` we exploit that launches are serial and copy after a partial run
buf = bufs[0]
L.mi_debug_set_tail_stamps(C.c_void_p(buf.data_ptr()))
sel = res[which]
nst = int(re.search(r"in (\d+) stages", sel).group(1))
os.environ["MI_TAIL_STAMP_NST"] = str(nst)   # (launches with another stage count leave the buffer alone)
for _ in range(2):
    m.run(x)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.uint64)
lds = int(re.search(r"(\d+) B LDS per frame", sel).group(1))
if G == 0:
    G = max(1, min((160 * 1024 - 1024) // lds, B // 256))
nwg = (B + G - 1) // G
a = s[: nwg * nst * 4].reshape(nwg, nst, 4).astype(np.int64)
ok = (a[:, :, 0] > 0).all(axis=1)
print("stages", nst, "G", G, "workgroups", nwg, "with stamps", int(ok.sum()))
a = a[ok]
tot = a[:, -1, 3] - a[:, 0, 0]
print("per-workgroup total: median %.0f ticks of s_memtime (about 2.4 per ns on these boxes; compare with the launch's event time)" % np.median(tot))
for k in range(nst):
    print("  stage %2d  %6.0f ticks  %5.1f%%   depthwise / load %5.0f  units %6.0f  barrier %5.0f" % (
        k, np.median(a[:, k, 3] - a[:, k, 0]), 100 * np.median(a[:, k, 3] - a[:, k, 0]) / np.median(tot), np.median(a[:, k, 1] - a[:, k, 0]),
        np.median(a[:, k, 2] - a[:, k, 1]), np.median(a[:, k, 3] - a[:, k, 2])))
