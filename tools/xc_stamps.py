"""Where the first wave of xc_kernel's workgroup 0 spends a launch (development aid; stamps build: MI_VARIANT=xstamps
MI_EXTRA_FLAGS=-DMI_XC_STAMPS bash rs-face-detection-tflite_amd/build.sh).  usage: MI_XC_STAMPS=1 python tools/xc_stamps.py  (prints on stderr)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd.api as api
api.LIB_PATH = os.path.join(ROOT, "rs-face-detection-tflite_amd", "libmiface-xstamps.so")
import rs_face_detection_tflite_amd as mi
m = mi.Model(os.path.join(ROOT, "models", "face_detection_full_range.tflite"))
m.set_option("graph", 0)
x = torch.rand([128] + m.input_dims[1:], device="cuda")
m.run(x); m.run(x)
torch.cuda.synchronize()
