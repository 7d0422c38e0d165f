"""Throughput of BASELINE config 2 with the steps alternating between TWO detector handles on two streams (consecutive batches overlap at
the kernels' tails) against one handle on one stream.  Experiment only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
import rs_face_detection_tflite_amd as mi
import bench

B, cap = 256, 16
x = torch.from_numpy(bench.make_frames(B, seed=0)).cuda()
hs = []
for k in range(2):
    fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
    st = torch.cuda.Stream()
    out = torch.zeros((B, cap, 17), dtype=torch.float32, device="cuda")
    cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    hs.append((fd, st, out, cnt))
L = mi.lib()
def step(k):
    fd, st, out, cnt = hs[k]
    if L.mi_fd_infer_tensor(fd.h, C.c_void_p(x.data_ptr()), B, None, C.c_void_p(out.data_ptr()), cap, C.c_void_p(cnt.data_ptr()), mi.MI_MEM_DEVICE, C.c_void_p(st.cuda_stream)) != 0:
        raise RuntimeError(L.mi_last_error().decode())
for mode in ("one stream", "two streams", "one stream", "two streams"):
    for _ in range(10): step(0); step(1 if mode == "two streams" else 0)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K): step(i & 1 if mode == "two streams" else 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-11s: %.4f ms per step, %.0f faces/s" % (mode, dt * 1e3, B / dt))
