"""Config 5 (mi_pipeline_run, 128 frames of 192x192) with consecutive steps alternating between N pipeline handles on N streams against one
handle on one stream: the mesh / iris stages of batch n (64 faces, 128 eyes: launches of at most 128 workgroups) beside the detector of
batch n + 1.  Also config 3 (face mesh, 512 ROIs).  Experiment only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import rs_face_detection_tflite_amd as mi
import bench

B = 128
frames = torch.from_numpy(bench.make_rgb_frames(B, seed=0)).cuda()
NMAX = 3
# (streams are created apart from the handles: hipStreams share GPU_MAX_HW_QUEUES = 4 hardware queues in creation order, and a handle
# creates streams of its own — streams created right behind each handle all landed on ONE queue and nothing overlapped)
pipes = [mi.Pipeline(mi.FaceDetectionModel.Full) for _ in range(NMAX)]
skip = [torch.cuda.Stream() for _ in range(int(os.environ.get("PROBE_SKIP", "0")))]
hs = [(p, torch.cuda.Stream()) for p in pipes]
import ctypes as C
from rs_face_detection_tflite_amd import api
L = mi.lib()
res = {}
for k in range(NMAX):   # outputs allocated once: torch.zeros per call would put fill kernels on the NULL stream between the steps
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device="cuda")
    res[k] = dict(faces=z((B, 17), torch.float32), face_counts=z((B,), torch.int32), landmarks=z((B, 468, 3), torch.float32),
                  present=z((B,), torch.int32), eyes=z((B, 2, 76, 3), torch.float32))
torch.cuda.synchronize()
def step(k):
    p, st = hs[k]
    o = res[k]
    rc = L.mi_pipeline_run(p.h, C.c_void_p(frames.data_ptr()), B, 192, 192, 3 * 192, C.c_void_p(o["faces"].data_ptr()), C.c_void_p(o["face_counts"].data_ptr()),
                           C.c_void_p(o["landmarks"].data_ptr()), C.c_void_p(o["present"].data_ptr()), C.c_void_p(o["eyes"].data_ptr()), mi.MI_MEM_DEVICE, C.c_void_p(st.cuda_stream))
    if rc != 0: raise RuntimeError(L.mi_last_error().decode())
GRAPH = int(os.environ.get("PROBE_GRAPH", "1"))
for p, _ in hs:
    for m in p.models: m.set_option("graph", GRAPH)
print("graph replay" if GRAPH else "eager launches")
for n in (1, 2, 3, 1, 2, 3):
    for i in range(12): step(i % n)
    torch.cuda.synchronize()
    K = 120
    t0 = time.perf_counter()
    for i in range(K): step(i % n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("pipeline, %d handle(s): %.4f ms per step, %.0f frames/s" % (n, dt * 1e3, B / dt))
ref = {k: v.clone() if hasattr(v, "clone") else v for k, v in res[0].items()}
for k in (1, 2):
    for name in ref:
        assert torch.equal(ref[name], res[k][name]), name
print("results of the handles are identical")
x = torch.from_numpy(bench.make_rois(512, seed=0)).cuda()
fls = [(mi.FaceLandmark(), torch.cuda.Stream()) for _ in range(2)]
for n in (1, 2, 1, 2):
    for i in range(12): fls[i % n][0].infer_tensor(x, stream=fls[i % n][1].cuda_stream)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for i in range(K): fls[i % n][0].infer_tensor(x, stream=fls[i % n][1].cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("face mesh 512 ROIs, %d handle(s): %.4f ms per step, %.0f ROIs/s" % (n, dt * 1e3, 512 / dt))
