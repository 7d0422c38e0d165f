"""Two-slot host feed of the BackCamera detector (mi_fd_submit_images / mi_fd_collect) for a timeline: run under
   rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/feed -o feed --output-format csv -- python3 tools/host_feed_probe.py
and read the copy / kernel intervals with tools/host_feed_timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rs_face_detection_tflite_amd as mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
u8 = np.random.RandomState(2).randint(0, 256, (256, 256, 256, 3)).astype(np.uint8)
u8[1::2] = gold["man_back_u8"].astype(np.uint8)
pins = [mi.PinnedBuffer(u8.shape) for _ in range(2)]
for p in pins:
    p.array[...] = u8
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fd.submit_images(0, pins[0].array, cap=16)
t = time.perf_counter()
for b in range(1, nb + 1):
    t0 = time.perf_counter()
    fd.submit_images(b & 1, pins[b & 1].array, cap=16)
    t1 = time.perf_counter()
    out, counts = fd.collect((b - 1) & 1)
    t2 = time.perf_counter()
    if b <= 6:
        print("batch %d: submit %.3f ms, collect %.3f ms" % (b, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
dt = (time.perf_counter() - t) / nb
fd.collect(nb & 1)
print("two slots: %.3f ms per batch, %.0f frames/s" % (dt * 1e3, 256 / dt))
