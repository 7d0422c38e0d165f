"""Kernel time of the BackCamera / face-mesh networks at small batches with the row-pipelined chains run (a) block by block (option
small_chain) and (b) as pipelines: where the threshold belongs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rs_face_detection_tflite_amd as mi

def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for kind, size in (("back", 256), ("landmark", 192)):
    m = mi.Model(os.path.join(mi.DEFAULT_MODEL_DIR, {"back": "face_detection_back.tflite", "landmark": "face_landmark.tflite"}[kind]))
    for B in (1, 2, 4, 8, 12, 16, 24, 32):
        x = torch.rand((B, size, size, 3), device="cuda")
        row = []
        for sc in (64, 0):
            m.set_option("small_chain", sc)
            row.append(t(lambda: m.run(x)))
        print("%-9s batch %3d: block by block %7.1f us   pipelines %7.1f us" % (kind, B, row[0], row[1]))
    m.close()
