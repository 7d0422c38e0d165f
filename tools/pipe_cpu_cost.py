"""Development: host-side enqueue cost of Pipeline.run against its device time (config 5), graphs on / off."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mi = importlib.import_module("rs-face-detection-tflite_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = torch.from_numpy(bench.make_rgb_frames(B, 0)).cuda()
s = torch.cuda.Stream()
for graph in (1, 0):
    p = mi.Pipeline(mi.FaceDetectionModel.Full)
    p.set_option("graph", graph)
    for _ in range(5):
        p.run(frames, stream=s.cuda_stream)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        p.run(frames, stream=s.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # one call at a time: host enqueue, then wait
    enq = dev = 0.0
    for _ in range(20):
        a = time.perf_counter(); p.run(frames, stream=s.cuda_stream); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
        enq += b - a; dev += c - a
    print("graph %d: enqueue loop %.3f ms/step, total %.3f ms/step; single call: enqueue %.3f ms, enqueue+wait %.3f ms" %
          (graph, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, enq / 20 * 1e3, dev / 20 * 1e3), flush=True)
    p.close()
