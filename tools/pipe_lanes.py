"""Development experiment: config 5 (128 frames) split over L concurrent lanes = L Pipeline objects on L streams.
usage: python tools/pipe_lanes.py [frames=128]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mi = importlib.import_module("rs-face-detection-tflite_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
frames = torch.from_numpy(bench.make_rgb_frames(B, 0)).cuda()
for L in (1, 2, 4):
    pipes = [mi.Pipeline(mi.FaceDetectionModel.Full) for _ in range(L)]
    for p in pipes:
        p.set_option("graph", int(os.environ.get("LANES_GRAPH", "1")))
    streams = [torch.cuda.Stream() for _ in range(L)]
    parts = [frames[i * B // L:(i + 1) * B // L].contiguous() for i in range(L)]
    torch.cuda.synchronize()
    def step():
        outs = []
        for p, s, f in zip(pipes, streams, parts):
            outs.append(p.run(f, stream=s.cuda_stream))
        return outs
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    present = sum(int(o["present"].sum().item()) for o in outs)
    print("lanes %d: %.3f ms/step  %.0f frames/s  present %d" % (L, dt * 1e3, B / dt, present), flush=True)
    for p in pipes:
        p.close()
