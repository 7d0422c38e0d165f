"""Development aid: time the network (+post-processing) on the GPU for a few option sets."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rs_face_detection_tflite_amd as m

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "back"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    kinds = {"back": m.FaceDetectionModel.BackCamera, "front": m.FaceDetectionModel.FrontCamera, "full": m.FaceDetectionModel.Full}
    fd = m.FaceDetection(kinds[name])
    W, H = fd.input_size
    x = torch.rand((B, H, W, 3), device="cuda") * 2 - 1
    torch.cuda.synchronize()
    for fuse in [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]:
        for chunk in (0, 64, 32, 16):
            for graph in (0, 1):
                fd.model.set_option("fuse", fuse); fd.model.set_option("chunk", chunk); fd.model.set_option("graph", graph)
                for _ in range(3): fd.infer_tensor(x, cap=16)
                torch.cuda.synchronize()
                t = time.time(); n = 10
                for _ in range(n): fd.infer_tensor(x, cap=16)
                torch.cuda.synchronize()
                dt = (time.time() - t) / n
                print(json.dumps(dict(model=name, B=B, fuse=fuse, chunk=chunk, graph=graph, ms=round(dt * 1e3, 3), fps=round(B / dt))), flush=True)
main()
