"""Diagnostic: several threads on one FaceDetection handle through the asynchronous device entry point."""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import rs_face_detection_tflite_amd as mi
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
face = (gold["man_back_u8"].astype(np.float64) * 2.0 / 255.0 - 1.0).astype(np.float32)
from PIL import Image
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
fd.model.set_option("graph", graph)
batches = [np.stack([np.roll(face, (3 * k + j, -5 * j), axis=(0, 1)) for j in range(3)]) for k in range(4)]
want = [fd.infer_tensor(b, cap=8) for b in batches]
dev = [torch.from_numpy(b).cuda() for b in batches]
streams = [torch.cuda.Stream() for _ in range(4)]
outs = [torch.zeros((3, 8, 17), dtype=torch.float32, device="cuda") for _ in range(4)]
cnts = [torch.zeros((3,), dtype=torch.int32, device="cuda") for _ in range(4)]
torch.cuda.synchronize()
bad = []
def worker(k):
    for it in range(40):
        if mode in ("mixed", "alloc", "full"):
            o, c = fd.infer_tensor(batches[k], cap=8)
            if not (np.array_equal(o, want[k][0]) and np.array_equal(c, want[k][1])):
                bad.append((k, it, "host", [j for j in range(4) if np.array_equal(o, want[j][0])]))
        if mode in ("alloc", "full"):
            if mode == "full":
                fd.infer(img, None)
            ot = torch.zeros((3, 8, 17), dtype=torch.float32, device="cuda"); ct = torch.zeros((3,), dtype=torch.int32, device="cuda")
        else:
            ot, ct = outs[k], cnts[k]
            ot.zero_(); ct.zero_()
        torch.cuda.synchronize()
        fd.infer_tensor(dev[k], cap=8, out=ot, counts=ct, stream=streams[k].cuda_stream)
        streams[k].synchronize()
        o = ot.cpu().numpy(); c = ct.cpu().numpy()
        if not (np.array_equal(o, want[k][0]) and np.array_equal(c, want[k][1])):
            bad.append((k, it, "dev", [j for j in range(4) if np.array_equal(o, want[j][0])], c.tolist(), want[k][1].tolist(), float(np.abs(o - want[k][0]).max()), np.argwhere(o != want[k][0])[:6].tolist()))
ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print(mode, "graph", graph, "bad", len(bad), bad[:10])
