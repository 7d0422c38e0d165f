"""Config 5 alone (full_range -> face_landmark -> 2x iris on the device), for profiling: python tools/pipe_bench.py [frames] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rs_face_detection_tflite_amd as mi
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rs = np.random.RandomState(0)
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB").resize((192, 192)))
frames = np.stack([np.roll(img, (int(rs.randint(-10, 11)), int(rs.randint(-10, 11))), axis=(0, 1)) if b % 2 else rs.randint(0, 256, img.shape).astype(np.uint8) for b in range(B)])
fd_ = torch.from_numpy(frames).cuda()
pipe = mi.Pipeline(mi.FaceDetectionModel.Full)
for kv in sys.argv[3:]:
    k, v = kv.split('=')
    pipe.set_option(k, int(v))
for _ in range(5): out = pipe.run(fd_)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(iters): out = pipe.run(fd_)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
print("pipeline %d frames: %.3f ms  %.0f frames/s  faces %d meshes %d" % (B, dt * 1e3, B / dt, int((out["face_counts"] > 0).sum().item()), int(out["present"].sum().item())))
