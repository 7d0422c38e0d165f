"""Secondary BASELINE configs (parity-test cases, not the bench line): config 3 (FaceLandmark 192x192, batch 512) and
config 5 (full_range detector -> mesh -> iris pipeline, 128 frames/GPU).  Prints one JSON line per config."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rs_face_detection_tflite_amd as mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))

def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

# config 3
fl = mi.FaceLandmark()
B = 512
rs = np.random.RandomState(0)
x = rs.uniform(0, 1, (B, 192, 192, 3)).astype(np.float32)
x[1::2] = gold["man_face_u8"].astype(np.float32) / 255.0
xd = torch.from_numpy(x).cuda()
dt = timeit(lambda: fl.infer_tensor(xd))
print(json.dumps({"config": "FaceLandmark 192x192 batch 512 (net + projection + face flag)", "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(B / dt)}))
recs = fl.model.profile(xd, reps=3)
by = {}
for r in recs:
    k = by.setdefault(r["kernel"], [0.0, 0]); k[0] += r["ms"]; k[1] += 1
print("   ", {k: (round(v[0], 3), v[1]) for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])})

# iris alone
ir = mi.IrisLandmark()
xe = torch.rand((1024, 64, 64, 3), device="cuda")
dt = timeit(lambda: ir.infer_tensor(xe))
print(json.dumps({"config": "IrisLandmark 64x64 batch 1024", "ms_per_batch": round(dt * 1e3, 3), "eyes_per_s": round(1024 / dt)}))
recs = ir.model.profile(xe, reps=3)
by = {}
for r in recs:
    k = by.setdefault(r["kernel"], [0.0, 0]); k[0] += r["ms"]; k[1] += 1
print("   ", {k: (round(v[0], 3), v[1]) for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])})

# config 5
from PIL import Image
img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB").resize((192, 192)))
B = 128
frames = np.stack([np.roll(img, (int(rs.randint(-10, 11)), int(rs.randint(-10, 11))), axis=(0, 1)) if b % 2 else rs.randint(0, 256, img.shape).astype(np.uint8) for b in range(B)])
fd_ = torch.from_numpy(frames).cuda()
pipe = mi.Pipeline(mi.FaceDetectionModel.Full)
out = pipe.run(fd_)
torch.cuda.synchronize()
dt = timeit(lambda: pipe.run(fd_))
print(json.dumps({"config": "full_range 192x192 -> face_landmark -> 2x iris, 128 frames/GPU, all on device", "ms_per_batch": round(dt * 1e3, 3),
                  "frames_per_s": round(B / dt), "faces_found": int((out["face_counts"] > 0).sum().item()), "meshes": int(out["present"].sum().item())}))

# convert_image_to_mat (utils.rs:8-21): JPEG bytes -> RGB, host entropy decoding + device sample arithmetic
jpg = open(os.path.join(ROOT, "tests", "golden", "man.jpg"), "rb").read()
for to_dev in (False, True):
    for _ in range(3): mi.convert_image_to_mat(jpg, to_device=to_dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): mi.convert_image_to_mat(jpg, to_device=to_dev)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(json.dumps({"config": "convert_image_to_mat man.jpg 540x360 4:2:0 (%s)" % ("RGB left in HBM" if to_dev else "RGB copied to host"), "ms_per_image": round(dt * 1e3, 3), "images_per_s": round(1 / dt)}))

# config 2 with the host boundary: pageable host frames in, detections out (H2D + net + post-processing + D2H per call) — never bench.py's `value`
fd = mi.FaceDetection(mi.FaceDetectionModel.BackCamera)
xh = np.random.RandomState(1).uniform(-1, 1, (256, 256, 256, 3)).astype(np.float32)
xh[1::2] = (gold["man_back_u8"].astype(np.float32) * 2.0 / 255.0 - 1.0)
xp = torch.from_numpy(xh).pin_memory().numpy()
for name, arr in (("pageable", xh), ("pinned", xp)):
    for _ in range(2): fd.infer_tensor(arr, cap=16)
    t = time.perf_counter()
    for _ in range(5): fd.infer_tensor(arr, cap=16)
    dt = (time.perf_counter() - t) / 5
    print(json.dumps({"config": "BackCamera 256 frames from %s HOST memory (H2D 201 MB + net + NMS + D2H per call)" % name, "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(256 / dt)}))

# the same from u8 frames (what the reference's callers hold: Mat 8UC3, utils.rs:8-21) — mi_fd_infer_images: 50 MB instead of 201 MB per
# batch over the bus, image_to_tensor on the device; then the two-slot host feed (mi_fd_submit_images / mi_fd_collect): the copy of
# batch n+1 on its own stream while the kernels of batch n run
u8 = np.random.RandomState(2).randint(0, 256, (256, 256, 256, 3)).astype(np.uint8)
u8[1::2] = gold["man_back_u8"].astype(np.uint8)
pins = [mi.PinnedBuffer(u8.shape) for _ in range(2)]
for p in pins: p.array[...] = u8
ud = torch.from_numpy(u8).cuda()
dt = timeit(lambda: fd.infer_images(ud, cap=16))
print(json.dumps({"config": "BackCamera 256 u8 frames resident in HBM (image_to_tensor + net + NMS, mi_fd_infer_images)", "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(256 / dt)}))
for name, arr in (("pageable", u8), ("pinned", pins[0].array)):
    for _ in range(2): fd.infer_images(arr, cap=16)
    t = time.perf_counter()
    for _ in range(10): fd.infer_images(arr, cap=16)
    dt = (time.perf_counter() - t) / 10
    print(json.dumps({"config": "BackCamera 256 u8 frames from %s HOST memory, one call at a time (H2D 50 MB + image_to_tensor + net + NMS + D2H)" % name,
                      "ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(256 / dt), "h2d_GBps_if_copy_only": round(u8.nbytes / dt / 1e9, 1)}))
nb = 40
fd.submit_images(0, pins[0].array, cap=16)
t = time.perf_counter()
for b in range(1, nb + 1):
    fd.submit_images(b & 1, pins[b & 1].array, cap=16)
    out, counts = fd.collect((b - 1) & 1)
dt = (time.perf_counter() - t) / nb
fd.collect(nb & 1)
# the copy alone, for the H2D rate of this box
hs = torch.cuda.Stream()
dst = torch.empty_like(ud)
src = torch.from_numpy(pins[0].array)
with torch.cuda.stream(hs):
    for _ in range(3): dst.copy_(src, non_blocking=True)
    hs.synchronize(); t = time.perf_counter()
    for _ in range(10): dst.copy_(src, non_blocking=True)
    hs.synchronize(); dc = (time.perf_counter() - t) / 10
print(json.dumps({"config": "BackCamera 256 u8 frames from pinned HOST memory, two slots (copy of batch n+1 overlaps the kernels of batch n)", "ms_per_batch": round(dt * 1e3, 3),
                  "frames_per_s": round(256 / dt), "faces_in_last_batch": int((counts > 0).sum()), "h2d_copy_alone_ms": round(dc * 1e3, 3), "h2d_GBps": round(u8.nbytes / dc / 1e9, 1)}))

# round 5: the mesh and the iris network from u8 frames + ROIs (mi_fl_infer_images / mi_iris_infer_images: the warp to 192x192 / 64x64 on
# the device) against their f32-crop entries from host memory.  512 ROIs = 64 camera frames of 540x360 with 8 face ROIs each (the
# frames and the ROIs are what crosses the bus: 37 MB instead of 226 MB); then 512 frames of 192x192 with one ROI each.
man = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
Hm, Wm = man.shape[:2]
rsr = np.random.RandomState(3)
base = gold["man_face_roi"]
def jitter(k):
    r = base.copy()
    r[0] += rsr.uniform(-0.02, 0.02); r[1] += rsr.uniform(-0.02, 0.02); r[4] += rsr.uniform(-0.3, 0.3)
    return mi.Rect(*[float(v) for v in r[:5]], int(r[5]))
rois = [jitter(k) for k in range(512)]
frames64 = np.stack([np.roll(man, (int(rsr.randint(-6, 7)), int(rsr.randint(-6, 7))), axis=(0, 1)) for _ in range(64)])
pin = mi.PinnedBuffer(frames64.shape); pin.array[...] = frames64
for _ in range(3): lm, pres, _f = fl.infer_images(pin.array, rois, items_per_frame=8)
t = time.perf_counter()
for _ in range(20): lm, pres, _f = fl.infer_images(pin.array, rois, items_per_frame=8)
dt = (time.perf_counter() - t) / 20
print(json.dumps({"config": "FaceLandmark 512 ROIs on 64 u8 frames 540x360 from pinned HOST memory (mi_fl_infer_images: H2D 37 MB + device warp + net + projection + D2H)",
                  "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(512 / dt), "faces_present": int(pres.sum())}))
crops = np.stack([mi.image_to_tensor(man, rois[k], (192, 192), False, (0., 1.), False)[0] for k in range(0, 512, 8)] * 8)
sizes = np.array([[Wm, Hm]] * 512, np.int32)
cp = torch.from_numpy(crops).pin_memory().numpy()
for _ in range(2): fl.infer_tensor(cp, rois, sizes)
t = time.perf_counter()
for _ in range(5): fl.infer_tensor(cp, rois, sizes)
dt = (time.perf_counter() - t) / 5
print(json.dumps({"config": "FaceLandmark 512 f32 crops from pinned HOST memory (mi_fl_infer_tensor: H2D 226 MB + net + projection + D2H)", "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(512 / dt)}))
f192 = np.stack([np.asarray(Image.fromarray(man).resize((192, 192)))] * 512)
pin2 = mi.PinnedBuffer(f192.shape); pin2.array[...] = f192
r192 = [mi.Rect(0.5, 0.42, 0.5, 0.75, 0.0, 1)] * 512
for _ in range(3): fl.infer_images(pin2.array, r192)
t = time.perf_counter()
for _ in range(20): fl.infer_images(pin2.array, r192)
dt = (time.perf_counter() - t) / 20
print(json.dumps({"config": "FaceLandmark 512 ROIs on 512 u8 frames 192x192 from pinned HOST memory (mi_fl_infer_images: H2D 57 MB + device warp + net + projection + D2H)",
                  "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(512 / dt)}))
pinb = [mi.PinnedBuffer(frames64.shape) for _ in range(2)]
for pb in pinb: pb.array[...] = frames64
import ctypes as _C
rarr = (mi.Rect * 512)(*rois)
nb = 40
fl.submit_images(0, pinb[0].array, rarr, items_per_frame=8)
t = time.perf_counter()
for b in range(1, nb + 1):
    fl.submit_images(b & 1, pinb[b & 1].array, rarr, items_per_frame=8)
    lm, pres, _f = fl.collect((b - 1) & 1)
dt = (time.perf_counter() - t) / nb
fl.collect(nb & 1)
print(json.dumps({"config": "FaceLandmark 512 ROIs on 64 u8 frames 540x360 from pinned HOST memory, two slots (mi_fl_submit_images / mi_fl_collect: the copy of batch n+1 beside the kernels of batch n)",
                  "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(512 / dt), "faces_present": int(pres.sum())}))
fdev = torch.from_numpy(frames64).cuda()
rdev = torch.from_numpy(np.frombuffer(bytes((mi.Rect * 512)(*rois)), np.uint8).copy()).cuda()
dt = timeit(lambda: fl.infer_images(fdev, rdev, items_per_frame=8))
print(json.dumps({"config": "FaceLandmark 512 ROIs on 64 u8 frames resident in HBM (mi_fl_infer_images)", "ms_per_batch": round(dt * 1e3, 3), "rois_per_s": round(512 / dt)}))
eye_rois = [mi.Rect(*[float(v) for v in gold["man_eye_left_roi" if k % 2 == 0 else "man_eye_right_roi"][:5]], int(gold["man_eye_left_roi"][5])) for k in range(1024)]
flips = np.array([k % 2 for k in range(1024)], np.int32)
for _ in range(3): ir.infer_images(pin.array, eye_rois, flips, items_per_frame=16)
t = time.perf_counter()
for _ in range(20): ir.infer_images(pin.array, eye_rois, flips, items_per_frame=16)
dt = (time.perf_counter() - t) / 20
print(json.dumps({"config": "IrisLandmark 1024 eye ROIs on 64 u8 frames 540x360 from pinned HOST memory (mi_iris_infer_images)", "ms_per_batch": round(dt * 1e3, 3), "eyes_per_s": round(1024 / dt)}))
