"""Dense batch-size sweep (development aid, GPU): which kernel runs a layer — and how it cuts the batch into bands, groups and workgroups —
depends on the batch, so every batch size from 1 to N is compared with the LDS-tensor kernels ("strip" = 0, "tail" = 0) on the same
frames (frames are independent: reference[:nb]).  Found round 5's ms2_kernel band-parity bug at 37 / 70 frames.
usage: python tools/batch_sweep.py [max_batch=200] [model ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rs_face_detection_tflite_amd as mi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = {"back": "face_detection_back.tflite", "front": "face_detection_front.tflite", "full": "face_detection_full_range.tflite",
         "sparse": "face_detection_full_range_sparse.tflite", "landmark": "face_landmark.tflite", "iris": "iris_landmark.tflite"}
nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 200
models = sys.argv[2:] or list(FILES)
extra = [255, 256, 257, 300, 383, 384, 385, 511, 512, 513, 600]
bad = 0
for name in models:
    m = mi.Model(os.path.join(ROOT, "models", FILES[name]))
    h, w = m.input_dims[1], m.input_dims[2]
    nb_all = sorted(set(list(range(1, nmax + 1)) + [e for e in extra if name in ("landmark", "iris", "front") or e <= 300]))
    N = max(nb_all)
    rs = np.random.RandomState(5)
    lo = -1.0 if name in ("back", "front", "full", "sparse") else 0.0
    x = torch.from_numpy(rs.uniform(lo, 1, (N, h, w, 3)).astype(np.float32)).cuda()
    m.set_option("strip", 0); m.set_option("tail", 0); m.set_option("mchain", 0)
    ref = [o.clone() for o in m.run(x)]
    m.set_option("strip", 1); m.set_option("tail", 1); m.set_option("mchain", 1)
    worst = 0.0
    for nb in nb_all:
        outs = m.run(x[:nb])
        for o, r in zip(outs, ref):
            r = r[:nb]
            scale = max(1.0, float(r.abs().max()))
            err = float((o - r).abs().max()) / scale
            worst = max(worst, err)
            if not err <= 1e-4:
                bad += 1
                print("MISMATCH %s batch %d: %.3e (relative to max|x| %.1f)" % (name, nb, err, scale))
                break
    print("%-9s batches 1..%d (+%s): worst relative difference %.2e" % (name, nmax, [e for e in nb_all if e > nmax], worst))
    m.close()
print("sweep done, mismatching (model, batch) pairs:", bad)
sys.exit(1 if bad else 0)
