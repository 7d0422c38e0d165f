"""The host-side lowering (plan.cpp) under AddressSanitizer, over mutated graphs.  CPU only: the file is listed in .gpurunignore (the GPU
pool refuses sanitizer builds next to an offload architecture, and the GPU run has no use for it) — it builds the HOST side of plan.cpp /
tflite_graph.cpp with hipcc's AddressSanitizer and links the product's other objects, no device code is instrumented or run."""
import os
import subprocess

import numpy as np

from conftest import MODELS, ROOT

CSRC = os.path.join(ROOT, "rs-face-detection-tflite_amd", "csrc")


def test_lowering_of_mutated_models_under_address_sanitizer(tmp_path):
    """The lowering (plan.cpp) reads shapes, axes and option fields of an untrusted graph: built with AddressSanitizer (host side of hipcc,
    linked against the product's other objects) and run over mutated blobs of three graphs.  Round 5 found a CONCATENATION whose axis lay behind
    its output's shape that way — a four-byte read behind a vector, which crashed the plain build once in twenty runs — and a PRELU on a
    tensor of rank 0."""
    build = os.path.join(ROOT, "rs-face-detection-tflite_amd", "build")
    objs = sorted(f for f in (os.path.join(build, n) for n in os.listdir(build)) if f.endswith(".o") and os.path.basename(f) not in ("plan.o", "tflite_graph.o"))
    assert len(objs) >= 15, "build the product first (__graft_entry__.build())"
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
             "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer"]
    hipcc = "/opt/rocm/bin/hipcc"
    mine = []
    for src in (os.path.join(CSRC, "plan.cpp"), os.path.join(CSRC, "tflite_graph.cpp"), os.path.join(ROOT, "tests", "asan_lowering.cpp")):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call([hipcc] + flags + ["-x", "hip", "-c", src, "-o", o], stderr=subprocess.DEVNULL)
        mine.append(o)
    exe = str(tmp_path / "asan_lowering")
    subprocess.check_call([hipcc, "-fsanitize=address", "--offload-arch=gfx950", "-o", exe] + mine + objs, stderr=subprocess.DEVNULL)
    blobs = []
    # (seed 7 = the mutations of test_mutated_models_never_crash_the_lowering: blob 81 of the BackCamera graph has the CONCATENATION axis; the iris
    # graph with seed 5 holds a PRELU on a tensor of rank 0 — shape.back() of an empty vector)
    for name, seed in (("face_detection_back.tflite", 7), ("face_detection_full_range_sparse.tflite", 7), ("iris_landmark.tflite", 7), ("iris_landmark.tflite", 5)):
        rs = np.random.RandomState(seed)
        orig = np.frombuffer(open(os.path.join(MODELS, name), "rb").read(), np.uint8)
        for r in range(150):
            b = orig.copy()
            lo = 0 if r % 3 == 0 else max(0, len(b) - 60000)
            for _ in range(rs.randint(1, 6)):
                b[rs.randint(lo, len(b))] = rs.randint(0, 256)
            f = str(tmp_path / ("%s_%d_%03d.bin" % (name, seed, r)))
            open(f, "wb").write(b.tobytes())
            blobs.append(f)
    r = subprocess.run([exe] + blobs, capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stderr[-3000:]
    ok, refused = (int(v) for v in r.stdout.split()[1::2])
    assert ok > 100 and refused > 100
