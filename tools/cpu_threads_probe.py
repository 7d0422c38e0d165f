"""How the CPU baseline (C oracle, OpenMP over frames) scales with threads on this host: python tools/cpu_threads_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import pyoracle as po

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", bench.usable_cpus())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
om = po.Model(os.path.join(bench.ROOT, "models", "face_detection_back.tflite"))
x = bench.make_frames(256, 0)
for t in (16, 32, 64, 128, 256):
    if t > 2 * (os.cpu_count() or 1):
        break
    om.run(x[:t], nthreads=t)
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        om.run(x, nthreads=t); n += 256
    print("threads %3d: %.1f frames/s" % (t, n / (time.time() - t0)), flush=True)
