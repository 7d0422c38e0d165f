"""Timeline of ONE single-image call on the device: run under rocprofv3 (kernel trace + memory-copy trace), then summarise.
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 tools/single_call_timeline.py run back
  python3 tools/single_call_timeline.py show gpurun_out/tl
`run` makes 40 calls of the entry (FaceDetection::infer on man.jpg, host Mat in, results out) and prints the host-side p50; `show`
prints, for the LAST call, every copy and kernel with its start and end relative to the call's first device activity."""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind):
    import numpy as np
    from PIL import Image
    import rs_face_detection_tflite_amd as mi
    img = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "man.jpg")).convert("RGB"))
    gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    if kind == "landmark":
        h = mi.FaceLandmark()
        roi = mi.Rect(*[float(v) for v in gold["man_face_roi"][:5]], int(gold["man_face_roi"][5]))
        fn = lambda: h.infer(img, roi)
    elif kind == "iris":
        h = mi.IrisLandmark()
        roi = mi.Rect(*[float(v) for v in gold["man_eye_left_roi"][:5]], int(gold["man_eye_left_roi"][5]))
        fn = lambda: h.infer(img, roi, False)
    elif kind == "pipe":   # the whole flow of lib.rs:24-40 on one picture (detector -> mesh -> both eyes), one call
        h = mi.Pipeline(mi.FaceDetectionModel.BackCamera)
        one = np.ascontiguousarray(img[None])
        fn = lambda: h.run(one)
    else:
        h = mi.FaceDetection({"back": mi.FaceDetectionModel.BackCamera, "short": mi.FaceDetectionModel.Short, "full": mi.FaceDetectionModel.Full}[kind])
        fn = lambda: h.infer(img, None)
    ts = []
    for i in range(40):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
        time.sleep(0.002)   # a gap between calls: the last call is easy to find in the trace
    ts = sorted(ts[10:])
    print("host p50 %.1f us" % (ts[len(ts) // 2] * 1e6))
    h.close()


def show(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s %s B" % (r.get("Direction", "?"), r.get("Size", r.get("Bytes", "?")))))
    ev.sort()
    # the last call = the events behind the last gap of more than 1 ms
    k = len(ev) - 1
    while k > 0 and ev[k][0] - ev[k - 1][1] < 1000000:
        k -= 1
    call = ev[k:]
    t0 = call[0][0]
    busy = 0
    for s, e, n in call:
        print("%8.1f %8.1f  %6.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
        busy += e - s
    print("first start to last end %.1f us, sum of durations %.1f us, %d events" % ((call[-1][1] - t0) / 1e3, busy / 1e3, len(call)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        show(sys.argv[2])
