"""bench.py at 1 - 3 batches in flight for the configs given (this process never touches the GPU; the runs are children, one after the other).
python tools/probes/in_flight_sweep.py 2 3 5"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for c in [int(v) for v in sys.argv[1:]] or [2]:
    for n in (2, 3):
        for env_extra in ({}, {"GPU_MAX_HW_QUEUES": "8"}) if n == 3 else ({},):
            env = dict(os.environ, **env_extra)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(c), "--in-flight", str(n), "--no-secondary", "--no-cpu-baseline",
                                "--no-latency", "--no-host-feed", "--no-event-profile"], capture_output=True, text=True, env=env, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not lines:
                print("config", c, "in flight", n, "FAILED", r.stderr[-300:])
                continue
            d = json.loads(lines[-1]); t = d["timing"]
            print("config %d, %d in flight%s: first window %.4f ms, median %.4f, one in flight %.4f" % (
                c, n, " (GPU_MAX_HW_QUEUES=8)" if env_extra else "", d["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_one_batch_in_flight"]), flush=True)
