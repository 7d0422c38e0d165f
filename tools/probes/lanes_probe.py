"""One batch in flight, the batch cut into frame ranges on concurrent streams INSIDE the call (engine option lanes), replayed graph and eager:
python tools/probes/lanes_probe.py  (this process never touches the GPU)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(config, opts):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(config), "--in-flight", "1", "--no-secondary", "--no-cpu-baseline", "--no-latency",
                        "--no-host-feed", "--no-event-profile"] + [x for o in opts for x in ("--opt", o)], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return "FAILED " + r.stderr[-300:].replace("\n", " | ")
    d = json.loads(lines[-1])
    return "first window %.4f ms, median %.4f" % (d["ms_per_step"], d["timing"]["ms_per_step_median"])
for config in (2, 3):
    for opts in ([], ["lanes=2"], ["lanes=2", "graph=0"], ["graph=0"], ["lanes=2", "graph=0", "pipe_band=4096"], ["lanes=3", "graph=0"]):
        print("config %d, one batch in flight, %s: %s" % (config, " ".join(opts) or "default", run(config, opts)), flush=True)
