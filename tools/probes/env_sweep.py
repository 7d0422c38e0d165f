"""bench.py (config 2, two batches in flight) under tuning environment variables, one child run per setting (this process never touches the GPU).
python tools/probes/env_sweep.py MI_PIPE_BAND=16,32,64,128 MI_PIPE_PERCU=1,2 ...   -> one line per setting: first window / median ms per step"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
config = os.environ.get("SWEEP_CONFIG", "2")
def run(env_extra):
    # KEY=value in capitals: an environment variable; opt:key=value: an engine option (bench.py --opt)
    opts = [k[4:] + "=" + v for k, v in env_extra.items() if k.startswith("opt:")]
    env = dict(os.environ, **{k: v for k, v in env_extra.items() if not k.startswith("opt:")})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--no-secondary", "--no-cpu-baseline", "--no-latency", "--no-host-feed",
                        "--no-event-profile"] + [x for o in opts for x in ("--opt", o)], capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return "FAILED " + r.stderr[-200:].replace("\n", " | ")
    d = json.loads(lines[-1]); t = d["timing"]
    return "first window %.4f ms, median %.4f, one in flight %.4f" % (d["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_one_batch_in_flight"])
print("default:", run({}), flush=True)
for arg in sys.argv[1:]:
    k, vs = arg.split("=")
    for v in vs.split(","):
        print("%s=%s: %s" % (k, v, run({k: v})), flush=True)
