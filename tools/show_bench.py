import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], r["unit"], "ms/step", r["ms_per_step"], "| dominant", r["roofline"]["kernel"], "frac", r["roofline"]["frac"], "net_event_ms", r["roofline"]["net_event_ms"])
for k, v in sorted(r["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
    print("   %-28s %s" % (k, v))
if "cpu_baseline" in r: print("cpu", r["cpu_baseline"])
