import json,sys
r=json.loads(sys.stdin.read())
k=r["roofline"]["kernels"]
print(r["value"], r["ms_per_step"])
for n,v in sorted(k.items(), key=lambda kv:-kv[1]["ms"]): print("  %-34s %8.4f ms %3d calls" % (n, v["ms"], v["calls"]))
