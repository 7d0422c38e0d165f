#!/bin/bash
mkdir -p gpurun_out/r06_s12
S=$(date +%s.%N)
python bench.py > gpurun_out/r06_s12/bench_default.json 2> gpurun_out/r06_s12/bench_default.err; echo "rc $?"
E=$(date +%s.%N); echo "bench.py wall: $(echo "$E - $S" | bc) s"
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06_s12/bench_default.json"))
print(d["value"], d["ms_per_step"], d["batches_in_flight"], d["value_one_batch_in_flight"], d["ms_per_step_one_batch_in_flight"], d["secondary_errors"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("rocprof_in_flight",{}).get("frac"))
for k,v in d["secondary_configs"].items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
for k,v in d["single_image_latency_us"].items(): print("  ", v.get("p50") if isinstance(v,dict) else v, k[:100])
print(d.get("host_feed",{}).get("frames_per_s"), d.get("cpu_baseline",{}).get("value"))
PY
