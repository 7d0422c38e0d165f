#!/bin/bash
O=gpurun_out/r06_s30; mkdir -p $O
S=$(date +%s); python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $? in $(( $(date +%s) - S )) s"
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06_s30/bench_default.json").read().strip().splitlines()[-1])
print(r["metric"], r["value"], r["ms_per_step"], r["ms_per_step_one_batch_in_flight"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["traffic"], r["cpu_baseline"]["value"])
print({k:(v.get("value"), v.get("ms_per_step"), v.get("ms_per_step_one_batch_in_flight")) for k,v in r.get("secondary_configs",{}).items()})
print(r.get("secondary_errors"))
print({k:v["p50"] for k,v in r["single_image_latency_us"].items()})
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
