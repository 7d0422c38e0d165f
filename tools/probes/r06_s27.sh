#!/bin/bash
O=gpurun_out/r06_s27; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.txt
for c in 2 1; do python bench.py --config $c --no-secondary --no-latency --no-cpu-baseline --no-host-feed > $O/bench$c.json 2> $O/bench$c.err; python - $O/bench$c.json <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["config"]["workload"][:30], r["value"], r["ms_per_step"], r.get("ms_per_step_one_batch_in_flight"), r.get("roofline",{}).get("kernel"), r.get("roofline",{}).get("frac"))
PY
done
