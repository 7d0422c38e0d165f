#!/bin/bash
O=gpurun_out/r06_s28; mkdir -p $O
for c in 2 3 5 1; do for fl in 2 3 4; do
   python bench.py --config $c --no-secondary --no-latency --no-cpu-baseline --no-host-feed --no-event-profile --in-flight $fl > $O/b_${c}_${fl}.json 2> $O/b_${c}_${fl}.err
   python - "$O/b_${c}_${fl}.json" $c $fl <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("config",sys.argv[2],"in-flight",sys.argv[3],"ms",r["ms_per_step"],"value",r["value"])
except Exception as e: print("fail",sys.argv[1:],e)
PY
done; done
