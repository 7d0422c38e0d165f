#!/bin/bash
O=gpurun_out/r06_s21; mkdir -p $O
for c in 3 2; do
 for ch in 0 64 128 256; do
  for fl in 1 2; do
   python bench.py --config $c --no-secondary --no-latency --no-cpu-baseline --no-host-feed --no-event-profile --in-flight $fl --chunk $ch > $O/b_${c}_${ch}_${fl}.json 2> $O/b_${c}_${ch}_${fl}.err
   python - "$O/b_${c}_${ch}_${fl}.json" $c $ch $fl <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("config",sys.argv[2],"chunk",sys.argv[3],"in-flight",sys.argv[4],"ms",r["ms_per_step"],"value",r["value"])
except Exception as e: print("fail",sys.argv[1:],e)
PY
  done
 done
done
