#!/bin/bash
# round 6, GPU session 5: the timing ablations of the row pipelines INSIDE the real step (bench.py --config 2, one batch in flight): in the harness loop
# the chip is power-managed down to 1.95 GHz, in the plan the pipelines run at ~2.37 GHz (r06_s4: GRBM_GUI_ACTIVE)
set -o pipefail
O=gpurun_out/r06_s5; mkdir -p $O
for v in "" -onebar -nobar -nohand_onebar ""; do
  MIFACE_LIB=$PWD/rs-face-detection-tflite_amd/libmiface$v.so timeout -k 10 300 python bench.py --config 2 --steps 200 --warmup 10 --no-secondary --no-cpu-baseline --no-latency --no-host-feed --in-flight 1 > $O/bench$v.json 2> $O/bench$v.err; echo "bench '$v' rc $?"
  python3 - "$O/bench$v.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("variant '%s': %.4f ms/step (median %s)" % (sys.argv[2], d["ms_per_step"], d["timing"].get("ms_per_step_median")))
for r in d["roofline"].get("by_shape", []):
    print("   %-36s %-28s %.4f ms" % (r.get("kernel"), r.get("shape"), r.get("ms", 0)))
PY
done 2>&1 | tee $O/summary.txt
