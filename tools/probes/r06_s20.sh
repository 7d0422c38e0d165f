#!/bin/bash
O=gpurun_out/r06_s20; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "landmark or first_convolution or config3 or pipeline or synthetic_graphs_vs" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.txt
python bench.py --config 3 --no-secondary --no-latency --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06_s20/bench3.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r.get("ms_per_step_one_batch_in_flight"), r.get("roofline",{}).get("kernel"), r.get("roofline",{}).get("frac"))
PY
