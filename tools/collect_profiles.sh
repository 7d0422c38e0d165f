#!/bin/bash
# Runs on the GPU box (via gpurun): the BASELINE bench line, the secondary configs, and the rocprofv3 passes the committed
# summaries under profiles/ come from (kernel trace + stats; FETCH_SIZE and WRITE_SIZE in separate PMC passes).
# usage: bash tools/collect_profiles.sh   -> everything under gpurun_out/prof_<tag>/ ; then tools/summarize_profiles.py
set -eo pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -1 "$OUT/bench.json" | cut -c1-200
python3 tools/bench_configs.py > "$OUT/configs.log" 2>&1
grep config "$OUT/configs.log"
for m in "back 256" "front 256" "full 128" "landmark 512" "iris 1024"; do set -- $m; python3 tools/profile_model.py $1 $2 2>/dev/null | grep -v amdgpu > "$OUT/launches_$1.txt"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 30 --no-cpu-baseline > "$OUT/trace.log" 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/write.log" 2>&1
echo write done
find "$OUT" -name "*.csv" | head -20
