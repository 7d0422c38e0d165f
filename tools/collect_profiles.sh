#!/bin/bash
# Runs on the GPU box (via gpurun): the bench lines of configs 1 / 2 / 3 / 5, the per-launch event lists, and the rocprofv3 passes
# the committed summaries under profiles/ come from — kernel trace + stats per config; FETCH_SIZE and WRITE_SIZE in separate PMC
# passes per config; SQ counters in passes of at most 8 per config (never --pmc together with a trace).  rocprofv3 is given the
# program itself (python3 bench.py ...), never a wrapper.
# The summaries are made ON the box (gpurun_out/prof_<tag>/summary) and the bench lines are taken AFTER them with MI_PMC_SUMMARY pointing at
# that summary, so a committed bench line and the PMC figures beside it always come from one collection.
# usage: bash tools/collect_profiles.sh r04   -> everything under gpurun_out/prof_r04/ ; then cp gpurun_out/prof_r04/summary/* profiles/
set -eo pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
python3 tools/bench_configs.py > "$OUT/configs.log" 2>&1; echo configs done
for m in "back 256" "front 256" "full 128" "landmark 512" "iris 1024"; do set -- $m; python3 tools/profile_model.py $1 $2 2>/dev/null | grep -v amdgpu > "$OUT/launches_$1.txt"; done; echo launches done
CONFIGS=${CONFIGS:-"2 1 3 5"}   # (CONFIGS="3" for a quick partial collection)
for c in $CONFIGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c$c" -- python3 bench.py --config $c --warmup 10 --steps 100 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --no-event-profile --single-window --in-flight 1 > "$OUT/trace_c$c.log" 2>&1; echo trace $c done
  # the mode bench.py times by default (VERDICT r5 missing #3): two batches in flight, whole-frame bands for the row pipelines
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2_c$c" -- python3 bench.py --config $c --warmup 10 --steps 100 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --no-event-profile --single-window --in-flight 2 > "$OUT/trace2_c$c.log" 2>&1; echo trace in-flight-2 $c done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_c$c" -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --single-window --in-flight 1 > "$OUT/fetch_c$c.log" 2>&1; echo fetch $c done
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_c$c" -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --single-window --in-flight 1 > "$OUT/write_c$c.log" 2>&1; echo write $c done
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d "$OUT/sq${i}_c$c" -- python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-host-feed --no-secondary --single-window --in-flight 1 > "$OUT/sq${i}_c$c.log" 2>&1; echo sq pass $i config $c done
  done
done
python3 tools/summarize_profiles.py "$TAG" "$OUT/summary" > "$OUT/summarize.log" 2>&1 || { tail -5 "$OUT/summarize.log"; exit 1; }
for c in $CONFIGS; do
  MI_PMC_SUMMARY="$OUT/summary/pmc_summary.json" python3 bench.py --config $c --no-secondary > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"; echo bench $c done
done
python3 tools/summarize_profiles.py "$TAG" "$OUT/summary" > "$OUT/summarize.log" 2>&1   # again: picks the bench lines up
# ---- the single-image calls (the reference's operating point): per-launch lists at one frame, the latency probe, and a kernel trace of it
{
  echo "# per-launch HIP-event times at a batch of ONE frame (tools/profile_model.py <model> 1): the batched plan, and for the graphs that have one"
  echo "# the single-launch plan the single-image entries take (option band=2 makes every small run take it)"
  for m in back front short full sparse landmark iris; do echo "== $m 1 (batched plan)"; python3 tools/profile_model.py $m 1 band=0 2>/dev/null | grep -v amdgpu; done
  for m in back front short full sparse landmark iris; do echo "== $m 1 (single-launch plan)"; python3 tools/profile_model.py $m 1 band=2 2>/dev/null | grep -v amdgpu; done
  echo; echo "# tools/latency_probe.py: per call, host Mat in, results out (us); then the same calls timed at the C ABI"
  python3 tools/latency_probe.py 2>/dev/null | grep -v amdgpu
} > "$OUT/summary/${TAG}_launches_batch1.txt"; echo batch-1 lists done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_single" -- python3 tools/latency_probe.py > "$OUT/trace_single.log" 2>&1; echo trace single done
f=$(find "$OUT/trace_single" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/summary/${TAG}_kernel_stats_single_image.csv"
# keep what gpurun merges back small: counter CSVs, stats and the summaries only
find "$OUT" -name "*.db" -delete 2>/dev/null || true
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null || true
du -sh "$OUT"
