#!/bin/bash
# round 6, GPU session 3: which clock does the chip hold under the pipelines' instruction mix, and under the real step?
set -o pipefail
O=gpurun_out/r06_s3; mkdir -p $O
timeout -k 10 200 tools/probes/_bin/clock_probe 20000 > $O/clock_probe.txt 2>&1; echo "probe rc $?"
cat $O/clock_probe.txt
( for i in $(seq 1 60); do date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge)" ; sleep 0.25; done ) > $O/smi_bench.txt 2>&1 &
SMI=$!
sleep 2
timeout -k 10 300 python bench.py --config 2 --steps 6000 --warmup 10 --no-secondary --no-cpu-baseline --no-latency --no-host-feed --no-event-profile --single-window --in-flight 1 > $O/bench_long.json 2> $O/bench_long.err; echo "bench rc $?"
wait $SMI
grep -E "sclk|Power" $O/smi_bench.txt | head -60
