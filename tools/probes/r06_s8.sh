#!/bin/bash
# round 6, GPU session 8: board power under the step with one and with two batches in flight (rocm-smi beside bench.py)
set -o pipefail
O=gpurun_out/r06_s8; mkdir -p $O
rocm-smi --showmaxpower > $O/maxpower.txt 2>&1
for nf in 1 2; do
  ( for i in $(seq 1 44); do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" ; sleep 0.25; done ) > $O/smi_nf$nf.txt 2>&1 &
  SMI=$!
  sleep 1.5
  timeout -k 10 300 python bench.py --config 2 --steps 6000 --warmup 10 --no-secondary --no-cpu-baseline --no-latency --no-host-feed --no-event-profile --single-window --in-flight $nf > $O/bench_nf$nf.json 2> $O/bench_nf$nf.err; echo "bench in-flight $nf rc $?"
  wait $SMI
  python3 -c "
import json,re,sys
d=json.load(open('$O/bench_nf$nf.json')); print('in flight $nf: %.4f ms per step' % d['ms_per_step'])
w=[float(x) for x in re.findall(r'Power \(W\): ([0-9.]+)', open('$O/smi_nf$nf.txt').read())]
hi=[x for x in w if x>600]; print('  power samples under load: n %d  median %.0f W  max %.0f W' % (len(hi), sorted(hi)[len(hi)//2] if hi else 0, max(hi) if hi else 0))
"
done
cat $O/maxpower.txt | grep -i "max\|power" | head -5
