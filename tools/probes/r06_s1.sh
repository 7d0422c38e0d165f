#!/bin/bash
# round 6, GPU session 1: one-barrier timing ablation of strip_pipe2m_kernel + the new pipe_band parity tests + a bench line
set -o pipefail
O=gpurun_out/r06_s1; mkdir -p $O
for v in base onebar nobar; do
  for sh in "256 128 24 128 1 1 4 0" "256 128 24 128 1 1 4 1" "256 64 24 64 1 1 4 0" "256 64 24 64 1 1 4 2"; do
    echo "== $v: $sh"; timeout -k 10 120 tools/bin/bb_strip_$v $sh 2>&1 | grep -E "^pipe |differ|check"
  done
done > $O/ablation.txt 2>&1
echo ablation done
for v in stamps onebar_stamps; do
  for sh in "256 128 24 128 1 1 4 0" "256 64 24 64 1 1 4 0"; do
    echo "== $v: $sh"; timeout -k 10 120 tools/bin/bb_strip_$v $sh 2>&1 | grep -E "^pipe |stamps|wave"
  done
done > $O/stamps.txt 2>&1
echo stamps done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipe_band or two_batches_in_flight" > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -3 $O/pytest.txt
timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline --no-latency --no-host-feed > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
cat $O/ablation.txt
