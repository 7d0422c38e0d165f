#!/bin/bash
# round 6, GPU session 2: NOHAND ablations, 2/3-block pipelines, band fixes
set -o pipefail
O=gpurun_out/r06_s2; mkdir -p $O
for v in nohand nohand_onebar; do
  for sh in "256 128 24 128 1 1 4 0" "256 64 24 64 1 1 4 0"; do
    echo "== $v: $sh"; timeout -k 10 120 tools/bin/bb_strip_$v $sh 2>&1 | grep -E "^pipe  "
  done
done > $O/ablation.txt 2>&1
for sh in "256 128 24 128 1 1 2 0" "256 128 24 128 1 1 3 0" "256 128 24 128 1 1 4 0" "256 128 24 128 1 1 2 1" "256 128 24 128 1 1 3 1" "256 64 24 64 1 1 2 0" "256 64 24 64 1 1 3 0" "256 64 24 64 1 1 2 2"; do
  echo "== base: $sh"; timeout -k 10 120 tools/bin/bb_strip_base $sh 2>&1 | grep -E "^pipe  |check"
done >> $O/ablation.txt 2>&1
echo ablation done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_launch or single_image or dist_broadcast" > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -15 $O/pytest.txt
cat $O/ablation.txt
