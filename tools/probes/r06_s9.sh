#!/bin/bash
set -o pipefail
O=gpurun_out/r06_s9; mkdir -p $O
for x in 0 1 0 1; do
  for m in back short landmark iris; do echo "== $m band_xcd=$x"; timeout -k 5 120 python tools/profile_model.py $m 1 band=2 band_xcd=$x 2>/dev/null | grep -E "bandnet|total"; done
done > $O/band_xcd.txt 2>&1
cat $O/band_xcd.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_launch or single_image or absent" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
