#!/bin/bash
set -o pipefail
O=gpurun_out/r06_s13; mkdir -p $O
for m in back short landmark iris full; do echo "== $m"; MI_BAND_DEBUG=1 timeout -k 5 120 python tools/profile_model.py $m 1 band=2 2>&1 | grep -E "bandnet|total|no single"; done > $O/band.txt 2>&1
cat $O/band.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_launch or single_image or absent" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.txt
