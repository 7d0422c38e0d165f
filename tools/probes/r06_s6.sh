#!/bin/bash
# round 6, GPU session 6: streamed JPEG entries (test + latency), band fixes test
set -o pipefail
O=gpurun_out/r06_s6; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_jpeg.py tests/test_gpu_parity.py -x -q -m gpu -k "jpeg or streamed or absent" > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -25 $O/pytest.txt
timeout -k 10 300 python tools/latency_probe.py > $O/latency.txt 2>&1; echo "latency rc $?"
cat $O/latency.txt | grep -v amdgpu
