#!/bin/bash
set -o pipefail
O=gpurun_out/r06_s10; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_jpeg.py tests/test_gpu_parity.py tests/test_cpp_host.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -15 $O/pytest.txt
timeout -k 10 300 python tools/latency_probe.py > $O/latency.txt 2>&1; echo "latency rc $?"
grep -v amdgpu $O/latency.txt | grep -i "jpeg\|flow"
