#!/bin/bash
set -o pipefail
O=gpurun_out/r06_s14; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_launch or single_image or absent or detector or pipeline or several" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.txt
timeout -k 10 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu > $O/latency.txt; grep -E "Full|BackCamera \(man|Short|FaceLandmark|IrisLand|flow on one" $O/latency.txt
