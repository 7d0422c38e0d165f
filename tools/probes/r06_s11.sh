#!/bin/bash
set -o pipefail
O=gpurun_out/r06_s11; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several_processes" -s > $O/pytest.txt 2>&1; echo "pytest rc $?"
tail -15 $O/pytest.txt
