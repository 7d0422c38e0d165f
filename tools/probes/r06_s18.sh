#!/bin/bash
O=gpurun_out/r06_s18; mkdir -p $O
MI_BAND_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "synthetic_graphs and sparse" > $O/synth.txt 2>&1; echo "rc $?"; grep -E "stages,|passed|failed|Error" $O/synth.txt | uniq | tail
timeout -k 10 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu > $O/latency.txt; grep -E "Sparse" $O/latency.txt | cut -c1-175
