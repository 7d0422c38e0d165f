#!/bin/bash
O=gpurun_out/r06_s16; mkdir -p $O
for m in full back short landmark iris; do echo "== $m"; MI_BAND_DEBUG=1 timeout -k 5 120 python tools/profile_model.py $m 1 band=2 2>&1 | grep -E "bandnet: [0-9]|total|bandnet_kernel" | uniq; done > $O/band.txt 2>&1
cat $O/band.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.txt
timeout -k 10 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu > $O/latency.txt; grep -E "p50" $O/latency.txt | cut -c1-175
