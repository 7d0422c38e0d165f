#!/bin/bash
O=gpurun_out/r06_s31; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "infer_images or submit or u8 or host or mfma_stem" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.txt
for i in 1 2; do
timeout -k 10 600 python tools/bench_configs.py 2>/dev/null | grep -E "BackCamera 256 u8 frames (resident|from pinned HOST memory, two)" | cut -c1-200
MI_NO_STEM_MFMA=1 timeout -k 10 600 python tools/bench_configs.py 2>/dev/null | grep -E "BackCamera 256 u8 frames (resident|from pinned HOST memory, two)" | cut -c1-200 | sed 's/^/NO_MFMA /'
done
