#!/bin/bash
for i in 1 2; do
timeout -k 10 600 python tools/bench_configs.py 2>/dev/null | grep -E "BackCamera 256 u8 frames (resident|from pinned HOST memory, two)" | cut -c1-200
MI_NO_STEM_MFMA=1 timeout -k 10 600 python tools/bench_configs.py 2>/dev/null | grep -E "BackCamera 256 u8 frames (resident|from pinned HOST memory, two)" | cut -c1-200 | sed 's/^/NO_MFMA /'
done
