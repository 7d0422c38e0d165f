#!/bin/bash
O=gpurun_out/r06_s22; mkdir -p $O
for b in 0 12 16 24 32 48 96; do echo "mdb_band $b"; timeout -k 5 120 python tools/profile_model.py landmark 512 mdb_band=$b 2>/dev/null | grep -E "mdblock|total"; done
