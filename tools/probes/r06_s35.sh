#!/bin/bash
# round 6: in-kernel stamps of config 5's two leading detector launches on the final tree (stamps builds: MI_VARIANT=xstamps MI_EXTRA_FLAGS=-DMI_XC_STAMPS and
# MI_VARIANT=dstamps MI_EXTRA_FLAGS=-DMI_DBLOCK_STAMPS bash rs-face-detection-tflite_amd/build.sh)
O=gpurun_out/r06_s35; mkdir -p $O
MI_XC_STAMPS=1 timeout -k 10 300 python tools/xc_stamps.py > $O/xc.txt 2>&1; echo "xc rc $?"; grep -v amdgpu $O/xc.txt | tail -14
for h in 24 48 96; do MI_DB_H=$h timeout -k 10 300 python tools/dblock_stamps.py 128 > $O/db_$h.txt 2>&1; echo "dblock H=$h rc $?"; grep -v amdgpu $O/db_$h.txt | tail -8; done
