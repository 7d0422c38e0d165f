#!/bin/bash
# builds tools/mdb_bench.hip variants: tools/bin/bb_mdb[_TAG] with extra -D flags.  usage: build_mdb_bench.sh [TAG -DFLAG ...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift || true
mkdir -p tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude -Irs-face-detection-tflite_amd/csrc "$@" tools/mdb_bench.hip -o tools/bin/bb_mdb${TAG:+_$TAG}
