#!/bin/bash
# builds tools/mpipe_bench.hip variants: tools/bin/bb_mpipe[_TAG] with extra -D flags.  usage: build_mpipe_bench.sh [TAG -DFLAG ...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift || true
mkdir -p tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Iinclude -Irs-face-detection-tflite_amd/csrc "$@" tools/mpipe_bench.hip -o tools/bin/bb_mpipe${TAG:+_$TAG}
