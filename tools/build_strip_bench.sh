#!/bin/bash
# builds tools/strip_bench.hip variants: tools/bin/bb_strip[_TAG] with extra -D flags.  usage: build_strip_bench.sh TAG [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
TAG=$1; shift || true
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -Irs-face-detection-tflite_amd/csrc "$@" tools/strip_bench.hip -o tools/bin/bb_strip${TAG:+_$TAG}
