#!/bin/bash
# Builds libmiface.so (C ABI in include/mi_face.h) for gfx950 with hipcc. Cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
SRC="$HERE/csrc"
# Development variants (in-kernel stamps ...): MI_VARIANT=name MI_EXTRA_FLAGS="-D..." -> build-name/, libmiface-name.so
VARIANT="${MI_VARIANT:-}"
OUT="$HERE/libmiface${VARIANT:+-$VARIANT}.so"
BUILD="$HERE/build${VARIANT:+-$VARIANT}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result -I"$HERE/../include" ${MI_EXTRA_FLAGS:-})
OBJ=()
mkdir -p "$BUILD"
for f in tflite_graph.cpp plan.cpp host_glue.cpp jpeg.cpp engine.cpp capi.cpp jpeg_kernels.hip kernels.hip block_kernels.hip strip_kernels.hip chain_kernels.hip resident_kernels.hip bneck_kernels.hip dblock_kernels.hip xc_kernels.hip mstrip_kernels.hip mdblock_kernels.hip mwalk_kernels.hip ms2_kernels.hip tail_kernels.hip bandnet_kernels.hip preproc.hip; do
  o="$BUILD/${f%.*}.o"
  if [[ ! -f "$o" || "$SRC/$f" -nt "$o" || -n "$(find "$SRC" -name '*.hpp' -newer "$o" -print -quit)" || "$HERE/../include/mi_face.h" -nt "$o" ]]; then
    echo "  hipcc $f"
    case "$f" in
      # mstrip / mdblock: scalar FMAs on purpose (a tap is one register for two pixel tiles); the SLP vectoriser would pack them and pay in moves
      mstrip_kernels.hip|mdblock_kernels.hip|mwalk_kernels.hip|ms2_kernels.hip) "$HIPCC" "${FLAGS[@]}" -fno-slp-vectorize -c "$SRC/$f" -o "$o" ;;
      *.hip) "$HIPCC" "${FLAGS[@]}" -c "$SRC/$f" -o "$o" ;;
      *)     "$HIPCC" "${FLAGS[@]}" -x hip -c "$SRC/$f" -o "$o" ;;
    esac
  fi
  OBJ+=("$o")
done
"$HIPCC" -shared -fPIC --offload-arch=gfx950 -o "$OUT" "${OBJ[@]}"
echo "built $OUT"
