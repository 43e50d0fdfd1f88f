#!/bin/bash
# Device assembly of the product translation units (gfx950), one .s per unit: tools/dump_asm.sh <outdir> [unit ...]
# Used to check that a source clean-up leaves the generated code untouched (diff of two dumps) and to read wait counts.
OUT=$1; shift
CSRC=$(dirname "$0")/../os2d_amd/csrc
mkdir -p "$OUT"
UNITS=${@:-$(cd $CSRC && ls *.hip | sed 's/\.hip$//')}
for u in $UNITS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Xclang -target-feature -Xclang -packed-fp32-ops $OS2D_EXTRA_HIPCC_FLAGS \
        -S --offload-device-only -o "$OUT/$u.s" "$CSRC/$u.hip" 2>/dev/null &
done
wait
