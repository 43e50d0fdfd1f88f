#!/bin/bash
# round 4, GPU call K: where the split-half GEMM (quads layout, blocks of 64 pairs) spends its launch: component-removal builds
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4k
OUT=gpurun_out/r4k/spectral_gemm_components.txt
: > $OUT
timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
for tag in sh_wring4 sh_nox sh_now sh_noy sh_nomfma sh_nosplit sh_nox_now; do
  [ -f tools/diag_libs/$tag/libos2d_hip.so ] || continue
  OS2D_HIP_LIB=tools/diag_libs/$tag/libos2d_hip.so timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep "TIME\|Error\|error" | tee -a $OUT
done
timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
