#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4e
for v in stamps stamps_seg; do
  echo "== $v" | tee -a gpurun_out/r4e/phases.txt
  for nb in 64 1024; do
    OS2D_HIP_LIB=tools/diag_libs/$v/libos2d_hip.so timeout 300 python tools/time_dft_phases.py $nb 2>&1 | grep "^dft" | tee -a gpurun_out/r4e/phases.txt
  done
done
