#!/bin/bash
# round 4, GPU call K2: the GEMM with the spectra conversion interleaved into the matrix loop against the sequential form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4k
OUT=gpurun_out/r4k/spectral_gemm_interleave.txt
: > $OUT
timeout 600 python -m pytest tests/test_spectral_gpu.py -m gpu -q -p no:cacheprovider -x -k "split_half or quad" 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
OS2D_HIP_LIB=tools/diag_libs/sh_il0/libos2d_hip.so timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep "TIME\|rror" | tee -a $OUT
done
