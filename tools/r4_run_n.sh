#!/bin/bash
# round 4, GPU call N: the GEMM with register-staged weights (exact wait counts) against the LDS-DMA form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4n
OUT=gpurun_out/r4n/spectral_gemm_regw.txt
: > $OUT
timeout 600 python -m pytest tests/test_spectral_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
OS2D_HIP_LIB=tools/diag_libs/sh_dma/libos2d_hip.so timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep "TIME\|rror" | tee -a $OUT
done
timeout 900 python -m pytest tests/test_head_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
