#!/bin/bash
# round 4, GPU call L: phase stamps of the transforms after the vectorised splits / blocked spectra; GEMM timing with the vectorised split
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4l
OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so timeout 300 python tools/time_dft_phases.py > gpurun_out/r4l/dft_phases_v3.txt 2>&1
cat gpurun_out/r4l/dft_phases_v3.txt | tail -8
timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep TIME | tee gpurun_out/r4l/gemm.txt
timeout 600 python -m pytest tests/test_spectral_gpu.py tests/test_dft_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
