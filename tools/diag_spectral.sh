#!/bin/bash
# Where does the split-half spectral GEMM spend its launch?  Diagnostic builds with one component removed each
# (built in the development container: python -m os2d_amd.build --variant sh_<x> -DOS2D_DIAG_SH_<X>), timed at 64 and 1024 pairs.
OUT=gpurun_out/diag_spectral.txt
: > $OUT
timeout 300 python tools/time_spectral16.py 64 1024 2>&1 | grep TIME | tee -a $OUT
for tag in sh_nox sh_now sh_noy sh_nomfma sh_nosplit sh_nox_now; do
  [ -f tools/diag_libs/$tag/libos2d_hip.so ] || continue
  OS2D_HIP_LIB=tools/diag_libs/$tag/libos2d_hip.so timeout 300 python tools/time_spectral16.py 64 1024 2>&1 | grep TIME | tee -a $OUT
done
