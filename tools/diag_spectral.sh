#!/bin/bash
# Where does the split-half spectral GEMM spend its launch, and what do its variants cost?  Diagnostic builds (made in the
# development container: python -m os2d_amd.build --variant <tag> -D...), timed at 64 and 1024 pairs:
#   component removal (round 3, first pass):  sh_nox / sh_now / sh_noy / sh_nomfma / sh_nosplit / sh_nox_now  (-DOS2D_DIAG_SH_*)
#   work decomposition:  product = 8-wave work-group, 32-byte stores;  sh_wg4 = 4-wave groups (-DOS2D_SH_WG8=0);
#                        sh_wave0 = round 2 (-DOS2D_SH_WG8=0 -DOS2D_SH_WAVE_BINS=0)
OUT=gpurun_out/diag_spectral.txt
: > $OUT
timeout 300 python tools/time_spectral16.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
for tag in sh_wg4 sh_wave0 sh_nox sh_now sh_noy sh_nomfma sh_nosplit sh_nox_now; do
  [ -f tools/diag_libs/$tag/libos2d_hip.so ] || continue
  OS2D_HIP_LIB=tools/diag_libs/$tag/libos2d_hip.so timeout 300 python tools/time_spectral16.py 64 256 1024 2>&1 | grep TIME | tee -a $OUT
done
