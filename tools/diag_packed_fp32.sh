#!/bin/bash
# Round-3 discrimination runs of the packed-FP32 finding (docs/DESIGN_HISTORY_r1-r3.md section 8).  Everything is BUILT in the development
# container (python -m os2d_amd.build --variant ..., hipcc tools/repro_packed_fp32.hip -> tools/bin/) and only RUN here.
#   1. the minimal victim (register-only complex multiply-add chain) next to MFMA aggressors: packed / scalar code,
#      shared CUs / CU-exclusive victim;
#   2. the library's own kernels as victims (transforms, split-half spectral GEMM, correlation) next to the direct 7x7 kernel,
#      in four builds: packed instructions + LDS-only barriers (the failing round-2 state), packed + full __syncthreads()
#      barriers in the transforms, packed everywhere except fft.hip, and the product build.
R=${1:-200}
OUT=gpurun_out/diag_pk
mkdir -p $OUT
{
  echo "== minimal victim"
  tools/bin/repro_pk_on $R 2000 0
  tools/bin/repro_pk_on $R 2000 1
  tools/bin/repro_pk_on $R 20000 0
  tools/bin/repro_pk_off $R 2000 0
  echo "== library kernels"
  for tag in pk_on_ldsbar pk_on_fullbar pk_fftoff; do
    OS2D_HIP_LIB=tools/diag_libs/$tag/libos2d_hip.so timeout 900 python tools/diag_aggressor.py --rounds $R
  done
  timeout 900 python tools/diag_aggressor.py --rounds $R
} 2>&1 | tee $OUT/log.txt
grep -E "RESULT|runs differ" $OUT/log.txt > $OUT/summary.txt
