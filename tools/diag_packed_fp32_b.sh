#!/bin/bash
# Second pass of the packed-FP32 discrimination (see diag_packed_fp32.sh): what kind of neighbour it takes.
#   - CU-exclusive victim with an aggressor that really cannot share its CU (the aggressor holds 1 KB of LDS now), long victim
#   - fp32-MFMA and scalar-VALU (no MFMA) aggressors
#   - scalar-code victim, long
#   - the resampler as a victim in the build that keeps packed instructions everywhere except fft.hip
R=${1:-200}
OUT=gpurun_out/diag_pk
mkdir -p $OUT
{
  echo "== minimal victim, second pass"
  tools/bin/repro_pk_on $R 20000 0 0
  tools/bin/repro_pk_on $R 20000 1 0
  tools/bin/repro_pk_on $R 20000 0 1
  tools/bin/repro_pk_on $R 20000 0 2
  tools/bin/repro_pk_off $R 20000 0 0
  echo "== resampler as victim"
  OS2D_HIP_LIB=tools/diag_libs/pk_fftoff/libos2d_hip.so timeout 600 python tools/diag_aggressor.py --rounds $R --victims sample
  timeout 600 python tools/diag_aggressor.py --rounds $R --victims sample
} 2>&1 | tee $OUT/log_b.txt
grep -E "RESULT|runs differ" $OUT/log_b.txt > $OUT/summary_b.txt
