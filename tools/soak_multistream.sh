#!/bin/bash
# Soak of the multi-stream guarantees with the PRODUCT library (docs/DESIGN_HISTORY_r1-r3.md section 8): the victim / aggressor harness for many
# rounds, and the 7-level pyramid on 7 HIP streams against the serial run in every frequency-domain mode (incl. the tiled
# 96 x 128 level), repeated.  Writes gpurun_out/soak.txt.
R=${1:-1000}
OUT=gpurun_out/soak.txt
: > $OUT
timeout 900 python tools/diag_aggressor.py --rounds $R --victims fft 2>&1 | grep RESULT | tee -a $OUT
timeout 900 python tools/diag_aggressor.py --rounds $((R / 4)) --victims gemm16,corr,sample 2>&1 | grep RESULT | tee -a $OUT
for i in 1 2 3; do
  timeout 600 python tools/diag_pyramid_determinism.py fftx3 fft32 2>&1 | grep "par" | awk '{bad = 0; for (i = 1; i <= NF; i++) if ($i ~ /e[-+][0-9]/ && $i !~ /0\.00e\+00/) bad = 1; print (bad ? "DIFF " : "same ") $0}' | cut -c1-60 | sort | uniq -c | tee -a $OUT
done
