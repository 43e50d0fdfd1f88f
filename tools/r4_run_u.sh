#!/bin/bash
# round 4, GPU call U: soak of the product build - victims on four streams next to the direct 7x7 kernel (incl. the packed
# correlation with its atomics, the register-staged GEMM, the reordered transforms), and the 7-level pyramid on 7 streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4u
OUT=gpurun_out/r4u/soak.txt
: > $OUT
timeout 600 python tools/diag_aggressor.py --rounds 400 --victims dft,gemm16 2>&1 | grep RESULT | tee -a $OUT
timeout 600 python tools/diag_aggressor.py --rounds 300 --victims corrp,corr,sample 2>&1 | grep RESULT | tee -a $OUT
for i in 1 2; do
  timeout 600 python tools/diag_pyramid_determinism.py fftx3 2>&1 | grep "par" | awk '{bad = 0; for (i = 1; i <= NF; i++) if ($i ~ /e[-+][0-9]/ && $i !~ /0\.00e\+00/) bad = 1; print (bad ? "DIFF " : "same ") $0}' | cut -c1-60 | sort | uniq -c | tee -a $OUT
done
timeout 300 python -m pytest tests/test_spectral_gpu.py -m gpu -q -p no:cacheprovider -x -k "victim or aggressor or streams" 2>&1 | tail -2
