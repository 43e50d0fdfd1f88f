#!/bin/bash
# One GPU-box call (through gpurun), made of named steps; everything lands under gpurun_out/<tag>/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> <step> [<step> ...]'
# steps:  tests            the whole GPU suite (pytest -m gpu)
#         tests:<expr>     pytest -m gpu -k <expr>
#         bench            the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5): line + details + stderr
#         stages[:<variant>]  stage times of the 64- and the 1024-class step (product library or a variant)
#         bench1024        python bench.py --classes 1024 (stage times of the 1024-class step only)
#         prof             rocprofv3 --kernel-trace --stats + PMC passes of the 64-class step (tools/profile_bench.sh)
#         prof1024         the same at 1024 classes
#         phases[:<variant>]  phase stamps of the transforms (needs tools/diag_libs/<variant>, default "stamps": build --variant stamps
#                          -DOS2D_DIAG_DFT_STAMPS [other -D flags])
#                          (the raw rocprofv3 databases are deleted after summarising: gpurun merges at most 64 MiB back)
#         gemm[:<variant>] tools/time_spectral16_quads.py 64 256 1024 (the per-bin GEMM alone) with the product library or a variant
#         power            tools/power_probe.sh: package power / clocks while the correlation, the step and the register-only MFMA loop run
#         env:VAR=VALUE / unset:VAR   export / unset an environment variable for the steps that follow
#         ab:VAR=VALUE[,VAR=VALUE]    the "stages" step under these variables (one A/B leg; runs in a sub-shell)
#         bin:<name>       tools/bin/<name> (a standalone HIP program built on the dev box, e.g. split_mix_check)
#         soak[:<rounds>]  tools/soak_multistream.sh: victim / aggressor rounds + the pyramid on 7 streams against the serial run (default 300)
#         mfma             tools/bin/mfma_peak: what v_mfma_f32_32x32x16_f16 sustains (register-only loop, zero / random operands)
#         counters:<classes>[:<kernel>]  FETCH x 2 / WRITE bytes and launch times of the step's kernels (tools/ab_counters.py; inside ab:...:counters:N for a variant)
#         smoke            __graft_entry__.smoke()
#         py:<script> ...  python <script> (rest of the arguments up to the next known step are NOT consumed: one script, no args)
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT
for STEP in "$@"; do
  echo "=== $STEP"
  case "$STEP" in
    tests)
      ( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log;;
    tests:*)
      ( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "${STEP#tests:}" ) > $OUT/pytest_k.log 2>&1; echo "pytest -k rc=$?" | tee -a $OUT/pytest_k.log; tail -15 $OUT/pytest_k.log;;
    bench)
      ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? line bytes=$(wc -c < $OUT/bench.json)"
      cp -f gpurun_out/bench_details.json $OUT/bench_details.json 2>/dev/null; cat $OUT/bench.json;;
    levels|levels:*)
      V=""; [ "$STEP" != levels ] && V=${STEP#levels:}
      if [ -n "$V" ]; then export OS2D_HIP_LIB=tools/diag_libs/$V/libos2d_hip.so; fi
      timeout 300 python tools/time_pyramid_levels.py 2>&1 | grep -v "amdgpu.ids\|^{" | sed "s/^/[${V:-product}] /" | tee -a $OUT/levels.txt | tail -11; unset OS2D_HIP_LIB;;
    stages|stages:*)
      V=""; [ "$STEP" != stages ] && V=${STEP#stages:}
      if [ -n "$V" ]; then export OS2D_HIP_LIB=tools/diag_libs/$V/libos2d_hip.so; fi
      for N in 64 1024; do timeout 300 python bench.py --classes $N --steps 10 --warmup 3 --no-cpu-baseline --no-other-precision --no-end-to-end --no-sweep --no-live-counters 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[${V:-product}${LABEL:+ $LABEL}] classes $N', d['ms_per_step'], d['stages_ms'])" | tee -a $OUT/stages.txt; done; unset OS2D_HIP_LIB;;
    bench1024)
      ( timeout 600 python bench.py --classes 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-other-precision --no-end-to-end --no-sweep --no-live-counters ) > $OUT/bench1024.json 2> $OUT/bench1024.err; echo "rc=$?"; cat $OUT/bench1024.json;;
    trace|trace:*)      # trace[:classes]: ONE rocprofv3 --kernel-trace --stats pass over the short bench of the step (kernel table + timeline of the last two steps)
      N=64; [ "$STEP" != trace ] && N=${STEP#trace:}
      D=$GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_$N${LABEL:+_x}; rm -rf $D; mkdir -p $D
      ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $D/stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --classes $N --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision --no-end-to-end --no-sweep --no-live-counters ) > $OUT/trace_$N.log 2>&1
      python tools/summarize_prof.py $D 2>&1 | grep -v "at6native\|rocclr\|vectorized\|reduce_kernel\|elementwise" | cut -c1-200 | sed "s|^|[${LABEL:-product} $N] |" | tee -a $OUT/trace_$N.txt | grep "_kernel\|span" | head -40; rm -rf $D;;
    prof)
      bash tools/profile_bench.sh ${TAG}_fftx3 --precision fftx3 > $OUT/prof.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_${TAG}_fftx3 > $OUT/rocprof_summary.txt 2>&1; rm -rf gpurun_out/prof_${TAG}_fftx3; grep -v "at6native\|rocclr" $OUT/rocprof_summary.txt | cut -c1-260 | head -60;;
    prof1024)
      bash tools/profile_bench.sh ${TAG}_fftx3_1024 --precision fftx3 --classes 1024 > $OUT/prof1024.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_${TAG}_fftx3_1024 > $OUT/rocprof_summary_1024.txt 2>&1; rm -rf gpurun_out/prof_${TAG}_fftx3_1024; grep -v "at6native\|rocclr" $OUT/rocprof_summary_1024.txt | cut -c1-260 | head -60;;
    phases|phases:*)
      V=stamps; [ "$STEP" != phases ] && V=${STEP#phases:}
      for ARGS in "64 60 80" "1024 60 80" "128 38 50" "128 72 96" "128 96 128"; do OS2D_HIP_LIB=tools/diag_libs/$V/libos2d_hip.so timeout 300 python tools/time_dft_phases.py $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dft_phases_${V}.txt | tail -12; done;;
    gemm|gemm:*)
      V=""; [ "$STEP" != gemm ] && V=${STEP#gemm:}
      if [ -n "$V" ]; then export OS2D_HIP_LIB=tools/diag_libs/$V/libos2d_hip.so; fi
      timeout 300 python tools/time_spectral16_quads.py 64 256 1024 2>&1 | grep -v amdgpu.ids | sed "s/^/[${V:-product}] /" | tee -a $OUT/gemm_times.txt | tail -6; unset OS2D_HIP_LIB;;
    power)
      bash tools/power_probe.sh > $OUT/power_probe.log 2>&1; cp -f gpurun_out/power/power_probe.txt $OUT/power_probe_raw.txt 2>/dev/null; grep -v "^LOOP\|{" $OUT/power_probe.log | tail -12; grep "^LOOP" $OUT/power_probe.log;;
    env:*)
      export "${STEP#env:}"; echo "exported ${STEP#env:}";;
    ab:*)        # ab:VAR=VALUE[,VAR=VALUE...][:step]: the stage times of the 64- and the 1024-class step (or another step) under these variables, labelled with them
      SPEC=${STEP#ab:}; SUB=stages; case "$SPEC" in *:*) SUB=${SPEC#*:}; SPEC=${SPEC%%:*};; esac
      ( IFS=,; for KV in $SPEC; do export "$KV"; done; export LABEL="$(echo "$SPEC" | sed 's|OS2D_HIP_LIB=tools/diag_libs/||; s|/libos2d_hip.so||')"
        case "$OS2D_HIP_LIB" in ""|/*) ;; *) export OS2D_HIP_LIB="$GRAFT_REPO_ROOT/$OS2D_HIP_LIB";; esac      # (some steps run from /tmp)
        bash "$0" "$TAG" $SUB ) | grep "classes\|_kernel\|span";;
    unset:*)
      unset "${STEP#unset:}";;
    bin:*)
      tools/bin/${STEP#bin:} 2>&1 | tee $OUT/${STEP#bin:}.txt | tail -12;;
    soak|soak:*)
      R=300; [ "$STEP" != soak ] && R=${STEP#soak:}
      ( time bash tools/soak_multistream.sh $R ) > $OUT/soak.log 2>&1; cp -f gpurun_out/soak.txt $OUT/soak.txt 2>/dev/null; grep -c "^ *[0-9]* same" $OUT/soak.txt; grep -i "DIFF\|RESULT\|real" $OUT/soak.log $OUT/soak.txt | cut -c1-200 | head -12;;
    mfma)
      tools/bin/mfma_peak 2>&1 | tee $OUT/mfma_peak.txt;;
    counters:*)      # counters:<classes>[:<kernel substring>]: tools/ab_counters.py under the current $OS2D_HIP_LIB / environment
      SPEC=${STEP#counters:}; N=${SPEC%%:*}; K=""; case "$SPEC" in *:*) K=$(echo "${SPEC#*:}" | tr ':' ' ');; esac
      timeout 600 python tools/ab_counters.py $N $K 2>&1 | grep "^\[\|error" | tee -a $OUT/counters.txt;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3;;
    py:*)
      timeout 900 python ${STEP#py:} 2>&1 | tee $OUT/$(basename ${STEP#py:} .py).txt | tail -40;;
    *) echo "unknown step $STEP";;
  esac
done
