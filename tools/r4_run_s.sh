#!/bin/bash
# round 4, GPU call S: non-temporal stores of the correlation output (write-once data, 276 MB per launch at 64 classes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s
for rep in 1 2; do
  for lib in product corr_nt; do
    if [ $lib = product ]; then unset OS2D_HIP_LIB; else export OS2D_HIP_LIB=tools/diag_libs/$lib/libos2d_hip.so; fi
    timeout 300 python tools/time_corr_vs_channels.py 64 2>&1 | grep "C=1024\|form" | sed "s/^/$lib /"
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-precision --no-live-counters --no-end-to-end > gpurun_out/r4s/bench_$lib.json 2> gpurun_out/r4s/bench_$lib.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4s/bench_$lib.json") if l.startswith("{")][-1])
sw={s["name"][:12]: (s["ms_per_step"], (s.get("stages_ms") or {}).get("corr")) for s in d["sweep"]}
print("$lib", d["value"], d["ms_per_step"], d["stages_ms"], sw)
PY
  done
done
