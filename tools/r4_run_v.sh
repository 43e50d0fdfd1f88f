#!/bin/bash
# round 4, GPU call V: the packed correlation in half tiles (two 4-wave work-groups per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4v
timeout 600 python -m pytest tests/test_head_gpu.py -m gpu -q -p no:cacheprovider -x -k "packed_correlation" 2>&1 | tail -3
for rep in 1 2; do
  for half in 0 1; do
    OS2D_CORR_HALF=$half OS2D_CORR_PACKED=1 timeout 300 python tools/time_corr_vs_channels.py 64 2>&1 | grep "packed.*C=1024\|form 1" | sed "s/^/half=$half /"
    OS2D_CORR_HALF=$half OS2D_CORR_PACKED=1 timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-precision --no-live-counters --no-end-to-end > gpurun_out/r4v/bench_$half.json 2> gpurun_out/r4v/bench_$half.err
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4v/bench_$half.json") if l.startswith("{")][-1])
sw={s["name"][:12]: (s["ms_per_step"], (s.get("stages_ms") or {}).get("corr")) for s in d["sweep"]}
print("half=$half packed=1", d["value"], d["ms_per_step"], d["stages_ms"], sw)
PY
  done
done
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-precision --no-live-counters --no-end-to-end --no-sweep 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default', d['value'], d['ms_per_step'], d['stages_ms'])"
