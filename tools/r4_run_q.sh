#!/bin/bash
# round 4, GPU call Q: plane borders written by the inverse transform instead of a launch of their own
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests/test_dft_gpu.py tests/test_head_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for v in 1 0 1 0; do
  OS2D_BORDERS_IN_INVERSE=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-precision --no-live-counters --no-end-to-end > gpurun_out/r4q/bench_$v.json 2> gpurun_out/r4q/bench_$v.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4q/bench_$v.json") if l.startswith("{")][-1])
print("borders_in_inverse=$v", d["value"], d["ms_per_step"], d["stages_ms"], d["config"].get("classes_1024_one_gpu",{}).get("ms_per_step"), d["config"].get("classes_1024_one_gpu",{}).get("layer7x7_ms"))
PY
done
