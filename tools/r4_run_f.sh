#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4f
( timeout 600 python -m pytest tests/test_dft_gpu.py -q -x -p no:cacheprovider ) > gpurun_out/r4f/dft.log 2>&1
echo "dft tests rc=$?"; tail -3 gpurun_out/r4f/dft.log
for g in 2 4; do
  echo "== forward G=$g" | tee -a gpurun_out/r4f/phases.txt
  for nb in 64 1024; do
    OS2D_DFT_FORWARD_G=$g OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so timeout 300 python tools/time_dft_phases.py $nb 2>&1 | grep "^dft forward" | tee -a gpurun_out/r4f/phases.txt
  done
done
( timeout 600 python bench.py --no-live-counters --no-cpu-baseline --no-end-to-end --no-other-precision ) > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4f/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print({k:(v["avg_launch_ms"], v["frac"]) for k,v in d["roofline_other"].items()})
    print("config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid"))}))
except Exception as e:
    print("no bench line", e)
PY
