#!/bin/bash
# round 4, GPU call C: full suite on the canonical-size planner + NMS pruning, phase breakdown of the transforms, decode timings, size churn
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4c
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r4c/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4c/pytest.log
tail -15 gpurun_out/r4c/pytest.log
for nb in 64 1024; do
  OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so timeout 300 python tools/time_dft_phases.py $nb 2>&1 | grep "^dft" | tee -a gpurun_out/r4c/phases.txt
done
OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so timeout 300 python tools/time_dft_phases.py 64 96 128 2>&1 | grep "^dft" | tee -a gpurun_out/r4c/phases.txt
timeout 600 python tools/bench_decode.py 512 -1e30 --pyramid --views 8 > gpurun_out/r4c/decode.txt 2>&1
grep "pyramid decode" gpurun_out/r4c/decode.txt
timeout 900 python tools/bench_size_churn.py > gpurun_out/r4c/size_churn.json 2> gpurun_out/r4c/size_churn.err
tail -c 1500 gpurun_out/r4c/size_churn.json
( time timeout 600 python bench.py --no-live-counters ) > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4c/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print({k:(v["avg_launch_ms"], v["frac"]) for k,v in d["roofline_other"].items()})
    print("config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid"))}))
except Exception as e:
    print("no bench line", e)
PY
