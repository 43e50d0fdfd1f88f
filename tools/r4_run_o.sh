#!/bin/bash
# round 4, GPU call O: transforms with the prefetched registers consumed before the store loops (no vmcnt(0) behind stores)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_dft_gpu.py tests/test_spectral_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
OS2D_HIP_LIB=tools/diag_libs/stamps/libos2d_hip.so timeout 300 python tools/time_dft_phases.py > gpurun_out/r4o/dft_phases_v4.txt 2>&1
tail -4 gpurun_out/r4o/dft_phases_v4.txt
( time timeout 900 python bench.py --steps 20 --warmup 3 ) > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4o/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print({k:(v["avg_launch_ms"], v["frac"]) for k,v in d["roofline_other"].items()})
    print("  config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid"))}))
except Exception as e:
    print("no bench line", e)
PY
