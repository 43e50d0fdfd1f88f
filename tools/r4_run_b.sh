#!/bin/bash
# round 4, GPU call B: the matrix-product transforms on the GPU for the first time - their own tests, then everything, then bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4b
( time timeout 600 python -m pytest tests/test_dft_gpu.py -q -x -p no:cacheprovider -s ) > gpurun_out/r4b/dft.log 2>&1
echo "dft rc=$?" | tee -a gpurun_out/r4b/dft.log
tail -25 gpurun_out/r4b/dft.log
( time timeout 1200 python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider --deselect tests/test_dft_gpu.py ) > gpurun_out/r4b/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4b/pytest.log
tail -30 gpurun_out/r4b/pytest.log
( time timeout 600 python bench.py --no-live-counters ) > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r4b/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4b/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print("roofline", {k:d["roofline"].get(k) for k in ("stage","bound","frac","avg_launch_ms")})
    print({k:(v["avg_launch_ms"], v["frac"]) for k,v in d["roofline_other"].items()})
    print("config", json.dumps(d["config"])[:1800])
    print("dev", d.get("max_abs_diff_vs_f32"))
except Exception as e:
    print("no bench line", e)
PY
