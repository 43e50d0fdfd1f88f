#!/bin/bash
# round 4, GPU call H: the packed correlation (classes stacked along M): stage tests, head tests, bench with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4h
( time timeout 900 python -m pytest tests/test_head_gpu.py -m gpu -q -p no:cacheprovider -x -k "correlation" ) > gpurun_out/r4h/corr_tests.log 2>&1
echo "corr tests rc=$?"; tail -15 gpurun_out/r4h/corr_tests.log
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r4h/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4h/pytest.log
tail -8 gpurun_out/r4h/pytest.log
for packed in 1 0; do
  ( time OS2D_CORR_PACKED=$packed timeout 900 python bench.py ) > gpurun_out/r4h/bench_packed$packed.json 2> gpurun_out/r4h/bench_packed$packed.err
  echo "bench packed=$packed rc=$?"
done
python - <<'PY'
import json
for packed in (1, 0):
    try:
        d=json.loads([l for l in open("gpurun_out/r4h/bench_packed%d.json" % packed) if l.startswith("{")][-1])
        print("packed", packed, {k:d[k] for k in ("value","ms_per_step","stages_ms")})
        print("  roofline", {k:d["roofline"].get(k) for k in ("stage","frac","avg_launch_ms","executed_frac_of_peak","traffic")})
        print("  config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid","same"))}))
        for sw in d["sweep"]:
            print("  sweep", sw["name"][:40], sw["value"], sw["ms_per_step"], sw.get("stages_ms"))
    except Exception as e:
        print("no bench line", packed, e)
PY
du -sh gpurun_out | tail -1
