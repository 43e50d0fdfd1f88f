#!/bin/bash
# round 4, GPU call A: full GPU test suite (tightened tolerances, traced transforms) + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4a
( time python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider ) > gpurun_out/r4a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -40 gpurun_out/r4a/pytest.log
( time python bench.py ) > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r4a/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4a/bench.json") if l.startswith("{")][-1])
print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
print("roofline", {k:d["roofline"].get(k) for k in ("stage","bound","frac","avg_launch_ms","traffic")})
print("config", json.dumps(d["config"])[:1500])
for e in d.get("sweep",[]): print(e["name"], e["value"], e["ms_per_step"], e.get("stages_ms"), e.get("pyramid_serial_ms"))
PY
