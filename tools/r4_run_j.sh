#!/bin/bash
# round 4, GPU call J: both correlation forms with fixed-point norms (same bits), form chosen per call; vectorised fp16 splits in
# the transforms; spectra in blocks of 64 pairs: full suite + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4j
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r4j/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4j/pytest.log
tail -8 gpurun_out/r4j/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4j/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print("  roofline", {k:d["roofline"].get(k) for k in ("stage","frac","avg_launch_ms","executed_frac_of_peak","traffic")})
    print({k:(v["avg_launch_ms"], v["frac"], v.get("traffic")) for k,v in d["roofline_other"].items()})
    print("  config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid","same"))}))
    for sw in d["sweep"]:
        print("  sweep", sw["name"][:40], sw["value"], sw["ms_per_step"], sw.get("stages_ms"))
    print("e2e", {k:d["end_to_end"][k] for k in ("value","ms_per_image","backbone_ms","head_ms","decode_nms_ms")})
except Exception as e:
    print("no bench line", e)
PY
du -sh gpurun_out | tail -1
