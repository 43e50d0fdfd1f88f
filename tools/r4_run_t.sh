#!/bin/bash
# round 4, GPU call T: full suite, the driver bench command, rocprofv3 passes (64 and 1024 classes) - final state of round 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4t
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r4t/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4t/pytest.log
tail -6 gpurun_out/r4t/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r4t/bench.json 2> gpurun_out/r4t/bench.err
echo "bench rc=$?"
bash tools/profile_bench.sh r04_fftx3 > gpurun_out/r4t/prof64.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r04_fftx3 > gpurun_out/r4t/r04_fftx3_rocprof_summary.txt 2>&1
bash tools/profile_bench.sh r04_fftx3_1024 --classes 1024 > gpurun_out/r4t/prof1024.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r04_fftx3_1024 > gpurun_out/r4t/r04_fftx3_1024_rocprof_summary.txt 2>&1
head -30 gpurun_out/r4t/r04_fftx3_rocprof_summary.txt
python - <<'PY'
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4t/bench.json") if l.startswith("{")][-1])
    print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
    print("roofline", json.dumps(d["roofline"])[:700])
    print({k:(v["avg_launch_ms"], v["frac"], v.get("traffic")) for k,v in d["roofline_other"].items()})
    print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("speedup_vs_cpu_baseline"))
    print("config", json.dumps({k:v for k,v in d["config"].items() if k.startswith(("classes_","pyramid","same"))}))
    print("e2e", {k:d["end_to_end"][k] for k in ("value","ms_per_image","backbone_ms","head_ms","decode_nms_ms")})
except Exception as e:
    print("no bench line", e)
PY
rm -rf gpurun_out/prof_r04_fftx3/*/*.db gpurun_out/prof_r04_fftx3_1024/*/*.db 2>/dev/null
find gpurun_out/prof_r04_fftx3 gpurun_out/prof_r04_fftx3_1024 -name "*.db" -size +8M -delete 2>/dev/null
du -sh gpurun_out | tail -1
