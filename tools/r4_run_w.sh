#!/bin/bash
# round 4, GPU call W: last verification of the round - full GPU suite + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4w
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r4w/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r4w/pytest.log
tail -6 gpurun_out/r4w/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r4w/bench.json 2> gpurun_out/r4w/bench.err
echo "bench rc=$?"
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r4w/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','stages_ms')})
print({k:(v['avg_launch_ms'], v['frac']) for k,v in d['roofline_other'].items()})
print(json.dumps({k:v for k,v in d['config'].items() if k.startswith(('classes_','pyramid'))}))
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
