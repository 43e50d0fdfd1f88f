#!/bin/bash
# Stage times of the default workload (64 classes, f16x3) + any extra bench args; prints value and stages_ms.
python bench.py --steps 20 --warmup 3 --no-sweep --no-cpu-baseline --no-end-to-end --no-other-precision "$@" | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('pairs/s', d['value'], 'ms/step', d['ms_per_step'], d.get('stages_ms'))"
