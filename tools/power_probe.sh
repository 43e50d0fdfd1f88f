#!/bin/bash
# Package power and clocks while one kernel of the head runs back to back (evidence for DESIGN 4.2: the matrix-bound kernels sit at
# the power limit).  Samples rocm-smi every 0.25 s during ~6 s loops of: the correlation stage (padded / packed), the whole step
# at 64 classes, the strict-fp32 step, and idle.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/power
OUT=gpurun_out/power/power_probe.txt
: > $OUT
sample() {   # $1 = label, runs until the background job $2 exits
  while kill -0 $2 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    keep={k:v for k,v in c.items() if any(t in k.lower() for t in ('power','sclk','mclk','fclk'))}
    print('$1', json.dumps(keep))
except Exception as e:
    print('$1 parse-error', e)
" >> $OUT
    sleep 0.25
  done
}
loop() {     # $1 = label, rest = python snippet body that defines run()
  python - "$@" <<'PY' &
import os, sys, time, torch
REPO=os.getcwd(); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import util
from os2d_amd import _lib
from os2d_amd.utils import synthetic
lib=_lib.load(); dev=torch.device("cuda:0"); label=sys.argv[1]
H,W,B,C=60,80,64,1024
creator=util.make_head_creator(6, True, synthetic.make_transform_net_state(6, seed=1), dev)
fm=synthetic.make_feature_map(C,H,W,seed=3).to(dev)
with torch.no_grad():
    head=creator.create_os2d_head([c.to(dev) for c in synthetic.make_class_feature_maps(B,C,seed=5)])
    if label.startswith("corr"):
        form = 1 if label.endswith("packed") else 0
        qs=head._split_class_operand()
        ws=torch.empty(lib.os2d_corr_f16x3_packed_workspace_bytes(1,B,C,H,W),dtype=torch.uint8,device=dev)
        corr=torch.empty(B,225,H*W,device=dev); invn=torch.empty(B,H*W,device=dev); st=_lib.current_stream(dev)
        def run(): _lib.check(lib.os2d_corr_f16x3_packed(_lib.ptr(fm),_lib.ptr(qs),_lib.ptr(corr),_lib.ptr(invn),1,B,C,H,W,form,_lib.ptr(ws),ws.numel(),st),"corr")
    elif label=="idle":
        def run(): time.sleep(0.01)
    else:
        prec = "f32" if label=="step_f32" else "fftx3"
        def run(): head(fm, precision=prec)
    for _ in range(5): run()
    torch.cuda.synchronize(); t0=time.time(); n=0
    while time.time()-t0 < 6.0:
        for _ in range(20): run()
        torch.cuda.synchronize(); n+=20
    print("LOOP", label, "iterations", n, "ms each", round((time.time()-t0)/n*1e3,4))
PY
  sample "$1" $!
  wait
}
for label in idle corr_padded corr_packed step_fftx3 step_f32; do loop $label >> $OUT 2>&1; done
# the register-only matrix loop of tools/mfma_peak.hip (no LDS, no memory): all-zero and random operands
for mode in zero random; do
  tools/bin/mfma_peak loop $mode 6 >> $OUT 2>&1 &
  sample mfma_$mode $!
  wait
done
python - <<'PY'
import json, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("gpurun_out/power/power_probe.txt"):
    if l.startswith("LOOP"): print(l.strip()); continue
    parts=l.split(" ",1)
    if len(parts)<2 or not parts[1].startswith("{"): continue
    for k,v in json.loads(parts[1]).items():
        try: acc[parts[0]][k].append(float(str(v).strip("()MhzW ").replace("Mhz","")))
        except Exception: pass
for label,d in acc.items():
    print(label, {k:(round(sum(v[2:])/max(len(v[2:]),1),1), len(v)) for k,v in d.items()})
PY
