#!/bin/bash
# Same-box A/B of the SSSP stage: tools/ab_sssp.sh BASE.so [log2_edges=27] [rounds=2]  (the tree's library is the other arm; results under gpurun_out/)
BASE=${1:-tools/ab_libs/libmatchtigs_base.so}; LG=${2:-27}; R=${3:-2}
export PYTHONUNBUFFERED=1
for i in $(seq 1 $R); do
  timeout -k 10 150 python -u tools/sssp_probe.py --log2-edges $LG --reps 5 --lib $BASE --out gpurun_out/probe_base_$i.json > gpurun_out/probe_base_$i.txt 2>&1 || exit 1
  timeout -k 10 150 python -u tools/sssp_probe.py --log2-edges $LG --reps 5 --out gpurun_out/probe_new_$i.json > gpurun_out/probe_new_$i.txt 2>&1 || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/probe_*_?.json')):
    d=json.load(open(f)); d=d[0] if isinstance(d,list) else d
    print(f, d.get('stage_ms_best'), [round(l['ms'],4) for l in d['levels']], [l['sources'] for l in d['levels']])
PY
