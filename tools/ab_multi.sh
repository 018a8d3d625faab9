#!/bin/bash
# Same-box comparison of the SSSP stage for several builds of the library: tools/ab_multi.sh ROUNDS LG LIB [LIB ...]
# ("default" = the tree's library). Every round runs one process per library; prints per library all best-of-5 stage times.
R=$1; LG=$2; shift 2
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/abm
for i in $(seq 1 $R); do
  n=0
  for L in "$@"; do
    n=$((n+1)); ARG=""; [ "$L" != default ] && ARG="--lib $L"
    timeout -k 10 150 python -u tools/sssp_probe.py --log2-edges $LG --reps 5 $ARG --out gpurun_out/abm/p_${n}_$i.json > gpurun_out/abm/p_${n}_$i.txt 2>&1 || { echo "probe failed: $L"; tail -5 gpurun_out/abm/p_${n}_$i.txt; exit 1; }
  done
done
python - "$@" <<'PY'
import json,glob,sys
libs=sys.argv[1:]
for n,L in enumerate(libs,1):
    best=[];lv=[]
    for f in sorted(glob.glob(f'gpurun_out/abm/p_{n}_*.json')):
        d=json.load(open(f)); d=d[0] if isinstance(d,list) else d
        best.append(d['stage_ms_best']); lv=[(round(l['ms'],3),l['sources']) for l in d['levels']]
    print(f"{L:45s} min {min(best):.4f} med {sorted(best)[len(best)//2]:.4f} all {best} levels {lv}")
PY
