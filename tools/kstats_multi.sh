#!/bin/bash
# rocprofv3 kernel stats of the SSSP probe for several library builds (same box): tools/kstats_multi.sh LG LIB [LIB ...]
LG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/kst; mkdir -p $OUT
n=0
for L in "$@"; do
  n=$((n+1)); ARG=""; [ "$L" != default ] && ARG="--lib $L"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$n -- python3 tools/sssp_probe.py --log2-edges $LG --reps 5 $ARG > $OUT/log_$n.txt 2>&1
  f=$(find $OUT/raw_$n -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('sssp_enum','sssp_kernel','sort_lists','fix_compact','active_range')) :
        print(f"   {n[:70]:70s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:.4f} ms min {float(r['MinNs'])/1e6:.4f}")
PY
  rm -rf $OUT/raw_$n
done
