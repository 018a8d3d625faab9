python -m pytest tests/test_gpu_parity.py tests/test_gpu_kats.py -m gpu -x -q > gpurun_out/t1.log 2>&1; tail -2 gpurun_out/t1.log
export TMPDIR=/tmp
for L in default PN; do
  LIBARG=""; [ "$L" != "default" ] && LIBARG="--lib matchtigs_amd/libmatchtigs_$L.so"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab_$L -- python3 tools/sssp_probe.py --log2-edges 24 27 --reps 3 $LIBARG > gpurun_out/probe_ab_$L.jsonl 2>/dev/null
  find gpurun_out/prof_ab_$L -name "*kernel_trace.csv" -exec cp {} gpurun_out/ab_$L.csv \; ; rm -rf gpurun_out/prof_ab_$L
  echo "== $L"; python3 tools/kernel_times.py gpurun_out/ab_$L.csv
  python3 -c "
import json,sys
for l in open('gpurun_out/probe_ab_$L.jsonl'):
    r=json.loads(l); print(r['workload'], r['stage_ms_all'], [round(x['ms'],3) for x in r['levels']])"
done
