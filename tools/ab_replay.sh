#!/bin/bash
# A/B of the claim replay's memory layout: kernel stats + FETCH/WRITE for lib A (records) and lib B (separate arrays)
export TMPDIR=/tmp
OUT=gpurun_out/ab_replay; mkdir -p $OUT
for L in A B; do
  LIB=""; [ $L = B ] && LIB="--lib matchtigs_amd/libmatchtigs_B.so"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$L -- python3 tools/sssp_probe.py --log2-edges 27 --reps 1 --replay $LIB > $OUT/s$L.out 2> $OUT/s$L.err
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$L -- python3 tools/sssp_probe.py --log2-edges 27 --reps 1 --replay $LIB > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$L -- python3 tools/sssp_probe.py --log2-edges 27 --reps 1 --replay $LIB > /dev/null 2>&1
  python3 tools/pmc_summary.py $OUT/pmc_$L.csv $OUT/f$L $OUT/w$L > /dev/null
  find $OUT/s$L -name "*kernel_stats.csv" -exec cp {} $OUT/stats_$L.csv \;
  echo "== $L"; grep -E "replay_rounds|replay_state_init|replay_dense|replay_compact" $OUT/stats_$L.csv | cut -d, -f1-4
  grep -E "^(replay_rounds|replay_state_init)" $OUT/pmc_$L.csv
  rm -rf $OUT/s$L $OUT/f$L $OUT/w$L
done
