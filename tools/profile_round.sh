#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel stats and four separate PMC passes of the default bench run.
# usage (on the MI355X box, from the repo root): bash tools/profile_round.sh gpurun_out/prof_TAG
set -u
OUT=${1:-gpurun_out/prof}
export TMPDIR=/tmp
BENCH=${BENCH:-"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --euler-device-steps 0"}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1" -- $BENCH > /dev/null 2> "$OUT/pmc1.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc2" -- $BENCH > /dev/null 2> "$OUT/pmc2.err"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc3" -- $BENCH > /dev/null 2> "$OUT/pmc3.err"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc4" -- $BENCH > /dev/null 2> "$OUT/pmc4.err"
python3 tools/pmc_summary.py "$OUT/pmc_summary.csv" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep the merged scratch small: the raw traces stay on the box
rm -rf "$OUT"/pmc1 "$OUT"/pmc2 "$OUT"/pmc3 "$OUT"/pmc4 "$OUT"/stats
ls -la "$OUT"
