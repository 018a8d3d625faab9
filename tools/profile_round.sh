#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel stats of the bench command and four separate PMC passes (one counter group
# each) of the same workload. usage (on the MI355X box, from the repo root): [SIZE="--log2-edges 24"] bash tools/profile_round.sh gpurun_out/prof_TAG
# The stats pass runs the bench command itself (default: python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline); the PMC passes run
# one warm-up and one timed step of it (the counters are per launch, and a step of the exact mode spends 11 s in the host walk).
set -u
OUT=${1:-gpurun_out/prof}
SIZE=${SIZE:-""}
export TMPDIR=/tmp
BENCH=${BENCH:-"python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --device-mode-steps 0 --extra-seeds= $SIZE"}
PMCBENCH=${PMCBENCH:-"python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --device-mode-steps 0 --extra-seeds= $SIZE"}
T="timeout -k 10 420"
mkdir -p "$OUT"
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.json" 2> "$OUT/stats.err"; echo "stats rc=$?"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1" -- $PMCBENCH > /dev/null 2> "$OUT/pmc1.err"; echo "pmc1 rc=$?"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc2" -- $PMCBENCH > /dev/null 2> "$OUT/pmc2.err"; echo "pmc2 rc=$?"
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc3" -- $PMCBENCH > /dev/null 2> "$OUT/pmc3.err"; echo "pmc3 rc=$?"
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc4" -- $PMCBENCH > /dev/null 2> "$OUT/pmc4.err"; echo "pmc4 rc=$?"
python3 tools/pmc_summary.py "$OUT/pmc_summary.csv" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep the merged scratch small: the raw traces stay on the box
rm -rf "$OUT"/pmc1 "$OUT"/pmc2 "$OUT"/pmc3 "$OUT"/pmc4 "$OUT"/stats
ls -la "$OUT"
