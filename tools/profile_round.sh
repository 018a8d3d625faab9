#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ (run on the MI355X box, from the repo root):
#   [SIZE="--log2-edges 24"] [MODES="device host"] bash tools/profile_round.sh gpurun_out/prof_TAG
# Per Euler mode (device = every stage on the GPU, host = reference-order walk over GPU-built records):
#   * one --kernel-trace --stats pass of the bench command (3 timed steps)           -> kernel_stats_<mode>.csv + bench_<mode>.json
#   * four separate --pmc passes (one counter group each: FETCH_SIZE and WRITE_SIZE cannot share a pass) of one warm-up and one timed step
#     -> pmc_summary_<mode>.csv (per kernel) and stage_traffic.json (per stage of the step, tools/stage_traffic.py)
# The program itself follows `--` (no env / bash -c hop: the profiler's library has initialised the GPU by then). A process that ran a
# cooperative launch segfaults in rocprofv3's teardown AFTER its result files are written on this image: exit codes 139, files complete.
set -u
# PASSES="stats pmc1 pmc2 pmc3 pmc4" (default: all); e.g. PASSES="pmc1 pmc2" re-measures only the HBM traffic
PASSES=${PASSES:-"stats pmc1 pmc2 pmc3 pmc4"}
has() { case " $PASSES " in *" $1 "*) return 0;; *) return 1;; esac; }
OUT=${1:-gpurun_out/prof}
SIZE=${SIZE:-""}
MODES=${MODES:-"device host"}
export TMPDIR=/tmp
T="timeout -k 10 420"
mkdir -p "$OUT"
LG=$(echo "$SIZE" | sed -n 's/.*--log2-edges \([0-9]*\).*/\1/p'); LG=${LG:-27}
for MODE in $MODES; do
  COMMON="--euler $MODE --device-mode-steps 0 --no-cpu-baseline --no-one-shot --extra-seeds= --full-size-log2 0 $SIZE"
  BENCH="python3 bench.py --steps 3 --warmup 1 $COMMON"
  PMCBENCH="python3 bench.py --steps 1 --warmup 1 --no-cold-steps $COMMON"
  has stats && $T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$MODE" -- $BENCH > "$OUT/bench_$MODE.json" 2> "$OUT/stats_$MODE.err"; echo "$MODE stats rc=$?"
  has pmc1 && $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1_$MODE" -- $PMCBENCH > /dev/null 2> "$OUT/pmc1_$MODE.err"; echo "$MODE pmc1 rc=$?"
  has pmc2 && $T rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc2_$MODE" -- $PMCBENCH > /dev/null 2> "$OUT/pmc2_$MODE.err"; echo "$MODE pmc2 rc=$?"
  has pmc3 && $T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc3_$MODE" -- $PMCBENCH > /dev/null 2> "$OUT/pmc3_$MODE.err"; echo "$MODE pmc3 rc=$?"
  has pmc4 && $T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc4_$MODE" -- $PMCBENCH > /dev/null 2> "$OUT/pmc4_$MODE.err"; echo "$MODE pmc4 rc=$?"
  python3 tools/pmc_summary.py "$OUT/pmc_summary_$MODE.csv" "$OUT/pmc1_$MODE" "$OUT/pmc2_$MODE" "$OUT/pmc3_$MODE" "$OUT/pmc4_$MODE"
  python3 tools/stage_traffic.py "$OUT/stage_traffic.json" "g_csr:log2_edges=$LG:k=31:seed=1:plan=0:gpus=1:$MODE" "$OUT/pmc1_$MODE" "$OUT/pmc2_$MODE"
  # (the raw per-dispatch FETCH / WRITE rows stay, compressed: the per-stage split can be redone without the GPU)
  for P in 1 2; do find "$OUT/pmc${P}_$MODE" -name "*counter_collection.csv" -exec sh -c 'gzip -c "$1" > "$2"' _ {} "$OUT/raw_pmc${P}_$MODE.csv.gz" \; ; done
  find "$OUT/stats_$MODE" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$MODE.csv" \;
  # keep the merged scratch small: the raw traces stay on the box
  rm -rf "$OUT"/pmc1_$MODE "$OUT"/pmc2_$MODE "$OUT"/pmc3_$MODE "$OUT"/pmc4_$MODE "$OUT"/stats_$MODE
done
ls -la "$OUT"
