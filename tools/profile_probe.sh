#!/bin/bash
# rocprofv3 evidence for the SSSP probe (tools/sssp_probe.py): kernel stats + separate PMC passes (one counter group each: FETCH_SIZE
# and WRITE_SIZE together exceed the hardware). usage: bash tools/profile_probe.sh OUTDIR "<probe args>"
set -u
OUT=${1:-gpurun_out/pprof}
ARGS=${2:-"--log2-edges 27 --reps 3"}
export TMPDIR=/tmp
mkdir -p "$OUT"
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/sssp_probe.py $ARGS > "$OUT/probe.json" 2> "$OUT/stats.err"; echo "stats rc=$?"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1" -- python3 tools/sssp_probe.py $ARGS > /dev/null 2> "$OUT/pmc1.err"; echo "pmc1 rc=$?"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc2" -- python3 tools/sssp_probe.py $ARGS > /dev/null 2> "$OUT/pmc2.err"; echo "pmc2 rc=$?"
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc3" -- python3 tools/sssp_probe.py $ARGS > /dev/null 2> "$OUT/pmc3.err"; echo "pmc3 rc=$?"
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc4" -- python3 tools/sssp_probe.py $ARGS > /dev/null 2> "$OUT/pmc4.err"; echo "pmc4 rc=$?"
python3 tools/pmc_summary.py "$OUT/pmc_summary.csv" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3" "$OUT/pmc4"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT"/pmc1 "$OUT"/pmc2 "$OUT"/pmc3 "$OUT"/pmc4 "$OUT"/stats
grep -E "sssp_|sort_cand" "$OUT/kernel_stats.csv" | cut -c1-200
grep -E "sssp_enum|sort_cand" "$OUT/pmc_summary.csv"
