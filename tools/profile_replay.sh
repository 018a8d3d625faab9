#!/bin/bash
# PMC passes over the claim replay alone (tools/replay_probe.py): requests by kind, L2 <-> fabric requests, L2 hits, address translation.
# usage (on the MI355X box, from the repo root): bash tools/profile_replay.sh gpurun_out/prof_replay [log2_edges]
set -u
OUT=${1:-gpurun_out/prof_replay}
N=${2:-27}
export TMPDIR=/tmp
T="timeout -k 10 300"
mkdir -p "$OUT"
i=0
for group in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
             "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_ATOMIC_sum" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i + 1))
    $T rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/p$i" -- python3 tools/replay_probe.py $N 2 > "$OUT/p$i.out" 2> "$OUT/p$i.err"; echo "pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "replay" in k or "scan_" in k:
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/replay_pmc_summary.csv", "w") as o:
    o.write("kernel,counter,launches,mean_per_launch\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            v = agg[k][c]
            o.write(f"{k},{c},{len(v)},{sum(v)/len(v):.1f}\n")
print(open(out + "/replay_pmc_summary.csv").read())
PY
rm -rf "$OUT"/p[0-9]
