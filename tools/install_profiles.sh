#!/bin/bash
# Copies the summaries of one tools/profile_round.sh run (DIR, e.g. gpurun_out/prof_r06) into profiles/ under the round's TAG and refreshes
# the two files bench.py reads its counter traffic from (profiles/stage_traffic.json per stage, profiles/traffic.json for the SSSP stage).
#   usage: bash tools/install_profiles.sh r06 gpurun_out/prof_r06 [SIZE_LABEL=2p27] [WORKLOAD_KEY]
set -eu
TAG=$1; DIR=$2; SZ=${3:-2p27}; KEY=${4:-g_csr:log2_edges=27:k=31:seed=1:plan=0:gpus=1}
for M in device host; do
  [ -f "$DIR/kernel_stats_$M.csv" ] || continue
  cp "$DIR/kernel_stats_$M.csv" "profiles/${TAG}_kernel_stats_${M}_$SZ.csv"
  cp "$DIR/pmc_summary_$M.csv" "profiles/${TAG}_pmc_${M}_$SZ.csv"
  cp "$DIR/bench_$M.json" "profiles/${TAG}_profiled_bench_${M}_$SZ.json"
  cp "$DIR/raw_pmc1_$M.csv.gz" "profiles/${TAG}_raw_fetch_$M.csv.gz"
  cp "$DIR/raw_pmc2_$M.csv.gz" "profiles/${TAG}_raw_write_$M.csv.gz"
done
python3 - "$DIR/stage_traffic.json" <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
cur = json.load(open("profiles/stage_traffic.json"))
for k, v in new.items():
    if k != "comment":
        cur[k] = v
open("profiles/stage_traffic.json", "w").write(json.dumps(cur, indent=1) + "\n")
print("profiles/stage_traffic.json:", [k for k in new if k != "comment"])
PY
[ -f "$DIR/pmc_summary_host.csv" ] && python3 tools/make_traffic_json.py "$DIR/pmc_summary_host.csv" "$KEY" "$DIR/bench_host.json" "profiles/${TAG}_pmc_host_$SZ.csv"
