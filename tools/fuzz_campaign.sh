#!/bin/bash
# One-off parity campaign on the GPU box, beyond what the suite holds: tests/fuzz_small.py (oracle == product through the HIP path and
# the clib.rs C-ABI; tests/fuzz_small.py says what is compared) over fresh seeds, under BOTH settings of the five out-of-tree policies
# (include/mtg_policy.h). One child process at a time; every chunk prints its TALLY line; a mismatch names its seed and ends the run.
#   usage: [SHIFT=0] tools/fuzz_campaign.sh OUT.txt [tiny_per_setting=10000] [medium_per_setting=600]   (SHIFT moves every seed range)
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/fuzz_campaign.txt}
TINY=${2:-10000}
MEDIUM=${3:-600}
SHIFT=${SHIFT:-0}
: > "$OUT"
run() {  # label mode first n [env...]
    local label=$1 mode=$2 first=$3 n=$4
    shift 4
    echo "== $label: $mode seeds $first..$((first + n - 1))" | tee -a "$OUT"
    env "$@" python tests/fuzz_small.py "$mode" "$first" "$n" 2>&1 | grep -E "^(TALLY|EVENTS|MISMATCH)" | tee -a "$OUT"
}
FL="MTG_POLICY=31 MATCHTIGS_LIBRARY=$PWD/matchtigs_amd/libmatchtigs_flipped.so"
for c in $(seq 0 $((TINY / 2500 - 1))); do
    run "default policies" gpu $((100000 + SHIFT + c * 2500)) 2500 MTG_NOP=1
    run "flipped policies" gpu $((200000 + SHIFT + c * 2500)) 2500 $FL
done
for c in $(seq 0 $((MEDIUM / 200 - 1))); do
    run "default policies" gpu_medium $((10000 + SHIFT + c * 200)) 200 MTG_NOP=1
    run "flipped policies" gpu_medium $((20000 + SHIFT + c * 200)) 200 $FL
done
echo "== done: $((2 * TINY)) tiny + $((2 * MEDIUM)) medium graphs, no mismatch" | tee -a "$OUT"
