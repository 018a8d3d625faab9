#!/bin/bash
# A/B of two builds of the library on ONE box: rounds kernel of the claim replay under the default list (first line of the sweep)
#   usage: tools/ab_replay_libs.sh LIB_A LIB_B [graph ...]     (graph: log2 of the G-csr edges, or gseq:LENGTH)
A=$1; B=$2; shift 2
[ $# -eq 0 ] && set -- 24 27
for g in "$@"; do
  for rep in 1 2; do
    for L in "$A" "$B"; do
      echo -n "$g $(basename $L): "; MATCHTIGS_LIBRARY=$L python tools/replay_window_sweep.py $g 1 2>&1 | grep windows | cut -c30-120
    done
  done
done
