#!/bin/bash
# Counters of the SSSP stage's kernels for two builds of the library, same box:
#   tools/pmc_ab_sssp.sh BASE.so [log2_edges=27]     -> gpurun_out/pmc_ab/{base,new}_{g1,g2,g3}.csv (tools/pmc_summary.py form)
BASE=${1:-tools/ab_libs/libmatchtigs_base.so}; LG=${2:-27}
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ab; mkdir -p $OUT
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"
G2="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"
for arm in base new; do
  LIBARG=""; [ $arm = base ] && LIBARG="--lib $BASE"
  for g in 1 2 3; do
    eval "CTR=\$G$g"
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/raw_${arm}_g$g -- python3 tools/sssp_probe.py --log2-edges $LG --reps 2 $LIBARG > $OUT/${arm}_g$g.log 2>&1
    echo "$arm g$g rc=$?"
    python3 tools/pmc_summary.py $OUT/${arm}_g$g.csv $OUT/raw_${arm}_g$g > /dev/null 2>&1
    rm -rf $OUT/raw_${arm}_g$g
  done
done
grep -h "sssp_enum" $OUT/*.csv | sort | cut -c1-160
