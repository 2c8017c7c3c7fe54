#!/bin/bash
# SQ counters of the rollout kernels at a given batch size: tools/pmc_sq.sh <tag> <B>
TAG=${1:-sq}; B=${2:-256}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BN_BS=$B timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/a -o a -- python $REPO/tools/wave_vs_role.py > $OUT/a.log 2>&1
BN_BS=$B timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/b -o b -- python $REPO/tools/wave_vs_role.py > $OUT/b.log 2>&1
cd $REPO
for k in a b; do DB=$(find $OUT/$k -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmc $DB $OUT/${TAG}_$k.csv || tail -3 $OUT/$k.log; done
find $OUT -name "*.db" -delete
grep -h "rollout" $OUT/${TAG}_a.csv $OUT/${TAG}_b.csv | cut -c1-40,100-400
