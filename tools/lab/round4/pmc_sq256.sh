REPO=$(pwd); OUT=$REPO/gpurun_out/sq256; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in B256 B256_lean; do
BN_N=60 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/a_$c -o a -- python $REPO/tools/pmc_case.py $c > $OUT/a_$c.log 2>&1
BN_N=60 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_SALU -d $OUT/b_$c -o b -- python $REPO/tools/pmc_case.py $c > $OUT/b_$c.log 2>&1
done
cd $REPO
for c in B256 B256_lean; do for k in a b; do DB=$(find $OUT/${k}_$c -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmc $DB $OUT/sq_${k}_$c.csv || tail -3 $OUT/${k}_$c.log; done; done
find $OUT -name "*.db" -delete
