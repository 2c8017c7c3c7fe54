#!/bin/bash
# rocprofv3 evidence on the GPU box (round 3; round 2's script with the reference operating point added): tools/profile_round3.sh <tag> <commit>
#   kernel trace + stats of the bench command; FETCH_SIZE / WRITE_SIZE in separate --pmc passes, ONE bench leg per pass
#   (tools/pmc_case.py), no trace domains mixed in; SQ counters of the single-instance and the 64-instance launch.
set -u
TAG=${1:-r3x}; COMMIT=${2:-unknown}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT/pmc
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline --steps 1000 --warmup 100 > $OUT/${TAG}_bench_profiled.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats.csv || tail -5 $OUT/trace.err
# the same command with every launch on one stream: the per-kernel durations here are what bench.py's no_overlap.kernel_ms must agree with
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o t -- python $REPO/bench.py --no-cpu-baseline --no-overlap --steps 1000 --warmup 100 > $OUT/${TAG}_bench_profiled_no_overlap.json 2> $OUT/trace1.err
DB=$(find $OUT/trace1 -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats_no_overlap.csv || tail -5 $OUT/trace1.err
for c in B1 B1_lean B64 B64_lean B256 B256_lean sampled c5 ref5000; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    k=$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/p_${c}_$k -o p -- python $REPO/tools/pmc_case.py $c > $OUT/p_${c}_$k.log 2>&1
    DB=$(find $OUT/p_${c}_$k -name "*.db" | head -1)
    [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py pmc $DB $OUT/pmc/${c}_$k.csv > /dev/null || tail -3 $OUT/p_${c}_$k.log
  done
done
for c in B1 B64; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sqa_$c -o a -- python $REPO/tools/pmc_case.py $c > $OUT/sqa_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/sqb_$c -o b -- python $REPO/tools/pmc_case.py $c > $OUT/sqb_$c.log 2>&1
  for k in a b; do DB=$(find $OUT/sq${k}_$c -name "*.db" | head -1); [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py pmc $DB $OUT/${TAG}_pmc_sq_${k}_$c.csv > /dev/null; done
done
cd $REPO
python tools/traffic_from_pmc.py $OUT/pmc $TAG $COMMIT
cp $OUT/pmc/traffic.json profiles/traffic.json      # the bench lines below carry this pass's traffic figures
timeout 600 python bench.py --steps 3000 --warmup 200 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_args.json 2>> $OUT/bench.err
tail -c 400 $OUT/${TAG}_bench.json
find $OUT -name "*.db" -delete; rm -rf $OUT/trace $OUT/trace1 $OUT/p_* $OUT/sqa_* $OUT/sqb_*
du -sh $OUT
