#!/bin/bash
# rocprofv3 evidence on the GPU box, round 6: tools/profile_round6.sh <tag> <commit>
#   kernel trace + stats of the bench command (overlapped and one stream); FETCH_SIZE / WRITE_SIZE in separate --pmc passes, ONE bench
#   leg per pass (tools/pmc_case.py: now also B8, one GPU's share of configs[3] over 8), no trace domains mixed in; SQ counters of the
#   1-, 64- and 256-instance launches (full and lean); a kernel trace of the driver's 20-step timed region (tools/region_trace.py).
set -u
TAG=${1:-r6x}; COMMIT=${2:-unknown}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT/pmc
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py --no-cpu-baseline --steps 1000 --warmup 100 > $OUT/${TAG}_bench_profiled.json 2> $OUT/trace.err
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats.csv || tail -5 $OUT/trace.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o t -- python $REPO/bench.py --no-cpu-baseline --no-overlap --steps 1000 --warmup 100 > $OUT/${TAG}_bench_profiled_no_overlap.json 2> $OUT/trace1.err
DB=$(find $OUT/trace1 -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats_no_overlap.csv || tail -5 $OUT/trace1.err
# the driver's timed region itself (20 dependent solves + tail between two syncs), kernel by kernel
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/region -o r -- python $REPO/tools/region_trace.py > $OUT/region_host.txt 2> $OUT/region.err
CSV=$(find $OUT/region -name "*kernel_trace.csv" | head -1)
[ -n "$CSV" ] && (cat $OUT/region_host.txt; python $REPO/tools/region_trace_summary.py $CSV) > $OUT/${TAG}_region_trace.txt || tail -5 $OUT/region.err
for c in B1 B1_lean B8 B64 B64_lean B256 B256_lean sampled c5 ref5000; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    k=$( [ $ctr = FETCH_SIZE ] && echo fetch || echo write )
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/p_${c}_$k -o p -- python $REPO/tools/pmc_case.py $c > $OUT/p_${c}_$k.log 2>&1
    DB=$(find $OUT/p_${c}_$k -name "*.db" | head -1)
    [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py pmc $DB $OUT/pmc/${c}_$k.csv > /dev/null || tail -3 $OUT/p_${c}_$k.log
  done
done
for c in B1 B64 B256 B256_lean; do
  BN_N=60 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/sqa_$c -o a -- python $REPO/tools/pmc_case.py $c > $OUT/sqa_$c.log 2>&1
  BN_N=60 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/sqb_$c -o b -- python $REPO/tools/pmc_case.py $c > $OUT/sqb_$c.log 2>&1
  for k in a b; do DB=$(find $OUT/sq${k}_$c -name "*.db" | head -1); [ -n "$DB" ] && python $REPO/tools/rocpd_summary.py pmc $DB $OUT/${TAG}_pmc_sq_${k}_$c.csv > /dev/null; done
done
# round 6: the reference's own boundary -- MPPI.forward() once per control step, host in the loop -- kernel by kernel (one-launch path and
# host-paced loop), the timeline of one step on the chip-wide clock (timing build, tools/_ablate/lib_timing.so), and the class's latencies
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/fwd -o f -- python $REPO/tools/dropin_latency.py --quick > $OUT/${TAG}_dropin_latency_traced.txt 2> $OUT/fwd.err
DB=$(find $OUT/fwd -name "*.db" | head -1)
[ -n "$DB" ] && python $REPO/tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats_dropin.csv || tail -5 $OUT/fwd.err
(cd $REPO && timeout 200 python tools/dropin_latency.py > $OUT/${TAG}_dropin_latency.txt 2>&1)
if [ -f $REPO/tools/_ablate/lib_timing.so ]; then
  (cd $REPO && (timeout 120 python tools/stamps_forward.py; BN_PACED=1 timeout 120 python tools/stamps_forward.py; BN_REF=1 BN_PACED=1 timeout 120 python tools/stamps_forward.py) > $OUT/${TAG}_forward_timeline.txt 2>&1)
fi
(cd $REPO && [ -x tools/ubench_bg.bin ] || hipcc --offload-arch=gfx950 -O3 tools/ubench_bg.hip -o tools/ubench_bg.bin 2>/dev/null; timeout 300 python tools/cotenant_rate.py > $OUT/${TAG}_cotenant.txt 2>&1)
cd $REPO
python tools/traffic_from_pmc.py $OUT/pmc $TAG $COMMIT
cp $OUT/pmc/traffic.json profiles/traffic.json      # the bench lines below carry this pass's traffic figures
timeout 600 python bench.py --steps 3000 --warmup 200 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_args.json 2>> $OUT/bench.err
tail -c 300 $OUT/${TAG}_bench.json
find $OUT -name "*.db" -delete; rm -rf $OUT/trace $OUT/trace1 $OUT/region $OUT/p_* $OUT/sqa_* $OUT/sqb_*
du -sh $OUT
