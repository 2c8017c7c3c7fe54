#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box: tools/profile_round.sh <tag>
# (kernel trace + stats of the bench command; FETCH_SIZE and WRITE_SIZE in separate PMC passes, no trace domains mixed in)
set -u
TAG=${1:-rX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $BENCH --steps 1000 --warmup 100 > $OUT/bench_profiled.json 2> $OUT/trace.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- $BENCH --steps 200 --warmup 20 > /dev/null 2> $OUT/fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- $BENCH --steps 200 --warmup 20 > /dev/null 2> $OUT/write.err
cd $REPO
for kind in trace fetch write; do
  DB=$(find $OUT/$kind -name "*.db" | head -1)
  [ -n "$DB" ] || { echo "no db for $kind"; tail -5 $OUT/$kind.err; continue; }
  if [ $kind = trace ]; then python tools/rocpd_summary.py stats $DB $OUT/${TAG}_kernel_stats.csv; else python tools/rocpd_summary.py pmc $DB $OUT/${TAG}_pmc_$kind.csv; fi
done
timeout 300 python bench.py --steps 3000 --warmup 200 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
tail -c 600 $OUT/${TAG}_bench.json
find $OUT -name "*.db" -delete
du -sh $OUT
