#!/bin/bash
# The HBM-traffic passes alone (kernel trace for the workload key, then FETCH_SIZE and WRITE_SIZE in separate --pmc runs), every
# rocprofv3 run under a short timeout: for the workload whose counter passes hung in tools/prof.sh (8192 channels).
# usage: tools/prof_traffic.sh <tag> <bench args...>      -> gpurun_out/prof_<tag>/, condensed by tools/summarize_prof.py <tag>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-cpu-baseline --no-configs "$@" --sustain-seconds 0.3 > $OUT/trace.log 2>&1
echo "trace rc $?"
for CNT in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_$CNT -- python3 $REPO/bench.py --no-cpu-baseline --no-configs --no-verify "$@" --steps 2 --warmup 1 --sustain-seconds 0 > $OUT/pmc_$CNT.log 2>&1
  echo "$CNT rc $?"
done
cd $REPO && python tools/summarize_prof.py $TAG gpurun_out/summ; rm -rf $OUT/trace $OUT/pmc_*/
