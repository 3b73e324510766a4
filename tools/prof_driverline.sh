set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/prof_r19_driverline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/trace.log 2> $OUT/trace.err
echo rc $?
cd $REPO && python tools/summarize_prof.py r19_driverline gpurun_out/summ; rm -rf $OUT/trace
