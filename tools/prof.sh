#!/bin/bash
# rocprofv3 passes for one bench.py command line; summaries land in gpurun_out/prof_<tag>/
# (every rocprofv3 run under `timeout`: a counter pass that hangs — seen once, WRITE_SIZE on the 8192-channel workload — costs minutes, not the call)
# usage: tools/prof.sh <tag> <bench args...>   (the kernel trace runs the command with 1 s of pre-conditioning launches in front, as the driver's run has 2 s: its average is the clock the chip holds; the --pmc passes run the SAME command line with --steps 3 --warmup 1 and no pre-conditioning)
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-cpu-baseline --no-configs "$@" --sustain-seconds 1 > $OUT/trace.log 2>&1
for CNT in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $CNT | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT/pmc_$N -- python3 $REPO/bench.py --no-cpu-baseline --no-configs --no-verify "$@" --steps 3 --warmup 1 --sustain-seconds 0 > $OUT/pmc_$N.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
find $OUT -name "*.csv" | wc -l
