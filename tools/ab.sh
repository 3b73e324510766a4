#!/bin/bash
# A/B on ONE box: bench.py with the library builds named on the command line, alternating, REPS times each.
# usage: tools/ab.sh "<bench args>" libA.so libB.so@ENV=VAL ...      (paths relative to the repo root; an optional
# @NAME=VALUE suffix sets an environment variable for that variant, e.g. libsdr_amd/libsdrhip.so@SDRHIP_IQBB_HOT=0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for REP in $(seq 1 ${REPS:-3}); do
  for V in "$@"; do
    L=${V%%@*}; E=""; [ "$L" != "$V" ] && E=${V#*@}
    R=$(env SDRHIP_LIB=$L $E timeout 300 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_ms'], r.get('sustained_ms_per_launch'))" 2>&1)
    echo "[$V] ms_per_step, avg_launch_ms, sustained_ms: $R"
  done
done
