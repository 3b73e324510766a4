#!/bin/bash
# tools/ab.sh for any workload: usage tools/ab_wl.sh "<bench args incl. --workload>" libA.so libB.so ...   (REPS, default 2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=$1; shift
for REP in $(seq 1 ${REPS:-2}); do
  for V in "$@"; do
    L=${V%%@*}; E=""; [ "$L" != "$V" ] && E=${V#*@}
    R=$(env SDRHIP_LIB=$L $E timeout 300 python bench.py --no-cpu-baseline $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('sustained_frac'))" 2>&1 | tail -1)
    echo "[$V] value, ms_per_step, frac, sustained_frac: $R"
  done
done
