#!/bin/bash
# the hot kernel's small-decimation form (2 <= D <= 7) beside /8 and /9: ms per step of 1024 x 65536 samples, one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in iqbb_fm_cu8 iqbb_fm iqbb_usb; do
 for d in 2 3 4 5 6 7 8 9; do
  for hot in 1 0; do
  [ $hot = 0 ] && [ $d -ge 8 ] && continue
  SDRHIP_IQBB_HOT=$hot python bench.py --workload $wl --order 21 --decim $d --steps 100 --warmup 20 --no-cpu-baseline --sustain-seconds 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-12s order 21  D=%d  hot=$hot  %-24s ms/step %.4f  sustained %.4f  frac %.4f  verified %s' % ('$wl', $d, r['kernel'], d['ms_per_step'], r['sustained_ms_per_launch'], r['sustained_frac'], d.get('verified')))
"
  done
 done
done
