#!/bin/bash
# K1 headline kernel: ms per step against the channel count on one box — the intercept of t(C) = a + b C is what a launch
# costs whatever its size (dispatch gap, ramp, tail), the slope the steady rate.   usage: k1_scan_channels.sh [extra bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for c in 256 512 1024 2048 4096 8192 1024; do
  python bench.py --channels $c --steps 200 --warmup 50 --no-verify --no-cpu-baseline --sustain-seconds 1.5 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('C=%5d  ms/step %.4f  sustained %.4f  frac %.4f  sclk %s MHz  power %s W' % (d['config']['channels_per_gpu'], d['ms_per_step'], r['sustained_ms_per_launch'], r['sustained_frac'], r.get('sclk_mhz'), r.get('power_w')))
"
done
