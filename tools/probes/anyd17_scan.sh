#!/bin/bash
# orders 130 ... 257 at a decimation other than 8: the hot kernel's any-D form (8- and 16-wave workgroups) against the general kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in "--workload iqbb_fm_cu8 --order 255 --decim 125" "--workload iqbb_fm_cu8 --order 200 --decim 20 --fc 0" "--workload iqbb_usb --order 255 --decim 62" "--workload iqbb_usb --order 161 --decim 9"; do
  for hot in 0 1; do
    echo -n "$wl SDRHIP_IQBB_HOT=$hot: "
    SDRHIP_IQBB_HOT=$hot python bench.py $wl --no-cpu-baseline 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], r["sustained_ms_per_launch"], "%.1f %%" % (100*r["sustained_frac"]), r["kernels_per_step"], d.get("verified"))'
  done
done
