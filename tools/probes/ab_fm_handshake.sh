#!/bin/bash
# A/B on one box: FM at decimations other than 8 on few channels — the neighbours' handshake inside the hot kernel (one launch)
# against the second, tiny launch (SDRHIP_IQBB_FM_HANDSHAKE=0). ms per step of bench.py, every line verified.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for plan in "--workload iqbb_fm_cu8 --order 21 --decim 125 --fs 1e6 --width 12.5e3" "--workload iqbb_fm_cu8 --order 16 --decim 20 --fc 0" "--workload iqbb_fm_cu8 --order 21 --decim 4"; do
  for C in 1 16 128 512 1100; do
    for hs in 1 0 1 0; do
      SDRHIP_IQBB_FM_HANDSHAKE=$hs python bench.py $plan --channels $C --steps 200 --warmup 50 --no-cpu-baseline --sustain-seconds 0.3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$plan C=$C handshake=$hs', d['ms_per_step'], d['roofline']['sustained_ms_per_launch'], d['roofline']['kernels_per_step'], d['verified'])"
    done
  done
done
