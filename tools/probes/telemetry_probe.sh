#!/bin/bash
# what clock / power telemetry an ordinary user can read on the GPU box (for bench.py's roofline.sclk_mhz / power_w)
for d in /sys/class/drm/card*/device; do
  echo "== $d"; ls $d | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk gpu_busy_percent current_link_speed; do [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }; done
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo
    for f in power1_average power1_input power1_cap freq1_input freq1_label freq2_input freq2_label temp1_input; do [ -r $h/$f ] && echo "$f = $(cat $h/$f)"; done
  done
done
which amd-smi rocm-smi
timeout 60 rocm-smi --showclocks --showpower 2>&1 | head -40
timeout 60 amd-smi metric -g 0 --clock --power 2>&1 | head -60
