#!/bin/bash
# One profiling round over every bench workload (run on the GPU box through gpurun): per workload tools/prof.sh
# (kernel trace + the PMC passes of the same bench.py command line) under gpurun_out/prof_<round>_<name>/, and the
# un-profiled bench.py lines in gpurun_out/<round>_bench.json (headline) and <round>_bench_other_workloads.jsonl.
# usage: tools/prof_all.sh r09 [name ...]      then, here:  cp gpurun_out/summ/* gpurun_out/r09_bench* profiles/
R=${1:-r09}; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
declare -A W=( [fm127]="" [fm16]="--order 16" [fm21]="--order 21" [fm64]="--order 64" [fm255]="--order 255" [usb127]="--workload iqbb_usb"
               [cu8]="--workload iqbb_fm_cu8" [real]="--workload bb_real_fm" [fir255]="--workload fir255_fm" [fbb]="--workload fbb_f32"
               [fftconv]="--workload fftconv --fft-whole-blocks" [fftola]="--workload fftconv_ola" [fftbank]="--workload fftbank" [fmdemod]="--workload fm_demod" [sub8]="--workload subsample8"
               [sdrfm]="--workload iqbb_fm_cu8 --order 21 --decim 125 --fs 1e6 --width 12.5e3" [cfg5g1]="--workload iqbb_usb --channels 8192 --batches 2"
               [cfg2c1]="--workload fbb_f32 --channels 1" [cfg1]="--workload fir127_fm" [multi4]="--buffers 4 --batches 2"
               [real5]="--workload bb_real_fm --decim 5" [real20]="--workload bb_real_fm --decim 20" [real125]="--workload bb_real_fm --decim 125"
               [o300]="--order 300" [o513d20]="--workload iqbb_usb --order 513 --decim 20" [i8doc]="--workload iqbb_fm_cs8 --order 16 --decim 24 --fc 0" [i8d8]="--workload iqbb_fm_cs8 --order 21"
               [sdrrec]="--workload iqbb_fm_cu8 --order 16 --decim 83 --fc 0"
               [sdrfmchain]="--workload iqbb_fm_cu8 --order 21 --decim 125 --deemph" [wfmchain]="--workload iqbb_fm_cu8 --order 16 --decim 20 --fc 0 --deemph"
               [pocsag]="--workload iqbb_fm_cu8 --order 21 --decim 45 --fc 0" [ssb]="--workload iqbb_usb --order 16 --decim 83 --fc 0"
               [sd4]="--workload iqbb_fm_cu8 --order 21 --decim 4" [sd7usb]="--workload iqbb_usb --order 21 --decim 7"
               [d300]="--workload iqbb_fm_cu8 --order 21 --decim 300" [d1000]="--workload iqbb_fm_cu8 --order 21 --decim 1000"
               [o255d125]="--workload iqbb_fm_cu8 --order 255 --decim 125" )   # (d300 / d1000: the large-decimation form; o255d125: the any-D form's long-filter class)   # round 4: the small-decimation form (not in the default list)   # examples/sdr_pocsag.cc:117 / sdr_ax25.cc:117 (21 taps, 1 MS/s to 22.05 kS/s); sdr_rec's USB mode on complex<int16>
#   # ... and the whole chains: + FMDeemph (sdr_fm.cc:44-53; sdr_rec.cc WFM: 16 taps, no shift, 1 MS/s to 50 kS/s)
#    # the plans of examples/sdr_fm.cc:40 and examples/sdr_rec.cc:42-68 (narrow FM: no shift, 1 MS/s to 12 kS/s)
# (sdrfm, cfg5g1, cfg2c1, cfg1, multi4, real20: the command lines of bench.py's "configs" entries — their workload keys must match for `traffic`)
NAMES=${@:-fm127 fm16 fm21 fm64 fm255 usb127 cu8 real fir255 fbb fftconv fftola fftbank fmdemod sub8 sdrfm cfg5g1 cfg2c1 cfg1 multi4 real5 real20 real125 o300 o513d20 sdrrec sdrfmchain wfmchain pocsag ssb sd4 sd7usb d300 d1000 o255d125}
mkdir -p gpurun_out
: > gpurun_out/${R}_bench_other_workloads.jsonl
for n in $NAMES; do
  if [ "$n" = fm127 ]; then python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err   # (the driver's own command line: headline + "configs")
  else python bench.py ${W[$n]} --cpu-seconds 3 2>/dev/null | grep '^{' >> gpurun_out/${R}_bench_other_workloads.jsonl; fi
done
python bench.py --workload fbb_f32 --channels 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/${R}_bench_other_workloads.jsonl
for n in $NAMES; do
  tools/prof.sh ${R}_$n ${W[$n]} > gpurun_out/prof_${R}_$n.log 2>&1
  # (the raw rocprofv3 trees are hundreds of MB: condense here, keep the summaries — they come back under gpurun_out/summ/)
  python tools/summarize_prof.py ${R}_$n gpurun_out/summ > gpurun_out/summ_${R}_$n.log 2>&1
  rm -rf gpurun_out/prof_${R}_$n
done
du -sh gpurun_out | tail -1
