cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "any_decimation or small_decim or fuzz" > gpurun_out/r15_anyd_tests.log 2>&1; tail -3 gpurun_out/r15_anyd_tests.log
for wl in "--workload iqbb_fm_cu8 --order 21 --decim 125" "--workload iqbb_fm_cu8 --order 16 --decim 83 --fc 0" "--workload iqbb_fm_cu8 --order 21 --decim 125 --deemph" "--workload iqbb_fm_cu8 --order 16 --decim 20 --fc 0 --deemph" "--workload iqbb_fm_cu8 --order 21 --decim 45 --fc 0" "--workload iqbb_fm_cu8 --order 21 --decim 4"; do
  for r in 0 1 0 1; do
    echo -n "$wl resident=$r: "
    SDRHIP_IQBB_FM_RESIDENT=$r python bench.py $wl --no-cpu-baseline 2>/dev/null | grep '^{' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], r["sustained_ms_per_launch"], r["kernels_per_step"], d.get("verified"))'
  done
done
