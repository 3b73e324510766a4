#!/bin/bash
# Tuning aid for the GPU box: build the library once per set of -D switches and time bench.py for each, back to
# back ON THE SAME BOX. MI355X devices differ by ~10 % in the clock they hold under this kernel's load (DVFS), so
# only numbers from one gpurun call compare; always include the unmodified build ("") as the first variant.
# usage: tools/k1_variants.sh "" "-DFOO=1" "-DFOO=2" ...      (BENCH_ARGS="--workload iqbb_usb" to change the bench)
cd ${GRAFT_REPO_ROOT:-/root/repo}
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I../../include -I."
for V in "$@"; do
  rm -f libsdr_amd/csrc/_obj/iqbb_i16.o
  make -C libsdr_amd/csrc FLAGS="$BASE $V" > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  OUT=""
  for REP in 1 2; do
    R=$(timeout 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    OUT="$OUT $R"
  done
  echo "[$V] ms_per_step:$OUT"
done
rm -f libsdr_amd/csrc/_obj/iqbb_i16.o
make -C libsdr_amd/csrc > /dev/null 2>&1
