#!/bin/bash
# Builds libsdr_amd/libsdrhip_<name>.so = the library with iqbb_i16.hip compiled under extra flags (tuning A/B:
# tools/ab.sh runs bench.py against several such builds on one box). usage: tools/build_variant.sh <name> "<flags>"
set -e
cd $(dirname $0)/../libsdr_amd/csrc
NAME=$1; FLAGS=$2
make -s -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $FLAGS -c iqbb_i16.hip -o _obj/iqbb_i16_$NAME.o
OBJS=$(ls _obj/*.o | grep -v "iqbb_i16")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsdrhip_$NAME.so $OBJS _obj/iqbb_i16_$NAME.o -ldl
rm -f _obj/iqbb_i16_$NAME.o
echo built libsdr_amd/libsdrhip_$NAME.so
