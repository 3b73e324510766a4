#!/bin/bash
# As tools/build_variant.sh, for fftconv.hip: libsdr_amd/libsdrhip_<name>.so with fftconv.hip compiled under extra flags.
set -e
cd $(dirname $0)/../libsdr_amd/csrc
NAME=$1; FLAGS=$2
make -s -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $FLAGS -c fftconv.hip -o _obj/fftconv_$NAME.o
OBJS=$(ls _obj/*.o | grep -v "fftconv")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsdrhip_$NAME.so $OBJS _obj/fftconv_$NAME.o -ldl
rm -f _obj/fftconv_$NAME.o
echo built libsdr_amd/libsdrhip_$NAME.so
