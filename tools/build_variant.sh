#!/bin/bash
# Builds libsdr_amd/libsdrhip_<name>.so = the library with the K1 sources (iqbb_i16.hip, iqbb_hot_s*.hip; or the sources
# named on the command line) compiled under extra flags (tuning A/B: tools/abk1.py times several such builds against each other in one process on one box).
# usage: tools/build_variant.sh <name> "<flags>" [only-these-sources...]   e.g.  tools/build_variant.sh noepi "-DK1_ABL_NOEPI"
set -e
cd $(dirname $0)/../libsdr_amd/csrc
NAME=$1; FLAGS=$2; shift 2
make -s -j8 > /dev/null
K1=$(ls iqbb_i16.hip iqbb_hot_*.hip | tr '\n' ' ')
ALL=$(ls *.hip | tr '\n' ' ')
SRCS=" ${*:-$K1} "   # default: the K1 sources; name any other source (fir.hip, fftconv.hip ...) to rebuild it under the flags
mkdir -p _obj_$NAME
: > _obj_$NAME/Makefile.v
T=""
for f in $ALL; do
  o=_obj_$NAME/${f%.hip}.o
  if [[ "$SRCS" == *" $f "* ]]; then
    printf '%s: %s\n\t/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. %s -c %s -o %s\n' "$o" "$f" "$FLAGS" "$f" "$o" >> _obj_$NAME/Makefile.v
    T="$T $o"
  else
    cp _obj/${f%.hip}.o $o
  fi
done
echo "all:$T" >> _obj_$NAME/Makefile.v
make -s -j8 -f _obj_$NAME/Makefile.v all
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsdrhip_$NAME.so _obj_$NAME/*.o -ldl
rm -rf _obj_$NAME
echo built libsdr_amd/libsdrhip_$NAME.so
