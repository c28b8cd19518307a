#!/bin/bash
# Ablation / variant build of ONE source file: tools/build_abl_one.sh <file.hip> <name> [-DFLAG ...]  -> abl/lib<name>.so
set -e
src=$1; name=$2; shift 2
mkdir -p abl
b=$(basename $src)
case $b in conv.hip|wgrad.hip|dense_misc.hip) X="";; *) X="-ffp-contract=off";; esac
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/$b.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include $X "$@" -c mxdetection_amd/csrc/$b -o abl/${b}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/lib$name.so abl/${b}_$name.o $OBJS
ls abl/lib$name.so
