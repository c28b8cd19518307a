#!/bin/bash
# Build the library of a git revision (default HEAD) as abl/lib<name>.so for same-box A/B runs (MXDET_LIB=...).
# usage: tools/build_base.sh [rev] [name]
set -e
rev=${1:-HEAD}; name=${2:-base}
T=/tmp/mxdet_base_$name
rm -rf $T; mkdir -p $T abl
git archive $rev mxdetection_amd/csrc include | tar -x -C $T
for f in $T/mxdetection_amd/csrc/*.hip; do
  b=$(basename $f)
  case $b in conv.hip|wgrad.hip|dense_misc.hip) X="";; *) X="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I $T/include $X -c $f -o $T/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/lib$name.so $T/*.o
ls -la abl/lib$name.so
