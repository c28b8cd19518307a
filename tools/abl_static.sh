#!/bin/bash
# static-tap K loop ablations on the GPU box (libraries from tools/build_abl_one.sh conv.hip s_<V> -DMXDET_ABL_<V>):
# what a 1x1 / 3x3 small-tile layer would cost without its DMA, without the filter side of its LDS traffic (a
# filter-stationary kernel's upper bound), without MFMAs, and without the K loop at all (fixed cost of the launch)
for spec in "2 50 84 256 1024 1 1 40" "2 50 84 1024 256 1 1 40" "2 100 168 128 512 1 1 40" "2 100 168 512 128 1 1 40" "2 50 84 256 256 3 1 40" "2 100 168 128 128 3 1 40"; do
  for v in BASE NOLOAD NOBLOAD NOMFMA NOKLOOP; do
    if [ $v = BASE ]; then unset MXDET_LIB; else export MXDET_LIB=$PWD/abl/libs_$v.so; fi
    echo -n "$v: "; timeout -k 10 120 python tools/bench_one_conv.py fwd $spec 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
