#!/bin/bash
# per-K-step cost on a C4-sized layer (M = 2*50*84 = 8400 rows), 1x1 conv, growing reduction depth, per tile config
for cout in 256 1024; do
 for c in 5 12 6 14 4 7; do
  for cin in 256 512 1024 2048 4096; do
    timeout -k 10 120 python tools/bench_one_conv.py fwd 2 50 84 $cin $cout 1 1 40 $c 2>&1 | grep -v amdgpu.ids || exit 1
  done
 done
done
