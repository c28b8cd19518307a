set -e
# per-K-step cost: same tile grid, growing reduction depth. M = 2*64*128 = 16384 rows.
for cout in 128 256; do
 for c in 6 14 12; do
  for cin in 256 512 1024 2048; do
    timeout -k 10 120 python tools/bench_one_conv.py fwd 2 64 128 $cin $cout 1 1 40 $c 2>&1 | grep -v amdgpu.ids
  done
 done
done
