set -e
# wgrad: split-K sweep on small / mid layers (last arg = forced ksplit, 0 = heuristic)
for shape in "2 13 21 256 256 3 1" "2 25 42 256 256 3 1" "2 50 84 256 256 3 1" "2 50 84 256 1024 1 1" "2 25 42 512 2048 1 1"; do
  for ks in 0 1 2 4 8 16 32; do
    timeout -k 10 120 python tools/bench_one_conv.py wgrad $shape 20 $ks 2>&1 | grep -v amdgpu.ids
  done
done
