set -e
kind=${1:-fwd}
for shape in "2 200 336 256 256 3 1" "2 100 168 256 256 3 1" "2 200 336 256 256 1 1" "2 50 84 256 256 3 1" "2 50 84 1024 256 1 1" "2 100 168 128 128 3 1"; do
  for c in 0 14 15 16 17; do
    timeout -k 10 120 python tools/bench_one_conv.py $kind $shape 30 $c 2>&1 | grep -v amdgpu.ids
  done
done
