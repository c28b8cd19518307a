set -e
for shape in "2 200 336 256 256 3 1" "2 50 84 256 256 3 1" "2 100 168 128 128 3 1" "2 50 84 1024 256 1 1" "2 25 42 512 512 3 1"; do
  for v in BASE NOLOAD NOMFMA; do
    if [ $v = BASE ]; then unset MXDET_LIB; else export MXDET_LIB=$PWD/abl/libw_$v.so; fi
    echo -n "$v: "; timeout -k 10 120 python tools/bench_one_conv.py wgrad $shape 30 0
  done
done
