for spec in "2 50 84 1024 256 1 1 40" "2 100 168 512 128 1 1 40" "2 50 84 256 256 3 1 40" "2 100 168 128 128 3 1 40"; do
  for v in BASE NOBLOAD NOBFRAG; do
    if [ $v = BASE ]; then unset MXDET_LIB; else export MXDET_LIB=$PWD/abl/libs_$v.so; fi
    echo -n "$v: "; timeout -k 10 120 python tools/bench_one_conv.py fwd $spec 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
