#!/bin/bash
# Ablation builds of the conv kernel (run here, from the repo root): abl/libc_{NOLOAD,NOMFMA,ZEROSRC}.so
set -e
mkdir -p abl
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/conv.hip.o")
for v in NOLOAD NOMFMA ZEROSRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include -DMXDET_ABL_$v \
      -c mxdetection_amd/csrc/conv.hip -o abl/conv_$v.o &
done
wait
for v in NOLOAD NOMFMA ZEROSRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libc_$v.so abl/conv_$v.o $OBJS
done
ls -la abl/*.so
