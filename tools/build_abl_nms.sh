#!/bin/bash
# Ablation builds of the NMS scan kernel: abl/libn_{NOCHAIN,NOOR,NOLOAD}.so
set -e
mkdir -p abl
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/boxes.hip.o")
for v in NOCHAIN NOOR NOLOAD; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -ffp-contract=off -Wno-unused-result -I include -DNMS_ABL_$v \
      -c mxdetection_amd/csrc/boxes.hip -o abl/boxes_$v.o &
done
wait
for v in NOCHAIN NOOR NOLOAD; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libn_$v.so abl/boxes_$v.o $OBJS
done
