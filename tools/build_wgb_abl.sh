#!/bin/bash
# Ablation builds of the 256x256 weight-gradient tile (wrong results, timing only): abl/libwgb_{NOREAD,NODMA,NOMFMA,NOBAR,PRIO0,PRIO1}.so
set -e
mkdir -p abl
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/wgrad.hip.o")
for v in ABL_NOREAD ABL_NODMA ABL_NOMFMA ABL_NOBAR PRIO=0 PRIO=1; do
  n=${v//=/}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include -DMXDET_WGB_$v \
      -c mxdetection_amd/csrc/wgrad.hip -o abl/wgrad_$n.o &
done
wait
for v in ABL_NOREAD ABL_NODMA ABL_NOMFMA ABL_NOBAR PRIO=0 PRIO=1; do
  n=${v//=/}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libwgb_$n.so abl/wgrad_$n.o $OBJS
done
ls abl/libwgb_*.so
