#!/bin/bash
# Diagnostic build with in-kernel cycle stamps in the 256x256 weight-gradient tile: abl/libstamp.so (run from the repo root)
set -e
mkdir -p abl
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/wgrad.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include -DMXDET_WGB_STAMP \
    -c mxdetection_amd/csrc/wgrad.hip -o abl/wgrad_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libstamp.so abl/wgrad_stamp.o $OBJS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include -DMXDET_WGB_STAMP -DMXDET_WGB_PRIO=0 \
    -c mxdetection_amd/csrc/wgrad.hip -o abl/wgrad_stamp_l2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libstamp_l2.so abl/wgrad_stamp_l2.o $OBJS
ls -la abl/libstamp.so abl/libstamp_l2.so
