#!/bin/bash
# Diagnostic build with cycle stamps of a conv workgroup's life: abl/libconv_stamp.so (run from the repo root)
set -e
mkdir -p abl
OBJS=$(ls mxdetection_amd/_obj/*.o | grep -v "/conv.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I include -DMXDET_CONV_STAMP \
    -c mxdetection_amd/csrc/conv.hip -o abl/conv_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libconv_stamp.so abl/conv_stamp.o $OBJS
ls -la abl/libconv_stamp.so
