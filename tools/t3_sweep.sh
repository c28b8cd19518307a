#!/bin/bash
# three-tap weight-gradient kernel: whole-step A/B (one box, one call).  -> stdout
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q -k "wgrad or layer_gradients or grouped" 2>&1 | tail -3
run "MXDET_TUNE_T3_ENABLE=0"
run "MXDET_TUNE_T3_MIX=0"
run "MXDET_TUNE_T3_MIX=1"
run "MXDET_TUNE_T3_MIX=2"
run "MXDET_TUNE_T3_MIX=1 MXDET_TUNE_WG_TARGET=2048"
run "MXDET_TUNE_T3_MIX=1 MXDET_TUNE_WG_TARGET=1536 MXDET_TUNE_T3_TARGET=1024"
run "MXDET_TUNE_T3_MIX=1 MXDET_TUNE_T3_TARGET=2048"
run "MXDET_TUNE_T3_ENABLE=0"
run "MXDET_TUNE_T3_MIX=1"
