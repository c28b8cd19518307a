#!/bin/bash
run() { echo -n "[$1]: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -x -q -k "filter_stationary" 2>&1 | tail -12
for r in 1 2 3; do run "MXDET_TUNE_FS1X1=0"; run "MXDET_TUNE_FS1X1=1"; done
