#!/bin/bash
# whole-step A/B of conv kernel variants (one box, one call): each line = env override -> images/sec
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
run MXDET_TUNE_V64X128=0
run MXDET_TUNE_V64X128=1
run MXDET_TUNE_V64X128=2
run MXDET_TUNE_V64X64=1
run MXDET_TUNE_V64X64=2
run MXDET_TUNE_V64X128=0
run MXDET_TUNE_V64X128=1
run MXDET_TUNE_V64X128=2
run MXDET_TUNE_V64X64=1
run MXDET_TUNE_V64X64=2
