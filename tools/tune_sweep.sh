#!/bin/bash
# whole-step A/B of the conv tile thresholds (one box, one call): each line = env override -> images/sec
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
run MXDET_TUNE_T64=400
run MXDET_TUNE_T64=600
run MXDET_TUNE_T64=800
run MXDET_TUNE_T64=1200
run MXDET_TUNE_T64=2200
run MXDET_TUNE_T64=400
run MXDET_TUNE_T128=1024
run MXDET_TUNE_T128=3000
run MXDET_TUNE_T128=100000
run MXDET_TUNE_PAR64=800
run MXDET_TUNE_PAR64=4000
run MXDET_TUNE_T64=400
