#!/bin/bash
# whole-step A/B (one box, one call): each line = env override -> images/sec
run() { echo -n "$1: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
run "MXDET_TUNE_WG_TARGET=3072"
run "MXDET_TUNE_WG_TARGET=4608"
run "MXDET_TUNE_WG_TARGET=6144"
run "MXDET_TUNE_WG_TARGET=9216"
run "MXDET_TUNE_WG_TARGET=6144 MXDET_TUNE_WG_MAXSTEPS=192"
run "MXDET_TUNE_WG_TARGET=3072 MXDET_TUNE_WG_MAXSTEPS=256"
run "MXDET_TUNE_WG_TARGET=3072 MXDET_TUNE_WG_MINSTEPS=96 MXDET_TUNE_WG_MAXSTEPS=192"
run "MXDET_TUNE_WG_TARGET=2048"
run "MXDET_TUNE_WG_TARGET=3072"
