#!/bin/bash
# whole-step A/B on one box: tools/ab.sh libA.so libB.so [reps] [extra bench args]  ("-" = the in-tree library)
a=$1; b=$2; reps=${3:-2}; shift 3 || true
run() { if [ "$1" = "-" ]; then unset MXDET_LIB; else export MXDET_LIB=$PWD/$1; fi
  echo -n "$1: "; timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 60 "${@:2}" 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
for i in $(seq $reps); do run $a "$@"; run $b "$@"; done
