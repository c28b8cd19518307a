#!/bin/bash
# whole-step A/B of one library under two environments: tools/ab_env.sh "ENV_A=.." "ENV_B=.." [reps] [bench args]
a=$1; b=$2; reps=${3:-2}; shift 3 || true
run() { echo -n "[$1]: "; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-conv-timer --steps 60 "${@:2}" 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' || exit 1; }
for i in $(seq $reps); do run "$a" "$@"; run "$b" "$@"; done
